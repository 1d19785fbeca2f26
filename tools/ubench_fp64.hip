// ubench_fp64.hip -- issue cost of the FP64 instructions behind the F64 policy of kernels_u64.hip
// (v_add_f64, v_mul_f64, v_fma_f64, v_rndne_f64, v_bfi_b32, and the whole modular multiplication), in
// SHADER CLOCKS (s_memtime) per wave-instruction and SIMD, 8 waves per SIMD.  Same method as ubench_valu2.hip.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 16;

#define REP(STMT) \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { STMT; } }

template <int K>
__global__ void __launch_bounds__(256) probe(double *out, unsigned long long *clk, int iters, double y, double z)
{
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = (double)(threadIdx.x * 7 + c) + 0.5;
    const double p = 1337006139375617.0, pinv = 1.0 / p;
    uint32_t i32 = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 0) { REP(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y))) }
        else if constexpr (K == 1) { REP(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y))) }
        else if constexpr (K == 2) { REP(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z))) }
        else if constexpr (K == 3) { REP(asm volatile("v_rndne_f64_e32 %0, %0" : "+v"(x[c]))) }
        else if constexpr (K == 4) {
            REP(asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(((uint32_t *)&x[c])[1]) : "v"(0x7fffffffu), "v"(((uint32_t *)&y)[1])))
        }
        else if constexpr (K == 5) {       // the F64 modular multiplication, 8 instructions
            REP({
                const double h = x[c] * y;
                const double l = __builtin_fma(x[c], y, -h);
                const double q = __builtin_rint(h * pinv);
                const double r = __builtin_fma(-q, p, h) + l;
                x[c] = r - __builtin_copysign(p, r);
            })
        }
        else if constexpr (K == 6) { REP(asm volatile("v_add_f64 %0, %0, %2\n\tv_add_u32_e32 %1, %1, %3" : "+v"(x[c]), "+v"(i32) : "v"(y), "v"(3u))) }
        else if constexpr (K == 7) { REP(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y))) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = z + (double)i32;
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
    if (acc == 0.123456) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K>
static void run(const char *name, int instr_per_body, double *dout, unsigned long long *dclk)
{
    const int blocks = 256 * 8, threads = 256, iters = 500;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    probe<K><<<blocks, threads>>>(dout, dclk, 10, 1.0000001, 0.25);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        probe<K><<<blocks, threads>>>(dout, dclk, iters, 1.0000001, 0.25);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), dclk, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[blocks / 2];
    double per_wave = (double)iters * UNROLL * CHAINS * instr_per_body;
    double cyc = med / (per_wave * 8.0);      // 8 waves per SIMD (2048 blocks x 4 waves / 1024 SIMDs)
    double winstr = (double)blocks * (threads / 64) * per_wave;
    double ghz = med / (best * 1e-3) * 1e-9;
    printf("%-34s %7.3f ms  %6.2f Tlane-op/s  %5.2f clk/wave-instr/SIMD  (in-kernel clock ~%.2f GHz)\n",
           name, best, winstr * 64 / (best * 1e-3) * 1e-12, cyc, ghz);
}

int main()
{
    double *dout;
    unsigned long long *dclk;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 8));
    CK(hipMalloc(&dclk, 256 * 8 * 8));
    run<0>("v_add_f64", 1, dout, dclk);
    run<1>("v_mul_f64", 1, dout, dclk);
    run<2>("v_fma_f64", 1, dout, dclk);
    run<3>("v_rndne_f64", 1, dout, dclk);
    run<7>("v_max_f64", 1, dout, dclk);
    run<4>("v_bfi_b32", 1, dout, dclk);
    run<6>("v_add_f64 + v_add_u32", 2, dout, dclk);
    run<5>("F64 modmul (8 instr)", 8, dout, dclk);
    return 0;
}
