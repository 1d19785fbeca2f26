// ubench_valu2.hip -- second VALU probe for gfx950: which simple integer ops issue in 2 cycles
// and which in 4, measured in SHADER CLOCKS (s_memtime) so DVFS does not blur the answer.
// Feeds DESIGN.md's butterfly cost model.  Build like ubench_intmul.hip.
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
#define A2(INS) REP(asm volatile(INS " %0, %0, %1" : "+v"(x[c]) : "v"(y)))
#define A2R(INS) REP(asm volatile(INS " %0, %1, %0" : "+v"(x[c]) : "v"(y)))
#define A3(INS) REP(asm volatile(INS " %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))

template <int K>
__global__ void __launch_bounds__(256) probe(uint32_t *out, unsigned long long *clk, int iters,
                                            uint32_t y, uint32_t z)
{
    uint32_t x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 2654435761u + c;
    unsigned long long x64[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x64[c] = x[c];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 0) { A2("v_add_u32_e32") }
        else if constexpr (K == 1) { A2("v_add_u32_e64") }
        else if constexpr (K == 2) { A2("v_min_u32_e32") }
        else if constexpr (K == 3) { A2("v_max_u32_e32") }
        else if constexpr (K == 4) { A2("v_and_b32_e32") }
        else if constexpr (K == 5) { A2("v_xor_b32_e32") }
        else if constexpr (K == 6) { A2R("v_lshlrev_b32_e32") }
        else if constexpr (K == 7) { A2R("v_ashrrev_i32_e32") }
        else if constexpr (K == 8) { REP(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(y) : "vcc")) }
        else if constexpr (K == 9) { REP(asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1" : "+v"(x[c]) : "v"(y) : "vcc")) }
        else if constexpr (K == 10) {
            REP(asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc"
                             : "+v"(x[c]) : "v"(y), "v"(z) : "vcc"))
        }
        else if constexpr (K == 11) { REP(asm volatile("v_cmp_ge_u32_e32 vcc, %0, %1" :: "v"(x[c]), "v"(y) : "vcc")) }
        else if constexpr (K == 12) { A3("v_lshl_add_u32") }
        else if constexpr (K == 13) { A3("v_med3_u32") }
        else if constexpr (K == 14) { A2("v_mul_lo_u32") }
        else if constexpr (K == 15) {
            REP(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x64[c]) : "v"(y), "v"(z) : "vcc"))
        }
        else if constexpr (K == 16) { A2("v_subrev_u32_e32") }
        else if constexpr (K == 17) { REP(asm volatile("v_mov_b32_e32 %0, %1" : "+v"(x[c]) : "v"(y))) }
        else if constexpr (K == 18) { A3("v_add3_u32") }
        else if constexpr (K == 19) { A2("v_min_i32_e32") }
        else if constexpr (K == 20) { A2("v_sub_u32_e32") A2("v_min_u32_e32") }
        else if constexpr (K == 21) { A2("v_pk_add_u16") }
        else if constexpr (K == 22) { A2("v_or_b32_e32") }
        else if constexpr (K == 23) { A3("v_bfe_u32") }
        else if constexpr (K == 24) { A3("v_xad_u32") }
        else if constexpr (K == 25) { A3("v_and_or_b32") }
        else if constexpr (K == 26) { A2("v_mul_u32_u24_e32") }
        else if constexpr (K == 27) { A2("v_add_f32_e32") }
        else if constexpr (K == 28) { A3("v_fma_f32") }
        else if constexpr (K == 29) { A2("v_pk_fma_f32 %0, %0, %1,") }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c] ^ (uint32_t)x64[c];
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K>
static void run(const char *name, int instr_per_body, uint32_t *dout, unsigned long long *dclk)
{
    const int blocks = 256 * 8, threads = 256, iters = 1000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    probe<K><<<blocks, threads>>>(dout, dclk, 10, 3u, 5u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        probe<K><<<blocks, threads>>>(dout, dclk, iters, 3u, 5u);
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
    // each SIMD holds 8 waves (2048 blocks x 4 waves / 1024 SIMDs); per-wave instruction count:
    double per_wave = (double)iters * UNROLL * CHAINS * instr_per_body;
    double cyc = med / (per_wave * 8.0);
    double winstr = (double)blocks * (threads / 64) * per_wave;
    double ghz = med / (best * 1e-3) * 1e-9;
    printf("%-30s %7.3f ms  %6.2f Tlane-op/s  %5.2f clk/wave-instr/SIMD  (in-kernel clock ~%.2f GHz)\n",
           name, best, winstr * 64 / (best * 1e-3) * 1e-12, cyc, ghz);
}

int main()
{
    uint32_t *dout;
    unsigned long long *dclk;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    CK(hipMalloc(&dclk, 256 * 8 * 8));
#define R(K, NAME, N) run<K>(NAME, N, dout, dclk)
    R(0, "v_add_u32_e32", 1);
    R(1, "v_add_u32_e64", 1);
    R(16, "v_subrev_u32_e32", 1);
    R(2, "v_min_u32_e32", 1);
    R(3, "v_max_u32_e32", 1);
    R(19, "v_min_i32_e32", 1);
    R(4, "v_and_b32_e32", 1);
    R(22, "v_or_b32_e32", 1);
    R(5, "v_xor_b32_e32", 1);
    R(6, "v_lshlrev_b32_e32", 1);
    R(7, "v_ashrrev_i32_e32", 1);
    R(17, "v_mov_b32_e32", 1);
    R(8, "v_cndmask_b32_e32", 1);
    R(9, "v_sub_co_u32_e32", 1);
    R(10, "v_sub_co + v_cndmask", 2);
    R(11, "v_cmp_ge_u32_e32", 1);
    R(20, "v_sub + v_min", 2);
    R(12, "v_lshl_add_u32", 1);
    R(18, "v_add3_u32", 1);
    R(13, "v_med3_u32", 1);
    R(23, "v_bfe_u32", 1);
    R(24, "v_xad_u32", 1);
    R(25, "v_and_or_b32", 1);
    R(14, "v_mul_lo_u32", 1);
    R(26, "v_mul_u32_u24_e32", 1);
    R(15, "v_mad_u64_u32", 1);
    R(27, "v_add_f32_e32", 1);
    R(28, "v_fma_f32", 1);
    return 0;
}
