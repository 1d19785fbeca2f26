// ubench_intmul.hip -- VALU integer-instruction throughput probe for gfx950 (MI355X).
//
// SURVEY 7 "hard parts": the NTT butterfly is 3 x 32-bit multiplies + ~5 simple ops, and 32-bit
// integer multiplies are believed to be quarter-rate on CDNA.  This probe measures the issue
// rate of each candidate instruction (8 independent chains per lane, every SIMD holding 8
// waves) so that DESIGN.md's VALU roofline is a measured number, not a guess.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_intmul.hip -o tools/ubench_intmul
// Output: one line per instruction: lanes-ops/s chip-wide and cycles per wave64 instruction
// per SIMD at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 16;

#define BODY2(INS) \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) \
            asm volatile(INS " %0, %0, %1" : "+v"(x[c]) : "v"(y)); }
#define BODY3(INS) \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) \
            asm volatile(INS " %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)); }

template <int K>
__global__ void __launch_bounds__(256) probe(uint32_t *out, int iters, uint32_t y, uint32_t z)
{
    uint32_t x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 2654435761u + c;
    unsigned long long x64[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x64[c] = x[c];
    double d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) d[c] = (double)x[c];
    double dy = (double)y * 1e-9, dz = (double)z;
    for (int it = 0; it < iters; ++it) {
        if constexpr (K == 0) { BODY2("v_add_u32") }
        else if constexpr (K == 1) { BODY2("v_mul_lo_u32") }
        else if constexpr (K == 2) { BODY2("v_mul_hi_u32") }
        else if constexpr (K == 3) { BODY2("v_mul_u32_u24") }
        else if constexpr (K == 4) { BODY2("v_mul_hi_u32_u24") }
        else if constexpr (K == 5) { BODY3("v_mad_u32_u24") }
        else if constexpr (K == 6) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x64[c]) : "v"(y), "v"(z) : "vcc"); }
        }
        else if constexpr (K == 7) { BODY2("v_min_u32") }
        else if constexpr (K == 8) { BODY3("v_add3_u32") }
        else if constexpr (K == 9) {
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(dy), "v"(dz)); }
        }
        else if constexpr (K == 10) { BODY2("v_mul_lo_u32") BODY2("v_add_u32") BODY2("v_min_u32") BODY2("v_sub_u32") }
        else if constexpr (K == 11) { BODY2("v_sub_u32") }
        else if constexpr (K == 12) { BODY3("v_mad_i32_i24") }
        else if constexpr (K == 13) {
            // SGPR second operand (uniform twiddle): does a scalar source change the rate?
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "s"(y)); }
        }
    }
    uint32_t acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c] ^ (uint32_t)x64[c] ^ (uint32_t)d[c];
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int K>
static void run(const char *name, int instr_per_body, uint32_t *dout)
{
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    probe<K><<<blocks, threads>>>(dout, 10, 3u, 5u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        probe<K><<<blocks, threads>>>(dout, iters, 3u, 5u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double winstr = (double)blocks * (threads / 64) * iters * UNROLL * CHAINS * instr_per_body;
    double lane_ops = winstr * 64;
    double sec = best * 1e-3;
    double cyc_per_winstr_per_simd = sec * 2.4e9 * (256 * 4) / winstr;
    printf("%-28s %8.3f ms  %8.2f Tlane-op/s  %6.2f cyc/wave-instr/SIMD @2.4GHz\n", name, best,
           lane_ops / sec * 1e-12, cyc_per_winstr_per_simd);
}

int main()
{
    uint32_t *dout;
    CK(hipMalloc(&dout, 256 * 8 * 256 * 4));
    run<0>("v_add_u32", 1, dout);
    run<11>("v_sub_u32", 1, dout);
    run<7>("v_min_u32", 1, dout);
    run<8>("v_add3_u32", 1, dout);
    run<1>("v_mul_lo_u32", 1, dout);
    run<2>("v_mul_hi_u32", 1, dout);
    run<13>("v_mul_hi_u32 (sgpr src)", 1, dout);
    run<3>("v_mul_u32_u24", 1, dout);
    run<4>("v_mul_hi_u32_u24", 1, dout);
    run<5>("v_mad_u32_u24", 1, dout);
    run<12>("v_mad_i32_i24", 1, dout);
    run<6>("v_mad_u64_u32", 1, dout);
    run<9>("v_fma_f64", 1, dout);
    run<10>("mix mul_lo+add+min+sub", 4, dout);
    CK(hipFree(dout));
    return 0;
}
