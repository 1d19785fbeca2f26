// ubench_clock.hip -- does memory traffic beside an arithmetic loop pull the shader clock down?
// Stream A: the butterfly-mix loop of ubench_mix.hip (registers only), 5 workgroups of 4 waves per CU.
// Stream B (optional): a streaming copy over buffers far larger than the Infinity Cache (HBM traffic), or over a
// 64 MiB window (Infinity Cache traffic), or an LDS exchange loop, running the whole time.
// Prints the mix loop's rate and in-kernel clock alone and beside each kind of traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct P { uint32_t p, p2, nprime, w; };

__global__ void __launch_bounds__(256) mix(uint32_t *out, uint64_t *clk, P c, int iters)
{
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    const uint64_t C = (uint64_t)c.p2 * c.w;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int half = 8 >> s;
#pragma unroll
            for (int b = 0; b < 16; b += 2 * half)
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    uint32_t u = x[b + j], v = x[b + j + half];
                    uint32_t t, r, sum = u + v;
                    asm volatile("v_subrev_co_u32 %0, vcc, %2, %3\n\tv_cndmask_b32 %1, %0, %3, vcc"
                                 : "=&v"(t), "=v"(r) : "s"(c.p2), "v"(sum) : "vcc");
                    x[b + j] = r;
                    uint64_t T, carry;
                    asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(carry) : "v"(u - v), "v"(c.w + (uint32_t)j), "s"(C));
                    const uint32_t q = (uint32_t)T * c.nprime;
                    x[b + j + half] = (uint32_t)((T + (uint64_t)q * c.p) >> 32);
                }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= x[i];
    if (acc == 0x12345u) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x < 64) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(256) copy(const uint4 *in, uint4 *out, size_t n, int rounds)
{
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            uint4 v = in[i];
            v.x += r;
            out[i] = v;
        }
}

__global__ void __launch_bounds__(256) ldsloop(uint32_t *out, int iters)
{
    __shared__ uint32_t tile[4352];
    uint32_t x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[j * 256 + threadIdx.x + ((j * 256 + threadIdx.x) >> 8 << 4)] = x[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] += tile[(threadIdx.x >> 4) * 272 + j * 16 + (threadIdx.x & 15)];
        __syncthreads();
    }
    uint32_t acc = 0;
    for (int j = 0; j < 16; ++j) acc ^= x[j];
    if (acc == 0x1234567u) out[0] = acc;
}

int main()
{
    uint32_t *out;
    uint64_t *clk, h[128];
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&clk, sizeof(h)));
    const size_t big = (size_t)1 << 30;                       // 1 GiB in, 1 GiB out
    uint4 *in, *dst;
    CK(hipMalloc(&in, big));
    CK(hipMalloc(&dst, big));
    CK(hipMemset(in, 1, big));
    const P c{998244353u, 2 * 998244353u, 998244351u, 299560064u};
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 5, iters = 4000;
    const char *names[5] = {"alone", "beside an HBM copy (2 GiB per pass)", "beside an Infinity-Cache copy (64 MiB window)", "beside an LDS exchange loop", "beside an HBM copy AND an LDS exchange loop"};
    hipStream_t sc;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    for (int mode = 0; mode < 5; ++mode) {
        CK(hipDeviceSynchronize());
        // background traffic for ~3 s
        hipEvent_t b0, b1;
        CK(hipEventCreate(&b0));
        CK(hipEventCreate(&b1));
        CK(hipEventRecord(b0, sb));
        if (mode == 1 || mode == 4) copy<<<512, 256, 0, sb>>>(in, dst, big / 16, 8000);
        if (mode == 2) copy<<<512, 256, 0, sb>>>(in, dst, ((size_t)32 << 20) / 16, 500000);
        if (mode == 3) ldsloop<<<256 * 2, 256, 0, sb>>>(out, 12000000);
        if (mode == 4) ldsloop<<<256 * 2, 256, 0, sc>>>(out, 12000000);
        CK(hipEventRecord(b1, sb));
        for (int w = 0; w < 400; ++w) mix<<<blocks, 256, 0, sa>>>(out, clk, c, iters);   // ~3 s: the power controller settles in ~0.5 s
        CK(hipStreamSynchronize(sa));
        CK(hipEventRecord(e0, sa));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) mix<<<blocks, 256, 0, sa>>>(out, clk, c, iters);
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        const bool bg_running = mode == 0 || hipStreamQuery(sb) == hipErrorNotReady;
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        double ghz = 0;
        for (int i = 0; i < 64; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
        ghz /= 64;
        const double lane_ops = (double)reps * blocks * 4 * iters * 224.0 * 64;
        printf("mix loop %-46s: %5.1f T lane-op/s, in-kernel clock %.2f GHz%s\n", names[mode], lane_ops / (ms * 1e-3) / 1e12, ghz,
               bg_running ? "" : "  (background finished early)");
        fflush(stdout);
        if (mode == 1 || mode == 2 || mode == 4) {             // what the background copy moved (it outlives the mix loop)
            CK(hipEventSynchronize(b1));
            float bms;
            CK(hipEventElapsedTime(&bms, b0, b1));
            const double bytes = mode != 2 ? 8000.0 * 2 * big : 500000.0 * 2 * ((size_t)32 << 20);
            printf("    background copy: %.2f TB/s over %.1f s (the mix loop ran for the first %.1f s of it)\n", bytes / (bms * 1e-3) / 1e12, bms * 1e-3,
                   410 * ms / reps * 1e-3);
        }
    }
    CK(hipDeviceSynchronize());                               // every background kernel is bounded and drains here
    return 0;
}
