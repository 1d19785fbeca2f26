// ubench_pipeline_mem.hip -- the memory-side floor of the three-sweep product pipeline on MI355X.
// No arithmetic: three kernels that move exactly the bytes of col_fwd / row_mul / col_inv with the same
// access patterns and the same hand-off buffers (intermediates re-read by the next kernel, so whatever the
// Infinity Cache keeps is kept here too), over 1024 products of L = 2^18 in groups of `chunk` on 1 or 2 streams.
//   K1: read the lower 32 rows of a 64 x 4096 matrix per operand (the caller's a, b: n = L/2 values, streaming
//       loads), write all 64 rows of both work buffers (256 B per wave instruction, stride 16 KiB)
//   K2: one workgroup per 4096-point row: read the row of both work buffers (coalesced), write one
//   K3: read 64 rows of the work buffer (column pattern), write the caller's ab (2n - 1 values, streaming stores)
// Prints us per product: the time the same traffic costs when nothing is computed.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_pipeline_mem.hip -o tools/ubench_pipeline_mem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 64, S = 4096, L = ROWS * S, N = L / 2;

__global__ void __launch_bounds__(256) k1(const uint32_t *a, const uint32_t *b, uint32_t *w0, uint32_t *w1)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *pi = (blockIdx.z ? b : a) + (size_t)blockIdx.y * N + c;
    uint32_t *po = (blockIdx.z ? w1 : w0) + (size_t)blockIdx.y * L + c;
    uint32_t x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = __builtin_nontemporal_load(pi + (size_t)i * S);
#pragma unroll
    for (int i = 0; i < 32; ++i) { po[(size_t)i * S] = x[i] + 1; po[(size_t)(i + 32) * S] = x[i] ^ 5; }
}

__global__ void __launch_bounds__(256) k2(const uint32_t *w0, const uint32_t *w1, uint32_t *out)
{
    const size_t row = (size_t)blockIdx.x * S;
    uint32_t x[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = w0[row + threadIdx.x + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j) y[j] = w1[row + threadIdx.x + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j) out[row + threadIdx.x + 256 * j] = x[j] + y[j];
}

__global__ void __launch_bounds__(256) k3(const uint32_t *w0, uint32_t *ab)
{
    const uint32_t c = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t r = threadIdx.x >> 6;                       // 4 waves: 16 rows each (the col16 kernel's shape)
    const uint32_t *pi = w0 + (size_t)blockIdx.y * L + c;
    uint32_t *po = ab + (size_t)blockIdx.y * L + c;
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = pi[(size_t)(4 * i + r) * S];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((4 * i + r) * S + c < 2 * N - 1) __builtin_nontemporal_store(x[i] + 3, po + (size_t)(4 * i + r) * S);
}

int main(int argc, char **argv)
{
    const int B = 1024;
    const int chunk = argc > 1 ? atoi(argv[1]) : 48;
    const int lanes = argc > 2 ? atoi(argv[2]) : 2;
    uint32_t *a, *b, *ab, *w0[2], *w1[2];
    CK(hipMalloc(&a, (size_t)B * N * 4));
    CK(hipMalloc(&b, (size_t)B * N * 4));
    CK(hipMalloc(&ab, (size_t)B * L * 4));
    for (int l = 0; l < 2; ++l) {
        CK(hipMalloc(&w0[l], (size_t)chunk * L * 4));
        CK(hipMalloc(&w1[l], (size_t)chunk * L * 4));
    }
    CK(hipMemset(a, 1, (size_t)B * N * 4));
    CK(hipMemset(b, 2, (size_t)B * N * 4));
    hipStream_t st[2];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto step = [&] {
        int ci = 0;
        for (int c0 = 0; c0 < B; c0 += chunk, ++ci) {
            const int l = ci % lanes, cb = std::min(chunk, B - c0);
            k1<<<dim3(S / 256, cb, 2), 256, 0, st[l]>>>(a + (size_t)c0 * N, b + (size_t)c0 * N, w0[l], w1[l]);
            k2<<<dim3(cb * ROWS), 256, 0, st[l]>>>(w0[l], w1[l], w0[l]);
            k3<<<dim3(S / 64, cb), 256, 0, st[l]>>>(w0[l], ab + (size_t)c0 * L);
        }
    };
    for (int w = 0; w < 3; ++w) step();
    CK(hipDeviceSynchronize());
    const int steps = 20;
    CK(hipEventRecord(e0, st[0]));
    for (int s = 0; s < steps; ++s) step();
    CK(hipDeviceSynchronize());                                // both lanes drained (host clock below)
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / (steps * B);
    const double bytes = (2.0 * N + 2.0 * L + 2.0 * L + L + L + 2.0 * N - 1) * 4;
    printf("chunk %d lanes %d: %.3f us per product of pure traffic (%.2f MB per product -> %.2f TB/s)\n", chunk, lanes, us,
           bytes / 1e6, bytes / us / 1e6);
    return 0;
}
