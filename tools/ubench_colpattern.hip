// ubench_colpattern.hip -- what memory rate can the column-pass ACCESS PATTERN reach on MI355X?
// No arithmetic: read the lower 32 rows of a 64 x 4096 u32 matrix per operand (zero-padded input),
// write all 64 rows (stride 16 KiB between rows), batched like col_fwd (128 products x 2 operands).
//   A: one column per thread, dword accesses     (256 contiguous bytes per wave instruction)
//   B: four columns per thread, dwordx4 accesses (1 KiB contiguous per wave instruction)
//   C: same byte counts, fully contiguous dwordx4 streaming (upper bound for this r:w mix)
//   D: like A but 8 workgroup-columns interleaved so that concurrently running waves of one
//      workgroup cover a 2 KiB contiguous span of each row
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_colpattern.hip -o tools/ubench_colpattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 64, S = 4096;

__global__ void __launch_bounds__(256) patA(const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *pi = in + (size_t)blockIdx.y * (ROWS / 2) * S + c;
    uint32_t *po = out + (size_t)blockIdx.y * ROWS * S + c;
    uint32_t x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = pi[(size_t)i * S];
#pragma unroll
    for (int i = 0; i < 32; ++i) { po[(size_t)i * S] = x[i] + 1; po[(size_t)(i + 32) * S] = x[i] ^ 5; }
}

__global__ void __launch_bounds__(256) patB(const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 4;
    const uint32_t *pi = in + (size_t)blockIdx.y * (ROWS / 2) * S + c;
    uint32_t *po = out + (size_t)blockIdx.y * ROWS * S + c;
    uint4 x[8];
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const uint4 *>(pi + (size_t)(g * 8 + i) * S);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4 v = x[i];
            v.x += 1;
            *reinterpret_cast<uint4 *>(po + (size_t)(g * 8 + i) * S) = v;
            v.y ^= 5;
            *reinterpret_cast<uint4 *>(po + (size_t)(g * 8 + i + 32) * S) = v;
        }
    }
}

__global__ void __launch_bounds__(256) patC(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n_in)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_in) {
        uint4 v = in[i];
        v.x += 1;
        out[2 * i] = v;          // two writes per read, contiguous pairs of 16 B... keep simple
        v.y ^= 5;
        out[2 * i + 1] = v;
    }
}

template <class F>
static float timeit(F f)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int B = 256;                                   // 128 products x 2 operands
    const size_t in_elems = (size_t)B * (ROWS / 2) * S, out_elems = (size_t)B * ROWS * S;
    uint32_t *in, *out;
    CK(hipMalloc(&in, in_elems * 4));
    CK(hipMalloc(&out, out_elems * 4));
    CK(hipMemset(in, 1, in_elems * 4));
    const double bytes = (in_elems + out_elems) * 4.0;
    float a = timeit([&] { patA<<<dim3(S / 256, B), 256>>>(in, out); });
    float b = timeit([&] { patB<<<dim3(S / 1024, B), 256>>>(in, out); });
    float c = timeit([&] { patC<<<dim3((unsigned)((in_elems / 4 + 255) / 256)), 256>>>((const uint4 *)in, (uint4 *)out, in_elems / 4); });
    printf("bytes per launch: %.1f MiB (read %.1f + write %.1f)\n", bytes / 1048576, in_elems * 4.0 / 1048576, out_elems * 4.0 / 1048576);
    printf("A dword, 256 B segments, stride 16 KiB : %8.3f ms  %7.1f GB/s\n", a, bytes / a * 1e-6);
    printf("B dwordx4, 1 KiB segments              : %8.3f ms  %7.1f GB/s\n", b, bytes / b * 1e-6);
    printf("C contiguous dwordx4 stream            : %8.3f ms  %7.1f GB/s\n", c, bytes / c * 1e-6);
    return 0;
}
