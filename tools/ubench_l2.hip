// ubench_l2.hip -- does data written by one kernel stay readable from the producing XCD's L2 in the
// next kernel?  Kernel W: block b writes chunk (b/8) of region (b%8).  Kernel R: block b reads chunk
// (b/8) of region ((b+shift)%8).  shift = 0: same XCD as the writer (if blocks are dealt b%8 over
// XCDs, which the XCC_ID register confirms), shift = 1: a different XCD.  Timed for several sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kChunkWords = 4096;      // 16 KiB per block

__global__ void __launch_bounds__(256) wkernel(uint32_t *buf, size_t region_words, uint32_t tag, uint32_t *xcc)
{
    const uint32_t b = blockIdx.x;
    uint4 *p = reinterpret_cast<uint4 *>(buf + (size_t)(b & 7) * region_words + (size_t)(b >> 3) * kChunkWords);
    for (int i = threadIdx.x; i < kChunkWords / 4; i += 256) p[i] = make_uint4(tag + i, b, tag, i);
    if (threadIdx.x == 0 && xcc) xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}

__global__ void __launch_bounds__(256) rkernel(const uint32_t *buf, size_t region_words, uint32_t shift, uint32_t *sink)
{
    const uint32_t b = blockIdx.x;
    const uint4 *p = reinterpret_cast<const uint4 *>(buf + (size_t)((b + shift) & 7) * region_words + (size_t)(b >> 3) * kChunkWords);
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < kChunkWords / 4; i += 256) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    uint32_t *buf, *sink, *xcc;
    const size_t max_bytes = 512ull << 20;
    CK(hipMalloc(&buf, max_bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&xcc, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // XCC mapping check
    {
        const int blocks = 4096;
        hipLaunchKernelGGL(wkernel, dim3(blocks), dim3(256), 0, 0, buf, (size_t)(blocks / 8) * kChunkWords, 1u, xcc);
        std::vector<uint32_t> h(blocks);
        CK(hipMemcpy(h.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
        int same = 0; int hist[16] = {0};
        for (int b = 0; b < blocks; ++b) { hist[h[b] & 15]++; if (h[b] == h[b & 7]) same++; }
        printf("XCC_ID of blocks 0..7:"); for (int b = 0; b < 8; ++b) printf(" %u", h[b]);
        printf("\nblocks with XCC_ID == XCC_ID(block b%%8): %d of %d; histogram:", same, blocks);
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("\n");
    }
    for (size_t mib : {4, 8, 16, 24, 32, 48, 64, 128, 256}) {
        const size_t bytes = mib << 20;
        const int blocks = (int)(bytes / (kChunkWords * 4));
        const size_t region_words = (size_t)(blocks / 8) * kChunkWords;
        for (uint32_t shift : {0u, 1u, 4u}) {
            float tw = 0, tr = 0, tr2 = 0;
            const int reps = 20;
            for (int r = 0; r < reps + 2; ++r) {
                float ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(wkernel, dim3(blocks), dim3(256), 0, 0, buf, region_words, (uint32_t)r, (uint32_t *)nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tw += ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(rkernel, dim3(blocks), dim3(256), 0, 0, buf, region_words, shift, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tr += ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(rkernel, dim3(blocks), dim3(256), 0, 0, buf, region_words, shift, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tr2 += ms;
            }
            printf("%4zu MiB shift %u: write %7.1f us (%6.0f GB/s)  read-after-write %7.1f us (%6.0f GB/s)  read-again %7.1f us (%6.0f GB/s)\n",
                   mib, shift, tw / reps * 1e3, bytes / (tw / reps * 1e-3) / 1e9, tr / reps * 1e3,
                   bytes / (tr / reps * 1e-3) / 1e9, tr2 / reps * 1e3, bytes / (tr2 / reps * 1e-3) / 1e9);
        }
    }
    return 0;
}
