// ubench_xcdsync.hip -- producer/consumer through memory inside ONE persistent kernel, groups of
// workgroups separated by a spin barrier (agent-scope release/acquire).  mode 0: group = b % 8 (all
// members on one XCD, so the exchanged data can live in that XCD's L2), mode 1: group = b / W
// (members spread over all XCDs).  Each iteration: every block writes 16 KiB chunks of its group's
// region, barrier, reads the chunks another member wrote and checks them, barrier.
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to see the fabric traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kChunkWords = 4096;      // 16 KiB

#ifndef VARIANT
#define VARIANT 0
#endif
__device__ __forceinline__ bool group_barrier(unsigned *counter, unsigned target, unsigned *err)
{
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
#if VARIANT == 0
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#elif VARIANT == 1
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt only: stores have reached L2
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        int good = 0;
        for (int spin = 0; spin < (1 << 20); ++spin) {
#if VARIANT == 0
            if (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
#else
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
#endif
            __builtin_amdgcn_s_sleep(1);
        }
#if VARIANT >= 1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // one buffer_inv sc1
#endif
        if (!good) atomicAdd(err, 1u);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ void __launch_bounds__(256) k(uint32_t *buf, unsigned *counters, unsigned *err, int W, int chunks_per_block,
                                          int iters, int mode, unsigned *mism)
{
    const uint32_t b = blockIdx.x;
    const uint32_t group = mode == 0 ? (b & 7) : (b / W);
    const uint32_t j = mode == 0 ? (b >> 3) : (b % W);
    const size_t region_words = (size_t)W * chunks_per_block * kChunkWords;
    uint32_t *reg = buf + (size_t)group * region_words;
    unsigned *ctr = counters + group * 64;           // 256 B apart
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int c = 0; c < chunks_per_block; ++c) {
            uint4 *p = reinterpret_cast<uint4 *>(reg + ((size_t)j * chunks_per_block + c) * kChunkWords);
            for (int i = threadIdx.x; i < kChunkWords / 4; i += 256) p[i] = make_uint4(it, j, c, i);
        }
        if (!group_barrier(ctr, (2 * it + 1) * W, err)) return;
        const uint32_t src = (j * 7 + 3 + it) % W;     // someone else's chunks
        for (int c = 0; c < chunks_per_block; ++c) {
            const uint4 *p = reinterpret_cast<const uint4 *>(reg + ((size_t)src * chunks_per_block + c) * kChunkWords);
            for (int i = threadIdx.x; i < kChunkWords / 4; i += 256) {
                uint4 v = p[i];
                bad += (v.x != (uint32_t)it) | (v.y != src) | (v.z != (uint32_t)c) | (v.w != (uint32_t)i);
            }
        }
        if (!group_barrier(ctr, (2 * it + 2) * W, err)) return;
    }
    if (bad) atomicAdd(mism, bad);
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 2;           // blocks per CU
    const int region_kib = argc > 3 ? atoi(argv[3]) : 2048;    // per group
    const int iters = argc > 4 ? atoi(argv[4]) : 200;
    const int W = 32 * per_cu;
    const int chunks_per_block = region_kib / 16 / W;
    if (chunks_per_block < 1) { printf("region too small\n"); return 0; }
    uint32_t *buf; unsigned *ctr, *err, *mism;
    CK(hipMalloc(&buf, (size_t)8 * region_kib * 1024));
    CK(hipMalloc(&ctr, 8 * 256)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&mism, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 8 * 256)); CK(hipMemset(err, 0, 4)); CK(hipMemset(mism, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(8 * W), dim3(256), 0, 0, buf, ctr, err, W, chunks_per_block, iters, mode, mism);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr, hm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
        const double bytes = 2.0 * 8 * region_kib * 1024.0 * iters;
        printf("mode %d blocks/CU %d region %d KiB/group iters %d: %.3f ms, %.2f us/iter, %.0f GB/s (write+read), barrier timeouts %u, mismatches %u\n",
               mode, per_cu, region_kib, iters, ms, ms * 1e3 / iters, bytes / (ms * 1e-3) / 1e9, herr, hm);
    }
    return 0;
}
