// Host-visible completion latency of a short kernel chain on gfx950: what the caller of one synchronised product pays
// around the kernels.  Variants: stream synchronise; event recorded behind the chain + event synchronise; the last
// launch carrying its own completion event (hipExtLaunchKernelGGL stopEvent) + event synchronise; a flag in pinned host
// memory written by the last kernel and polled by the host.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench_sync.hip -o tools/ubench_sync
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(unsigned *p, unsigned v) { p[blockIdx.x * blockDim.x + threadIdx.x] += v; }
__global__ void touch_flag(unsigned *p, unsigned v, volatile unsigned *flag, unsigned *count, unsigned nblocks) {
    p[blockIdx.x * blockDim.x + threadIdx.x] += v;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        if (atomicAdd(count, 1u) == nblocks - 1) { *count = 0; __threadfence_system(); *flag = v; }
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int chain = argc > 1 ? atoi(argv[1]) : 1, blocks = argc > 2 ? atoi(argv[2]) : 64, reps = 2000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *d, *count, *flag;
    CK(hipMalloc(&d, blocks * 256 * 4)); CK(hipMemset(d, 0, blocks * 256 * 4));
    CK(hipMalloc(&count, 4)); CK(hipMemset(count, 0, 4));
    CK(hipHostMalloc(&flag, 4, hipHostMallocMapped)); *flag = 0;
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    for (int variant = 0; variant < 4; ++variant) {
        double best = 1e9;
        for (int round = 0; round < 3; ++round) {
            double t0 = now();
            for (int r = 1; r <= reps; ++r) {
                for (int k = 0; k < chain - 1; ++k) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, d, 1u);
                if (variant == 0) {
                    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, d, 1u);
                    CK(hipStreamSynchronize(st));
                } else if (variant == 1) {
                    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, d, 1u);
                    CK(hipEventRecord(ev, st));
                    CK(hipEventSynchronize(ev));
                } else if (variant == 2) {
                    hipExtLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, nullptr, ev, 0, d, 1u);
                    CK(hipEventSynchronize(ev));
                } else {
                    const unsigned tag = (unsigned)(round * reps + r);
                    hipLaunchKernelGGL(touch_flag, dim3(blocks), dim3(256), 0, st, d, tag, flag, count, (unsigned)blocks);
                    while (*(volatile unsigned *)flag != tag) { }
                }
            }
            CK(hipStreamSynchronize(st));
            double dt = (now() - t0) / reps * 1e6;
            if (dt < best) best = dt;
        }
        static const char *names[] = {"stream synchronise", "event record + event synchronise", "stopEvent on the last launch + event synchronise", "host flag polled"};
        printf("chain %d x %d blocks  %-52s %.2f us per round\n", chain, blocks, names[variant], best);
    }
    CK(hipGetLastError());
    return 0;
}
