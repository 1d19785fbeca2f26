// What moving ONE product's operands costs on the host side (n = 2^17: a, b 512 KiB each in, ab 1 MiB out), the parts
// of poly_multiply on pageable arrays: hipMemcpy from / to pageable memory, hipMemcpy from / to pinned memory, and a
// plain CPU memcpy between pageable and pinned memory (one thread, two threads).
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench_stage.hip -o tools/ubench_stage -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double best_us(F f, int reps = 200)
{
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        double t0 = now();
        for (int i = 0; i < reps; ++i) f(i);
        best = std::min(best, (now() - t0) / reps * 1e6);
    }
    return best;
}

int main()
{
    const size_t in_bytes = 512 << 10, out_bytes = 1 << 20;
    // several pageable buffers so that the source is not always hot in the cache of this core
    std::vector<std::vector<char>> pa(8, std::vector<char>(in_bytes, 1)), pb(8, std::vector<char>(in_bytes, 2)), pout(8, std::vector<char>(out_bytes, 3));
    char *ha, *hb, *hout;
    CK(hipHostMalloc(&ha, in_bytes)); CK(hipHostMalloc(&hb, in_bytes)); CK(hipHostMalloc(&hout, out_bytes));
    char *da, *db, *dout;
    CK(hipMalloc(&da, in_bytes)); CK(hipMalloc(&db, in_bytes)); CK(hipMalloc(&dout, out_bytes));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("pageable -> device, a and b (2 x 512 KiB), async + sync: %.1f us\n", best_us([&](int i) {
        CK(hipMemcpyAsync(da, pa[i & 7].data(), in_bytes, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(db, pb[i & 7].data(), in_bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st)); }));
    printf("pinned   -> device, a and b, async + sync:                %.1f us\n", best_us([&](int) {
        CK(hipMemcpyAsync(da, ha, in_bytes, hipMemcpyHostToDevice, st));
        CK(hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st)); }));
    printf("device -> pageable, ab (1 MiB), async + sync:             %.1f us\n", best_us([&](int i) {
        CK(hipMemcpyAsync(pout[i & 7].data(), dout, out_bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st)); }));
    printf("device -> pinned, ab, async + sync:                       %.1f us\n", best_us([&](int) {
        CK(hipMemcpyAsync(hout, dout, out_bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st)); }));
    printf("CPU memcpy pageable -> pinned, a and b, one thread:       %.1f us\n", best_us([&](int i) {
        memcpy(ha, pa[i & 7].data(), in_bytes); memcpy(hb, pb[i & 7].data(), in_bytes); }));
    printf("CPU memcpy pinned -> pageable, ab, one thread:            %.1f us\n", best_us([&](int i) {
        memcpy(pout[i & 7].data(), hout, out_bytes); }));
    printf("CPU memcpy pageable -> pinned, a and b, a thread each (spawned per call): %.1f us\n", best_us([&](int i) {
        std::thread t([&] { memcpy(hb, pb[i & 7].data(), in_bytes); });
        memcpy(ha, pa[i & 7].data(), in_bytes);
        t.join(); }, 100));
    return 0;
}
