// stream.cpp -- products whose operands live in HOST memory, streamed through the GPU: slices of the
// batch move host -> device, through the multiply and back on three streams, so the PCIe transfers of
// one slice overlap the kernels of another (the loader side of the path).
//
// The reference harness hands poly_multiply host arrays (main-montgomery.cc:297, :337); the one-shot
// entry points of this library copy them in, multiply and copy out, serially and through pageable
// memory.  ntt_poly_multiply_stream keeps that calling convention for any number of products:
//   * pinned host memory (ntt_host_alloc, or anything the HIP runtime reports as host memory): the
//     DMA engines read and write the caller's arrays directly;
//   * pageable memory: slices are staged through pinned bounce buffers by a few copy threads.
// A ring of three device slots carries the slices; events order copy-in -> multiply -> copy-out per
// slot and keep a slot from being refilled before its previous occupant has left.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/ntt_amd.h"
#include "engine.hpp"

using namespace ntt;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

namespace ntt {

constexpr int kRing = 3;

// joins on scope exit, so that an early return or an exception never destroys a running thread
struct JoinGuard {
    std::thread t;
    ~JoinGuard()
    {
        if (t.joinable()) t.join();
    }
};

struct StreamState {
    hipStream_t h2d = nullptr, d2h = nullptr;
    bool own_d2h = false;
    hipEvent_t in_done[kRing] = {}, mul_done[kRing] = {}, out_done[kRing] = {};
    DeviceBuffer da[kRing], db[kRing], dout[kRing];
    PinnedBuffer ha[kRing], hb[kRing], hout[kRing];
};

void stream_release(ntt_ctx *ctx)
{
    StreamState *st = ctx->stream_state;
    if (!st) return;
    for (int k = 0; k < kRing; ++k) {
        if (st->in_done[k]) (void)hipEventDestroy(st->in_done[k]);
        if (st->mul_done[k]) (void)hipEventDestroy(st->mul_done[k]);
        if (st->out_done[k]) (void)hipEventDestroy(st->out_done[k]);
        st->da[k].release(); st->db[k].release(); st->dout[k].release();
        st->ha[k].release(); st->hb[k].release(); st->hout[k].release();
    }
    if (st->h2d) (void)hipStreamDestroy(st->h2d);
    if (st->d2h && st->own_d2h) (void)hipStreamDestroy(st->d2h);
    delete st;
    ctx->stream_state = nullptr;
}

static int stream_state_create(ntt_ctx *ctx)
{
    auto *st = new StreamState();
    ctx->stream_state = st;
    HIP_TRY(hipStreamCreateWithFlags(&st->h2d, hipStreamNonBlocking));
    // The runtime has four hardware queues per process (GPU_MAX_HW_QUEUES); a fifth stream shares
    // one with another, and copy-out then serialises against copy-in or the kernels (measured: 28k
    // vs 41k products/s).  The streamed multiply runs on one compute lane, so the context's second
    // lane stream is free to carry the copy-out.
    if (ctx->nlanes > 1 && ctx->lane[1]) {
        st->d2h = ctx->lane[1];
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&st->d2h, hipStreamNonBlocking));
        st->own_d2h = true;
    }
    for (int k = 0; k < kRing; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&st->in_done[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&st->mul_done[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&st->out_done[k], hipEventDisableTiming));
    }
    return NTT_OK;
}

static int stream_state(ntt_ctx *ctx, StreamState **out)
{
    if (!ctx->stream_state) {
        const int rc = stream_state_create(ctx);
        if (rc) {                          // never leave a half-built ring behind for the next call
            stream_release(ctx);
            return rc;
        }
    }
    *out = ctx->stream_state;
    return NTT_OK;
}

// pinned (page-locked) host memory the DMA engines can address directly?
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at{};
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// rows x row_bytes from src (pitch spitch) to dst (pitch dpitch) on up to 8 threads
static void parallel_copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t row_bytes, size_t rows)
{
    if (rows == 0 || row_bytes == 0) return;
    if (dpitch == row_bytes && spitch == row_bytes) {          // contiguous: split by bytes
        row_bytes *= rows;
        rows = 1;
    }
    constexpr size_t kMinPerThread = 2u << 20;
    const size_t total = row_bytes * rows;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, hw), std::max<size_t>(1, total / kMinPerThread));
    auto work = [=](unsigned t) {
        if (rows == 1) {
            const size_t per = (total / nt + 63) & ~(size_t)63;
            const size_t off = std::min(total, (size_t)t * per), len = std::min(per, total - off);
            if (len) std::memcpy((char *)dst + off, (const char *)src + off, len);
        } else {
            for (size_t r = t; r < rows; r += nt)
                std::memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, row_bytes);
        }
    };
    if (nt <= 1) {
        work(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    unsigned started = 1;                  // share 0 is this thread's
    try {
        for (; started < nt; ++started) th.emplace_back(work, started);
    } catch (...) {                        // no more threads to be had: the remaining shares run here
    }
    work(0);
    for (unsigned t = started; t < nt; ++t) work(t);
    for (auto &x : th) x.join();
}
static void parallel_copy(void *dst, const void *src, size_t bytes) { parallel_copy2d(dst, bytes, src, bytes, bytes, 1); }

}  // namespace ntt

extern "C" int ntt_host_alloc(size_t bytes, void **ptr)
try {
    if (!ptr || bytes == 0) return fail(NTT_ERR_INVALID, "ntt_host_alloc: bad argument");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *ptr = nullptr;
        return fail(e == hipErrorOutOfMemory ? NTT_ERR_NOMEM : NTT_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return NTT_OK;
}
NTT_CATCH_STATUS

extern "C" int ntt_host_free(void *ptr)
try {
    if (!ptr) return NTT_OK;
    HIP_TRY(hipHostFree(ptr));
    return NTT_OK;
}
NTT_CATCH_STATUS

extern "C" int ntt_poly_multiply_stream(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                                        size_t ab_stride, size_t count)
try {
    if (!pl || !a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_stream: null argument");
    if (!pl->shard.empty()) return fail(NTT_ERR_UNSUPPORTED, "ntt_poly_multiply_stream: not for plans of a multi-device context (stream per slot: ntt_ctx_device)");
    if (count == 0) return NTT_OK;
    if (pl->negacyclic) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_stream: not for negacyclic plans");
    const size_t n = (size_t)pl->n, out_len = 2 * n - 1;
    if (ab_stride < out_len) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_stream: ab_stride < 2n-1");
    if (is_device_ptr(a) || is_device_ptr(b) || is_device_ptr(ab))
        return fail(NTT_ERR_INVALID, "ntt_poly_multiply_stream: operands must be host memory (device arrays: ntt_poly_multiply_batched)");
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    StreamState *st = nullptr;
    int rc = stream_state(cx, &st);
    if (rc) return rc;
    // The multiply is far from being the bottleneck here (PCIe is): one compute lane for the duration
    // of the call, the second lane's stream carries the copy-out (see stream_state).
    struct LaneGuard {
        ntt_ctx *c;
        int saved;
        explicit LaneGuard(ntt_ctx *cx_) : c(cx_), saved(cx_->lane_limit) { c->lane_limit = 1; }
        ~LaneGuard() { c->lane_limit = saved; }
    } lane_guard(cx);
    const size_t S = (size_t)pl->batch;                       // products per slice = plan capacity
    const bool pin_a = is_pinned_host(a), pin_b = is_pinned_host(b), pin_o = is_pinned_host(ab);
    for (int k = 0; k < kRing; ++k) {
        if ((rc = st->da[k].ensure(S * n * 4)) || (rc = st->db[k].ensure(S * n * 4)) ||
            (rc = st->dout[k].ensure(S * out_len * 4)))
            return rc;
        if (!pin_a && (rc = st->ha[k].ensure(S * n * 4))) return rc;
        if (!pin_b && (rc = st->hb[k].ensure(S * n * 4))) return rc;
        if (!pin_o && (rc = st->hout[k].ensure(S * out_len * 4))) return rc;
    }
    const size_t slices = (count + S - 1) / S;
    auto cnt_of = [&](size_t s) { return std::min(S, count - s * S); };
    // pageable output: copy slice s from its bounce buffer into the caller's rows (entries beyond
    // 2n-1 of a row are not ours and stay untouched)
    auto finalize = [&](size_t s) -> int {
        if (pin_o) return NTT_OK;
        const int k = (int)(s % kRing);
        HIP_TRY(hipEventSynchronize(st->out_done[k]));
        const uint32_t *src = static_cast<const uint32_t *>(st->hout[k].ptr);
        uint32_t *dst = ab + s * S * ab_stride;
        const size_t cnt = cnt_of(s);
        parallel_copy2d(dst, ab_stride * 4, src, out_len * 4, out_len * 4, cnt);
        return NTT_OK;
    };
    // the compute stream may still hold work of earlier calls that reads or writes nothing of ours;
    // the ring's own events are from completed calls (every call drains its slices before it returns)
    auto issue_in = [&](size_t s) -> int {
        const int k = (int)(s % kRing);
        const size_t in_bytes = cnt_of(s) * n * 4;
        if (s >= (size_t)kRing) {
            HIP_TRY(hipStreamWaitEvent(st->h2d, st->mul_done[k], 0));          // da[k], db[k] consumed
            if (!pin_a || !pin_b) HIP_TRY(hipEventSynchronize(st->in_done[k])); // ha[k], hb[k] consumed
        }
        const uint32_t *sa = a + s * S * n, *sb = b + s * S * n;
        if (!pin_a) {
            parallel_copy(st->ha[k].ptr, sa, in_bytes);
            sa = static_cast<const uint32_t *>(st->ha[k].ptr);
        }
        HIP_TRY(hipMemcpyAsync(st->da[k].ptr, sa, in_bytes, hipMemcpyHostToDevice, st->h2d));
        if (!pin_b) {
            parallel_copy(st->hb[k].ptr, sb, in_bytes);
            sb = static_cast<const uint32_t *>(st->hb[k].ptr);
        }
        HIP_TRY(hipMemcpyAsync(st->db[k].ptr, sb, in_bytes, hipMemcpyHostToDevice, st->h2d));
        HIP_TRY(hipEventRecord(st->in_done[k], st->h2d));
        return NTT_OK;
    };
    // The copy-in of slice s + 1 is issued BEFORE the multiply and copy-out of slice s: measured
    // (rocprofv3 copy trace, profiles/), a device-to-host copy call can hold the calling thread until the
    // copy has run, and a copy-in issued after it starts only then.
    auto run = [&]() -> int {
        int rc;
        if ((rc = issue_in(0))) return rc;
        for (size_t s = 0; s < slices; ++s) {
            const int k = (int)(s % kRing);
            const size_t cnt = cnt_of(s);
            // pageable memory: the result of slice s - kRing leaves its bounce buffer on a helper thread
            // while this thread stages the next copy-in
            int frc = NTT_OK;
            {
                JoinGuard fin;
                if (s >= (size_t)kRing && !pin_o) {
                    try {
                        fin.t = std::thread([&, s] { frc = finalize(s - kRing); });
                    } catch (const std::system_error &) {                           // no thread: do it here
                        frc = finalize(s - kRing);
                    }
                }
                if (s + 1 < slices) rc = issue_in(s + 1);
            }                                                                      // joined: hout[k] is free again
            if (rc) return rc;
            if (frc) return frc;
            // multiply on the context's stream once the slice has landed and the slot's previous result
            // has left
            HIP_TRY(hipStreamWaitEvent(cx->stream, st->in_done[k], 0));
            if (s >= (size_t)kRing) HIP_TRY(hipStreamWaitEvent(cx->stream, st->out_done[k], 0));
            rc = multiply_device_any(pl, static_cast<const uint32_t *>(st->da[k].ptr),
                                     static_cast<const uint32_t *>(st->db[k].ptr),
                                     static_cast<uint32_t *>(st->dout[k].ptr), out_len, (int)cnt);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(st->mul_done[k], cx->stream));
            HIP_TRY(hipStreamWaitEvent(st->d2h, st->mul_done[k], 0));
            if (pin_o && ab_stride == out_len) {
                HIP_TRY(hipMemcpyAsync(ab + s * S * ab_stride, st->dout[k].ptr, cnt * out_len * 4, hipMemcpyDeviceToHost, st->d2h));
            } else if (pin_o) {
                HIP_TRY(hipMemcpy2DAsync(ab + s * S * ab_stride, ab_stride * 4, st->dout[k].ptr, out_len * 4, out_len * 4,
                                         cnt, hipMemcpyDeviceToHost, st->d2h));
            } else {
                HIP_TRY(hipMemcpyAsync(st->hout[k].ptr, st->dout[k].ptr, cnt * out_len * 4, hipMemcpyDeviceToHost, st->d2h));
            }
            HIP_TRY(hipEventRecord(st->out_done[k], st->d2h));
        }
        for (size_t s = slices > (size_t)kRing ? slices - kRing : 0; s < slices; ++s)
            if ((rc = finalize(s))) return rc;
        HIP_TRY(hipStreamSynchronize(st->d2h));
        HIP_TRY(hipStreamSynchronize(st->h2d));
        HIP_TRY(hipStreamSynchronize(cx->stream));
        return NTT_OK;
    };
    // Whatever ends the call -- a status or an exception -- nothing of it may still be in flight afterwards: the
    // copies read and write the caller's arrays, and the next call relies on a drained ring.
    struct Drain {
        StreamState *st;
        ntt_ctx *cx;
        bool armed = true;
        ~Drain()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(st->h2d);
            (void)hipStreamSynchronize(cx->stream);
            (void)hipStreamSynchronize(st->d2h);
        }
    } drain{st, cx};
    rc = run();
    if (rc == NTT_OK) drain.armed = false;
    return rc;
}
NTT_CATCH_STATUS
