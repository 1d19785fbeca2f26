// engine.hpp -- host-side engine objects behind the C ABI (include/ntt_amd.h).
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

struct ntt_plan;
struct ntt_plan64;
struct ntt_ctx;
struct ntt_dist_plan;

namespace ntt {

enum KernelId {
    K_COL_FWD = 0, K_ROW_MUL, K_COL_INV, K_ROW_FWD, K_ROW_INV, K_SMALL, K_PERMUTE, K_CRT_REDUCE,
    K_CRT_MERGE, K_COL64, K_LDS64, K_COUNT
};
extern const char *const kKernelNames[K_COUNT];

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
int exception_status() noexcept;
// function-try-block handler of every status-returning extern "C" entry point
#define NTT_CATCH_STATUS catch (...) { return ntt::exception_status(); }

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);     // grows (never shrinks); returns ntt_status
    void release();
};

// Page-locked host memory owned by a context (hipHostMalloc: mapped into the device's address space).
struct PinnedBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    void release();
};

// Buffers and per-prime plans of ntt_crt_multiply, cached per context for the last n used.
struct CrtWorkspace {
    int n = 0, batch = 0;                 // plans hold `batch` products of length n
    ntt_plan *plans[4] = {nullptr, nullptr, nullptr, nullptr};
    DeviceBuffer ra, rb, res, a64, b64, out64, merge_in;
    DeviceBuffer area;                    // work areas of the grouped batch (crt.cpp): 2 lanes x 2 operands x group x L words
};
void crt_release(ntt_ctx *ctx);
void crt_direct_release(ntt_ctx *ctx);
struct StreamState;                       // stream.cpp: ring of device slots + copy streams
void stream_release(ntt_ctx *ctx);
const char *last_error_cstr();
// NTT_OK, or NTT_ERR_INTERNAL (and the message) if a dispatcher skipped a launch since the last call; clears the mark
int take_launch_error();
bool is_device_ptr(const void *p);
bool is_pinned_host_ptr(const void *p);    // page-locked host memory of this HIP runtime (device-accessible)
int sync_stream(ntt_ctx *cx);              // engine.cpp: hipStreamSynchronize on the context's stream
bool is_prime64(uint64_t n);              // direct64.cpp: deterministic Miller-Rabin

// Where a multiply keeps the intermediates of its passes when they are not the plan's own work buffers:
// room for `chunk` products (w0 / w1: chunk * L words each); the multiply then runs on ONE stream (ctx->stream).
struct WorkArea {
    uint32_t *w0, *w1;
    int chunk;
    // true: stop in front of the outermost inverse column pass -- w0 then holds, per product, the input of that pass
    // (the CRT path runs it for four primes at once, fused with the merge: kernels_crt.hip); `ab` is not written
    bool stop_before_outer_inverse = false;
    // non-null: a and b of the multiply are the caller's u64 operands and the first forward column pass reduces them
    // modulo the plan's prime as it loads (Shoup constants of 1 and 2^32 mod p); needs a 32- or 64-point register level
    const CrtPrime *u64_in = nullptr;
};
// device-pointer entry used by the CRT path (caller holds ctx->mu); area == nullptr: the plan's buffers and lanes
int multiply_device_any(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                        size_t ab_stride, int batch, const WorkArea *area = nullptr);
// one transform of the plan's length, src -> dst (device pointers on the plan's GPU; src == dst allowed), on the context's stream
int transform_device_any(ntt_plan *pl, const uint32_t *src, uint32_t *dst, bool inverse, bool scrambled = false);

// ---- multi-device ("group") contexts: ntt_ctx_create with ndev > 1 (multidev.cpp) ----------------------------------
// (start, count) of the contiguous block of `total` units owned by slot `rank` of `world`: ceil(total / world) units per
// slot, the last slots may own fewer or none -- main-mpi.cc:657-660 (mods_per_process = ceil(4 / size)), generalised
inline void partition(long total, int world, int rank, long *start, long *count)
{
    const long per = (total + world - 1) / world;
    *start = std::min((long)rank * per, total);
    *count = std::max(0L, std::min(per, total - *start));
}
struct GroupCrt;                          // per-slot exchange buffers of the prime-per-device CRT product
int group_ctx_create(const int *device_ids, int ndev, ntt_ctx **out);
int group_ctx_destroy(ntt_ctx *g);
int group_ctx_synchronize(ntt_ctx *g);
int group_plan_create(ntt_ctx *g, uint32_t p, int n, int batch, int kind, bool negacyclic, ntt_plan **out);
int group_plan_destroy(ntt_plan *pl);
int group_multiply(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab, size_t ab_stride, int batch, bool negacyclic);
int group_transform(ntt_plan *pl, uint32_t *data, int batch, bool inverse);
int group_crt_multiply(ntt_ctx *g, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64, int batch);
int group_dist_plan_create(ntt_ctx *g, uint32_t p, int lgL, int world, int kind, ntt_dist_plan **out);
void group_dist_plan_release(ntt_dist_plan *dp);
// multidev.cpp: the pointer-table column kernel (kernels_dist.hip) on one rank's piece -- row i1 read at src[i1], row k1
// written at dst[k1], columns [col0, col0 + ncols) of the S / world the rank owns; world <= kMaxDistWorld
int dist_colpass_tables(ntt_dist_plan *part, bool inverse, bool half, const uint32_t *const *src, uint32_t *const *dst,
                        uint32_t col0, uint32_t ncols, hipStream_t stream);

}  // namespace ntt

struct ntt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Independent chunks of a batch are issued round-robin on `nlanes` streams so that the
    // memory-bound column kernels of one chunk overlap the VALU-bound row kernel of another.
    // Lane 0 is `stream` itself; lanes 1.. fork from it and join back through events.
    static constexpr int kMaxLanes = 4;
    int nlanes = 1;
    hipStream_t lane[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_done[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;
    hipStream_t cur = nullptr;                        // stream the next kernel goes to
    int lane_limit = 0;                               // > 0: use at most this many lanes (stream.cpp)
    // this device's copy of the static scratch array of ntt_poly_multiply_batched_noalloc (engine.cpp): its
    // address from hipGetSymbolAddress at context creation, the pointer every use of that memory goes through
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::mutex mu;                                    // serialises calls on this handle
    // A context over several devices (ntt_ctx_create with ndev > 1, multidev.cpp): one ordinary single-device context
    // per SLOT; device ids may repeat (a one-GPU box rehearses {0, 0, 0, 0}).  The group itself owns no stream: plans,
    // products and transforms made from it are sharded over the slots by ntt::partition.  peer_ok: every pair of
    // distinct devices has peer access enabled, so a kernel on one slot may load from and store to another's arrays.
    std::vector<ntt_ctx *> sub;
    bool peer_ok = true;
    ntt::GroupCrt *gcrt = nullptr;
    bool is_group() const { return !sub.empty(); }

    // per-kernel timing (bench.py roofline block)
    bool timing = false;
    int timing_only = -1;                             // >= 0: bracket only this kernel id
    int timing_stride = 1, timing_tick = 0;           // bracket every timing_stride-th launch of the selected kernel(s)
    bool stamp_open = false;                          // begin() recorded a start event
    struct Stamp { int id; hipEvent_t e0, e1; };
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> event_pool;
    double acc_ms[ntt::K_COUNT] = {0};
    long acc_n[ntt::K_COUNT] = {0};
    void begin(int id);
    void end();
    // for launches that carry their own start / stop events: 1 (and the pair) when this launch is to be timed, -1 when
    // timing is on for this kernel but this launch is not in the sample (timing_stride), 0 when timing does not apply
    int begin_ext(int id, hipEvent_t *e0, hipEvent_t *e1);
    int fold();                                       // sync + accumulate + recycle events
    hipEvent_t grab_event();

    // caches owned by the context
    std::map<std::array<uint64_t, 3>, ntt_plan *> plan_cache;     // (p, n, kind): poly_multiply / CRT
    ntt::CrtWorkspace crt;
    // ntt_crt_multiply[_batched] for a modulus that is itself an NTT-friendly prime for the length: the direct 64-bit
    // plan (direct64.cpp) gives the same canonical coefficients in half the time; cached for the last (p64, n) used
    struct CrtDirect {
        std::mutex mu;                                // held across a routed call (the plan may be rebuilt by the next)
        ntt_plan64 *plan = nullptr;
        uint64_t p = 0;
        int n = 0, batch = 0;
    } crt_direct;
    ntt::DeviceBuffer stage_a, stage_b, stage_out;   // host-pointer staging
    ntt::PinnedBuffer pin_a, pin_b, pin_out;         // small calls on pageable arrays: CPU memcpy into / out of these, kernels work on them in place
    ntt::StreamState *stream_state = nullptr;        // ntt_poly_multiply_stream (stream.cpp)
};

struct ntt_plan {
    ntt_ctx *ctx = nullptr;
    uint32_t p = 0, g = 0;
    int n = 0, logL = 0, batch = 0, kind = 0, chunk = 1, lanes = 1;
    uint32_t L = 0;
    bool lazy = false;
    bool wide8 = false;                 // p < 2^29: the row kernels and forward register column passes keep sums unreduced up to 4p (device_ntt.hpp, WIDE)
    int lazy_mode() const { return wide8 ? 2 : lazy ? 1 : 0; }
    ntt::ModParams mp{};
    // 4-step structure (L >= 8192): column levels, outermost first
    int nlevels = 0;
    int level_log[8] = {0};      // log2 N1_j
    int level_logM[8] = {0};     // log2 M_j
    // device tables (tw_t = u32 for Montgomery, {u32,u32} for Barrett)
    void *tw_fwd[8] = {nullptr};
    void *tw_inv[8] = {nullptr};        // innermost level carries the multiply scale (L^-1 [* R])
    void *tw_inv_plain = nullptr;       // innermost level with plain L^-1 (ntt_inverse)
    void *twA = nullptr, *twB = nullptr, *itwA = nullptr, *itwB = nullptr;
    void *twA2 = nullptr, *twB2 = nullptr, *itwA2 = nullptr, *itwB2 = nullptr;     // Montgomery plans: rotated row tables of the fused forward passes (RowArgs)
    void *small_w = nullptr, *small_iw = nullptr;
    // negacyclic plans (products modulo x^n + 1, L = n): psi^i and psi^-i, psi = g^((p-1)/2n);
    // small_itwist = psi^-i with the multiply scale folded in (L <= 4096 path)
    bool negacyclic = false;
    // products of 512 < n <= 1024 coefficients (L = 2048): the 4096-point row kernel on zero-padded rows is
    // twice as fast as the LDS kernel at its natural length (0.021 vs 0.041 us per product), so the
    // multiply goes through this companion plan of L = 4096; transforms keep the plan's own length
    ntt_plan *wide = nullptr;
    bool fold2048 = false;                // companion of a negacyclic n = 2048 plan: linear product of L = 4096, folded at the store
    bool fused = false;                   // L = 8192 / 16384: the product runs in one kernel (seq_product_kernel, kernels_fused.hip)
    void *seq_tw = nullptr;               // its forward table [N1][NIN * 4096]: (psi w_L^k1)^i
    void *twist_fwd = nullptr, *twist_inv = nullptr, *small_itwist = nullptr;
    // by-value kernel constants, raw bytes of Roots64<MM> / tw_t
    std::vector<unsigned char> roots, iroots, scale_mul, scale_plain;
    std::vector<unsigned char> tw_in_fwd[8], tw_in_inv[8];   // per level with N1 >= 32 (col16 kernels)
    // LDS-staged column kernel per direction (NTT_AMD_COL16 bit 0 = forward, bit 1 = inverse).
    // Measured at L = 2^18: inverse 22 % faster with it, forward 7 % slower (profiles/).
    bool col16_fwd = false, col16_inv = true;
    ntt::DeviceBuffer work0, work1;
    std::vector<void *> owned;          // device allocations to free
    // plan of a multi-device context: one plan per slot, each sized for that slot's share of the batch
    std::vector<ntt_plan *> shard;
};

// Distributed transform of length L = 2^lgL over `world` ranks (include/ntt_amd.h, ntt_dist_*):
// `local` transforms this rank's S = L/world points, `colp` is a one-level column-pass plan over the
// [world][S/world] array a rank holds after the first all-to-all, with this rank's slice of the
// 4-step twiddles w_L^(i2*k1).
struct ntt_dist_plan {
    ntt_ctx *ctx = nullptr;
    ntt_plan *local = nullptr;
    ntt_plan *colp = nullptr;
    int lgL = 0, world = 1, rank = 0;
    uint32_t S = 0;
    std::vector<unsigned char> scale;      // world^-1 (inverse column pass, row 0)
    // plan of a multi-device context (world = its slots): the pieces of EVERY rank (part[r] lives on slot r) and what
    // the in-library exchange needs per slot (ntt_dist_forward / ntt_dist_inverse, multidev.cpp)
    std::vector<ntt_dist_plan *> part;
    struct GroupSlot {
        ntt::DeviceBuffer held, sendbuf, rowbuf;          // [world][S/world] staging (copy transports), local inverse output
        ntt::DeviceBuffer spec_a, spec_b;                 // ntt_dist_multiply: the two spectra of this slot
        hipStream_t copy_in = nullptr, copy_out = nullptr;
        hipEvent_t start = nullptr, ready = nullptr, done = nullptr, fin = nullptr, out_done = nullptr;
        std::vector<hipEvent_t> ev_in, ev_k;              // per column chunk
    };
    std::vector<GroupSlot> gslot;
    int transport = 0;                     // 0 p2p (peer loads + peer stores), 1 copy-in + peer stores, 2 copies both ways, 3 RCCL send / recv both ways
    std::vector<void *> comms;             // transport 3: one ncclComm_t per slot (ncclCommInitAll)
    int chunks = 1;                        // column chunks of the copy transports (exchange of chunk c + 1 beside the kernel of chunk c)
    std::mutex gmu;
};
