// multidev.cpp -- one context over several GPUs: ntt_ctx_create(device_ids, ndev > 1).
//
// The reference's distributed program is C++ end to end (main-mpi.cc:642-819): ranks own primes by
// mods_per_process = ceil(4 / size) (:657-660), multiply their residues (:770), send them to rank 0 (:772-791),
// which merges alone (:793-796).  Here the same layout lives INSIDE the library for one process that drives all
// GPUs of a node over HIP peer-to-peer (xGMI): a multi-device context is a list of ordinary single-device contexts
// ("slots"; a device id may repeat, which is how a one-GPU box rehearses {0, 0, 0, 0}) and
//
//   * ntt_poly_multiply_batched / ntt_negacyclic_multiply_batched / ntt_forward / ntt_inverse shard the batch rows
//     over the slots by partition() -- independent units, no exchange;
//   * ntt_crt_multiply puts prime k on slot k / ceil(4 / ndev) (the reference's rule), then ONE exchange step: every
//     slot receives coefficient slice h of the four residue polynomials (peer copies issued by the owner behind its
//     product), merges its slice on its GPU and writes it to the caller's array -- an all-to-all instead of the
//     reference's gather-and-merge-on-root, so the merge is ndev-way parallel and every link carries traffic
//     (SURVEY 8e(2)); a batch of large-modulus products is sharded by rows instead;
//   * ntt_dist_forward / ntt_dist_inverse run ONE transform of length L = ndev * S whose block r lives on slot r
//     (SURVEY 8e(3)): the column kernel of kernels_dist.hip loads its G points from the peers' arrays and stores row
//     k1 into slot k1's array, so both all-to-alls of the distributed 4-step are peer loads / stores inside one
//     kernel per device ("p2p" transport); without peer access, or on request, the exchanges are
//     hipMemcpyPeerAsync per peer slice around the kernel, cut into column chunks so that the copies of chunk c + 1
//     run beside the kernel of chunk c ("copyin": copies in, peer stores out; "copy": copies both ways).
//
// One process per GPU with torch.distributed / RCCL between the processes stays available (ntt_amd/shard.py,
// ntt_amd/launch.py): that is how bench.py --gpus N is launched.  Never run on more than one physical GPU by us.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include <dlfcn.h>

#include "../../include/ntt_amd.h"
#include "engine.hpp"
#include "rccl_dl.hpp"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

namespace ntt {

// ---- RCCL, loaded on demand (transport "rccl" of the split transform) ----------------------------------------------------
// The library does not link librccl: a Python process that also runs torch.distributed already holds torch's own copy,
// and every default path here is plain HIP.  Only NTT_AMD_DIST_TRANSPORT=rccl opens it (dlopen, local symbols) and uses
// the six calls below -- the grouped ncclSend / ncclRecv form of an all-to-all, one communicator per slot in this one
// process (ncclCommInitAll), each pair of calls on the slot's own exchange stream.
Rccl *rccl_load()
{
    static std::mutex mu;
    static Rccl r;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            auto sym = [&](const char *n) { return dlsym(r.lib, n); };
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
            r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
                dlclose(r.lib);
                r.lib = nullptr;
            }
        }
    }
    return r.lib ? &r : nullptr;
}
struct GroupCrt {
    int n = 0;
    size_t sl = 0;                                       // coefficients per slot slice
    struct Slot {
        DeviceBuffer send, recv, mine;                   // [primes owned][ndev * sl] u32, [4][sl] u32, [sl] u64
        hipEvent_t sent = nullptr;
    };
    std::vector<Slot> slot;
};

// f(slot) for every slot: slot 0 on the calling thread, the others on threads of their own (a call with host arrays
// blocks until its copies are done; with a thread per slot the devices work side by side).  The error message of the
// library is thread-local: a failing worker's message is carried over to the caller's thread.
template <class F>
static int for_each_slot(int nslots, F &&f)
{
    std::vector<int> rc((size_t)nslots, NTT_OK);
    std::vector<std::string> msg((size_t)nslots);
    auto run = [&](int s) {
        try {
            rc[(size_t)s] = f(s);
        } catch (...) {
            rc[(size_t)s] = exception_status();
        }
        if (rc[(size_t)s]) msg[(size_t)s] = last_error_cstr();
    };
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
    } joiner;
    joiner.th.reserve((size_t)nslots);
    for (int s = 1; s < nslots; ++s) joiner.th.emplace_back(run, s);
    run(0);
    for (auto &t : joiner.th) t.join();
    for (int s = 0; s < nslots; ++s)
        if (rc[(size_t)s]) return fail(rc[(size_t)s], "slot " + std::to_string(s) + ": " + msg[(size_t)s]);
    return NTT_OK;
}

// -1: host memory (or unknown to every runtime); -2: device memory the HIP runtime of this library did not allocate
// (another runtime in the process); >= 0: the HIP device that owns the allocation
static int pointer_device(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) == hipSuccess && (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged))
        return at.device;
    (void)hipGetLastError();
    return is_device_ptr(p) ? -2 : -1;
}

// may a kernel on slot `s` touch p?  (host arrays are staged by the single-device entry points: always fine)
static int check_reachable(const ntt_ctx *g, const void *p, int s, const char *what)
{
    const int dev = pointer_device(p), mine = g->sub[(size_t)s]->device;
    if (dev == -1 || dev == mine) return NTT_OK;
    bool in_group = dev == -2;
    for (const ntt_ctx *c : g->sub) in_group = in_group || c->device == dev;
    if (in_group && g->peer_ok) return NTT_OK;
    return fail(NTT_ERR_UNSUPPORTED, std::string(what) + ": the array lives on device " + std::to_string(dev) +
                                         ", which slot " + std::to_string(s) + " (device " + std::to_string(mine) +
                                         ") cannot reach (no peer access): pass host arrays, or arrays on the slot's own device");
}

int group_ctx_create(const int *device_ids, int ndev, ntt_ctx **out)
{
    if (!device_ids) return fail(NTT_ERR_INVALID, "ntt_ctx_create: device_ids is null with ndev > 1");
    if (ndev > 64) return fail(NTT_ERR_INVALID, "ntt_ctx_create: at most 64 slots");
    std::unique_ptr<ntt_ctx> g(new ntt_ctx());
    struct Undo {
        ntt_ctx *g;
        bool armed = true;
        ~Undo() { if (armed) for (ntt_ctx *c : g->sub) ntt_ctx_destroy(c); }
    } undo{g.get()};
    for (int i = 0; i < ndev; ++i) {
        ntt_ctx *c = nullptr;
        const int rc = ntt_ctx_create(&device_ids[i], 1, &c);
        if (rc) return rc;
        g->sub.push_back(c);
    }
    g->device = device_ids[0];
    // peer access between every ordered pair of distinct devices (xGMI on an MI355X node)
    std::vector<int> uniq;
    for (int i = 0; i < ndev; ++i)
        if (std::find(uniq.begin(), uniq.end(), device_ids[i]) == uniq.end()) uniq.push_back(device_ids[i]);
    for (int di : uniq)
        for (int dj : uniq) {
            if (di == dj) continue;
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, di, dj));
            if (!can) { g->peer_ok = false; continue; }
            HIP_TRY(hipSetDevice(di));
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (e != hipErrorPeerAccessAlreadyEnabled) g->peer_ok = false;
            }
        }
    if (const char *e = std::getenv("NTT_AMD_NO_PEER"))    // rehearsal of the no-peer-access routes on any box
        if (e[0] == '1') g->peer_ok = false;
    undo.armed = false;
    *out = g.release();
    return NTT_OK;
}

static void group_crt_release(ntt_ctx *g)
{
    if (!g->gcrt) return;
    for (size_t s = 0; s < g->gcrt->slot.size(); ++s) {
        (void)hipSetDevice(g->sub[s]->device);
        auto &sl = g->gcrt->slot[s];
        sl.send.release(); sl.recv.release(); sl.mine.release();
        if (sl.sent) (void)hipEventDestroy(sl.sent);
    }
    delete g->gcrt;
    g->gcrt = nullptr;
}

int group_ctx_synchronize(ntt_ctx *g)
{
    int rc = NTT_OK;
    for (ntt_ctx *c : g->sub) {
        const int r = ntt_ctx_synchronize(c);
        if (r && !rc) rc = r;
    }
    return rc;
}

int group_ctx_destroy(ntt_ctx *g)
{
    (void)group_ctx_synchronize(g);
    group_crt_release(g);
    for (ntt_ctx *c : g->sub) ntt_ctx_destroy(c);
    delete g;
    return NTT_OK;
}

// ---- plans: one per slot, sized for the slot's share of the batch -----------------------------------------------------
int group_plan_create(ntt_ctx *g, uint32_t p, int n, int batch, int kind, bool negacyclic, ntt_plan **out)
{
    if (batch <= 0) return fail(NTT_ERR_INVALID, "ntt_plan_create: n and batch must be positive");
    const int nd = (int)g->sub.size();
    std::unique_ptr<ntt_plan> pl(new ntt_plan());
    pl->ctx = g;
    pl->p = p; pl->n = n; pl->batch = batch; pl->kind = kind; pl->negacyclic = negacyclic;
    long start = 0, share = 0;
    partition(batch, nd, 0, &start, &share);               // slot 0 owns the largest share
    for (int s = 0; s < nd; ++s) {
        ntt_plan *sp = nullptr;
        const int rc = negacyclic ? ntt_negacyclic_plan_create(g->sub[(size_t)s], p, n, (int)share, kind, &sp)
                                  : ntt_plan_create(g->sub[(size_t)s], p, n, (int)share, kind, &sp);
        if (rc) {
            for (ntt_plan *q : pl->shard) ntt_plan_destroy(q);
            return rc;
        }
        pl->shard.push_back(sp);
    }
    pl->L = pl->shard[0]->L;
    pl->logL = pl->shard[0]->logL;
    pl->g = pl->shard[0]->g;
    *out = pl.release();
    return NTT_OK;
}

int group_plan_destroy(ntt_plan *pl)
{
    for (ntt_plan *q : pl->shard) ntt_plan_destroy(q);
    delete pl;
    return NTT_OK;
}

int group_multiply(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab, size_t ab_stride, int batch, bool negacyclic)
{
    if (!a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: null argument");
    if (batch <= 0 || batch > pl->batch) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: batch must be in 1..plan capacity");
    ntt_ctx *g = pl->ctx;
    const int nd = (int)g->sub.size();
    const size_t n = (size_t)pl->n;
    for (int s = 0; s < nd; ++s) {
        long start, count;
        partition(batch, nd, s, &start, &count);
        if (!count) continue;
        int rc;
        if ((rc = check_reachable(g, a, s, "ntt_poly_multiply_batched: a")) || (rc = check_reachable(g, b, s, "ntt_poly_multiply_batched: b")) ||
            (rc = check_reachable(g, ab, s, "ntt_poly_multiply_batched: ab")))
            return rc;
    }
    return for_each_slot(nd, [&](int s) -> int {
        long start, count;
        partition(batch, nd, s, &start, &count);            // rows [start, start + count) of the batch: slot s
        if (!count) return NTT_OK;
        const uint32_t *sa = a + (size_t)start * n, *sb = b + (size_t)start * n;
        uint32_t *sab = ab + (size_t)start * ab_stride;
        return negacyclic ? ntt_negacyclic_multiply_batched(pl->shard[(size_t)s], sa, sb, sab, ab_stride, (int)count)
                          : ntt_poly_multiply_batched(pl->shard[(size_t)s], sa, sb, sab, ab_stride, (int)count);
    });
}

int group_transform(ntt_plan *pl, uint32_t *data, int batch, bool inverse)
{
    if (!data) return fail(NTT_ERR_INVALID, "ntt_forward/inverse: null argument");
    if (batch <= 0 || batch > pl->batch) return fail(NTT_ERR_INVALID, "ntt_forward/inverse: batch must be in 1..plan capacity");
    ntt_ctx *g = pl->ctx;
    const int nd = (int)g->sub.size();
    for (int s = 0; s < nd; ++s)
        if (int rc = check_reachable(g, data, s, "ntt_forward/inverse: data")) return rc;
    return for_each_slot(nd, [&](int s) -> int {
        long start, count;
        partition(batch, nd, s, &start, &count);
        if (!count) return NTT_OK;
        uint32_t *d = data + (size_t)start * pl->L;
        return inverse ? ntt_inverse(pl->shard[(size_t)s], d, (int)count) : ntt_forward(pl->shard[(size_t)s], d, (int)count);
    });
}

// ---- large-modulus product, prime per slot + one exchange ----------------------------------------------------------------
static int group_crt_workspace_build(ntt_ctx *g, int n);
static int group_crt_workspace(ntt_ctx *g, int n)
{
    if (g->gcrt && g->gcrt->n == n) return NTT_OK;
    const int rc = group_crt_workspace_build(g, n);
    if (rc) group_crt_release(g);                           // never leave a half-built workspace behind for the next call
    return rc;
}
static int group_crt_workspace_build(ntt_ctx *g, int n)
{
    const int nd = (int)g->sub.size();
    group_crt_release(g);
    std::unique_ptr<GroupCrt> w(new GroupCrt());
    w->n = n;
    const size_t out_len = 2 * (size_t)n - 1;
    w->sl = (out_len + (size_t)nd - 1) / (size_t)nd;
    w->slot.resize((size_t)nd);
    g->gcrt = w.release();                                  // group_crt_release cleans up after a partial failure below
    const int mpp = (4 + nd - 1) / nd;
    for (int s = 0; s < nd; ++s) {
        ntt_ctx *c = g->sub[(size_t)s];
        HIP_TRY(hipSetDevice(c->device));
        auto &sl = g->gcrt->slot[(size_t)s];
        long start, count;
        partition(4, nd, s, &start, &count);
        int rc = NTT_OK;
        if (count) {
            // the residue product fills 2n - 1 words of a [ndev * sl] row; the zero tail is written once, here
            const size_t bytes = (size_t)mpp * nd * g->gcrt->sl * sizeof(uint32_t);
            if ((rc = sl.send.ensure(bytes))) return rc;
            HIP_TRY(hipMemsetAsync(sl.send.ptr, 0, bytes, c->stream));
        }
        if ((rc = sl.recv.ensure(4 * g->gcrt->sl * sizeof(uint32_t))) || (rc = sl.mine.ensure(g->gcrt->sl * sizeof(uint64_t)))) return rc;
        HIP_TRY(hipEventCreateWithFlags(&sl.sent, hipEventDisableTiming));
    }
    return NTT_OK;
}

int group_crt_multiply(ntt_ctx *g, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64, int batch)
{
    if (!a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_crt_multiply: null argument");
    if (n <= 0 || batch <= 0) return fail(NTT_ERR_INVALID, "ntt_crt_multiply: n and batch must be positive");
    const int nd = (int)g->sub.size();
    const size_t N = (size_t)n, out_len = 2 * N - 1;
    std::lock_guard<std::mutex> lk(g->mu);
    if (batch > 1) {
        // independent products: rows of the batch per slot, each the whole path on its GPU -- no exchange
        for (int s = 0; s < nd; ++s) {
            int rc;
            if ((rc = check_reachable(g, a, s, "ntt_crt_multiply_batched: a")) || (rc = check_reachable(g, b, s, "ntt_crt_multiply_batched: b")) ||
                (rc = check_reachable(g, ab, s, "ntt_crt_multiply_batched: ab")))
                return rc;
        }
        const int rc = for_each_slot(nd, [&](int s) -> int {
            long start, count;
            partition(batch, nd, s, &start, &count);
            if (!count) return NTT_OK;
            return ntt_crt_multiply_batched(g->sub[(size_t)s], a + (size_t)start * N, b + (size_t)start * N,
                                            ab + (size_t)start * out_len, n, p64, (int)count);
        });
        const int src = group_ctx_synchronize(g);
        return rc ? rc : src;
    }
    // the reference's constraints, checked by the single-device entry points as well; here before anything is queued
    if ((p64 & 1) == 0 || p64 < 3 || p64 >= (1ULL << 62)) return fail(NTT_ERR_UNSUPPORTED, "ntt_crt_multiply: p64 must be odd and below 2^62");
    if (n > (1 << 22)) return fail(NTT_ERR_UNSUPPORTED, "ntt_crt_multiply: n too large for the 2^23-adic prime 998244353");
    {
        unsigned __int128 M = 1;
        for (int k = 0; k < 4; ++k) { uint32_t m = 0; ntt_crt_prime(k, &m); M *= m; }
        const unsigned __int128 sq = (unsigned __int128)(p64 - 1) * (p64 - 1);
        if (sq > M / (unsigned)n) return fail(NTT_ERR_UNSUPPORTED, "ntt_crt_multiply: n * p64^2 exceeds the product of the four CRT primes");
    }
    int rc = group_crt_workspace(g, n);
    if (rc) return rc;
    GroupCrt &w = *g->gcrt;
    const size_t sl = w.sl, row = (size_t)nd * sl;
    for (int s = 0; s < nd; ++s) {
        long start, count;
        partition(4, nd, s, &start, &count);
        if (!count) continue;
        if ((rc = check_reachable(g, a, s, "ntt_crt_multiply: a")) || (rc = check_reachable(g, b, s, "ntt_crt_multiply: b"))) return rc;
    }
    // step 1 (main-mpi.cc:770): slot s multiplies the residues of its primes; the products land in exchange order
    // (row h of `send` = the slice slot h merges) and the owner pushes every slice to its slot behind the product
    rc = for_each_slot(nd, [&](int s) -> int {
        long start, count;
        partition(4, nd, s, &start, &count);
        if (!count) return NTT_OK;
        ntt_ctx *c = g->sub[(size_t)s];
        uint32_t *send = static_cast<uint32_t *>(w.slot[(size_t)s].send.ptr);
        for (long j = 0; j < count; ++j)
            if (int r = ntt_crt_residue_multiply(c, a, b, send + (size_t)j * row, n, (int)(start + j))) return r;
        HIP_TRY(hipSetDevice(c->device));
        for (int h = 0; h < nd; ++h) {
            uint32_t *recv = static_cast<uint32_t *>(w.slot[(size_t)h].recv.ptr);
            for (long j = 0; j < count; ++j)
                HIP_TRY(hipMemcpyPeerAsync(recv + (size_t)(start + j) * sl, g->sub[(size_t)h]->device,
                                           send + (size_t)j * row + (size_t)h * sl, c->device, sl * sizeof(uint32_t), c->stream));
        }
        HIP_TRY(hipEventRecord(w.slot[(size_t)s].sent, c->stream));
        return NTT_OK;
    });
    // steps 2 and 3 (main-mpi.cc:793-796, on every slot instead of rank 0): merge the slice, write it to the caller
    for (int h = 0; h < nd && !rc; ++h) {
        ntt_ctx *c = g->sub[(size_t)h];
        const size_t lo = (size_t)h * sl;
        if (lo >= out_len) continue;
        const size_t cnt = std::min(sl, out_len - lo);
        if (hipSetDevice(c->device) != hipSuccess) { rc = fail(NTT_ERR_HIP, "hipSetDevice"); break; }
        for (int o = 0; o < nd; ++o) {
            long start, count;
            partition(4, nd, o, &start, &count);
            if (count && hipStreamWaitEvent(c->stream, w.slot[(size_t)o].sent, 0) != hipSuccess) { rc = fail(NTT_ERR_HIP, "hipStreamWaitEvent (CRT exchange)"); break; }
        }
        if (rc) break;
        const uint32_t *recv = static_cast<const uint32_t *>(w.slot[(size_t)h].recv.ptr);
        uint64_t *mine = static_cast<uint64_t *>(w.slot[(size_t)h].mine.ptr);
        if ((rc = ntt_crt_merge(c, recv, recv + sl, recv + 2 * sl, recv + 3 * sl, mine, cnt, p64))) break;
        const hipError_t e = hipMemcpyAsync(ab + lo, mine, cnt * sizeof(uint64_t), hipMemcpyDefault, c->stream);
        if (e != hipSuccess) rc = fail(NTT_ERR_HIP, std::string("hipMemcpyAsync (CRT result slice): ") + hipGetErrorString(e));
    }
    const int src = group_ctx_synchronize(g);               // the product is complete on return
    return rc ? rc : src;
}

// ---- one transform split over the slots ---------------------------------------------------------------------------------
void group_dist_plan_release(ntt_dist_plan *dp)
{
    for (size_t s = 0; s < dp->gslot.size(); ++s) {
        auto &gs = dp->gslot[s];
        if (s < dp->part.size() && dp->part[s]) {
            (void)hipSetDevice(dp->part[s]->ctx->device);
            (void)hipStreamSynchronize(dp->part[s]->ctx->stream);
        }
        if (gs.copy_in) { (void)hipStreamSynchronize(gs.copy_in); (void)hipStreamDestroy(gs.copy_in); }
        if (gs.copy_out) { (void)hipStreamSynchronize(gs.copy_out); (void)hipStreamDestroy(gs.copy_out); }
        for (hipEvent_t e : {gs.start, gs.ready, gs.done, gs.fin, gs.out_done})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : gs.ev_in) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : gs.ev_k) if (e) (void)hipEventDestroy(e);
        gs.held.release(); gs.sendbuf.release(); gs.rowbuf.release();
        gs.spec_a.release(); gs.spec_b.release();
    }
    dp->gslot.clear();
    if (!dp->comms.empty())
        if (Rccl *r = rccl_load())
            for (void *c : dp->comms) if (c) (void)r->CommDestroy(c);
    dp->comms.clear();
    for (ntt_dist_plan *q : dp->part) ntt_dist_plan_destroy(q);
    dp->part.clear();
}

int group_dist_plan_create(ntt_ctx *g, uint32_t p, int lgL, int world, int kind, ntt_dist_plan **out)
{
    const int nd = (int)g->sub.size();
    if (world != nd) return fail(NTT_ERR_INVALID, "ntt_dist_plan_create: on a multi-device context world must equal its number of slots");
    if (world > kMaxDistWorld || (world & (world - 1))) return fail(NTT_ERR_UNSUPPORTED, "ntt_dist_plan_create: a multi-device context splits a transform over 2, 4, 8 or 16 slots");
    std::unique_ptr<ntt_dist_plan> dp(new ntt_dist_plan());
    struct Undo {
        ntt_dist_plan *dp;
        bool armed = true;
        ~Undo() { if (armed) group_dist_plan_release(dp); }
    } undo{dp.get()};
    dp->ctx = g;
    dp->lgL = lgL;
    dp->world = world;
    dp->rank = 0;
    for (int r = 0; r < world; ++r) {
        ntt_dist_plan *q = nullptr;
        const int rc = ntt_dist_plan_create(g->sub[(size_t)r], p, lgL, world, r, kind, &q);
        if (rc) return rc;
        dp->part.push_back(q);
    }
    dp->S = dp->part[0]->S;
    // transport: peer loads and stores inside the column kernel need peer access (or one device); the copy forms do not
    dp->transport = g->peer_ok ? 0 : 2;
    if (const char *e = std::getenv("NTT_AMD_DIST_TRANSPORT")) {
        const std::string t(e);
        if (t == "p2p") dp->transport = 0;
        else if (t == "copyin") dp->transport = 1;
        else if (t == "copy") dp->transport = 2;
        else if (t == "rccl") dp->transport = 3;
        else return fail(NTT_ERR_INVALID, "NTT_AMD_DIST_TRANSPORT must be p2p, copyin, copy or rccl");
        if (dp->transport < 2 && !g->peer_ok)
            return fail(NTT_ERR_UNSUPPORTED, "NTT_AMD_DIST_TRANSPORT=" + t + " needs peer access between the devices of the context");
    }
    const uint32_t cols = dp->S / (uint32_t)world;
    int chunks = dp->transport == 0 ? 1 : 4;
    if (const char *e = std::getenv("NTT_AMD_DIST_CHUNKS")) chunks = std::atoi(e);
    while (chunks > 1 && (cols % ((uint32_t)chunks * 256u)) != 0) --chunks;       // whole workgroups per chunk
    dp->chunks = std::max(1, chunks);
    dp->gslot.resize((size_t)world);
    const size_t bytes = (size_t)dp->S * sizeof(uint32_t);
    for (int s = 0; s < world; ++s) {
        auto &gs = dp->gslot[(size_t)s];
        HIP_TRY(hipSetDevice(g->sub[(size_t)s]->device));
        int rc = gs.rowbuf.ensure(bytes);
        if (!rc && dp->transport >= 1) rc = gs.held.ensure(bytes);
        if (!rc && dp->transport >= 2) rc = gs.sendbuf.ensure(bytes);
        if (rc) return rc;
        if (dp->transport >= 1) HIP_TRY(hipStreamCreateWithFlags(&gs.copy_in, hipStreamNonBlocking));
        if (dp->transport >= 2) HIP_TRY(hipStreamCreateWithFlags(&gs.copy_out, hipStreamNonBlocking));
        for (hipEvent_t *e : {&gs.start, &gs.ready, &gs.done, &gs.fin, &gs.out_done})
            HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        gs.ev_in.assign((size_t)dp->chunks, nullptr);
        gs.ev_k.assign((size_t)dp->chunks, nullptr);
        for (int c = 0; c < dp->chunks; ++c) {
            HIP_TRY(hipEventCreateWithFlags(&gs.ev_in[(size_t)c], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&gs.ev_k[(size_t)c], hipEventDisableTiming));
        }
    }
    if (dp->transport == 3) {
        // one RCCL communicator per slot, all in this process.  RCCL wants distinct devices: a one-GPU rehearsal with
        // repeated ids is refused here (NTT_ERR_UNSUPPORTED with RCCL's own message) -- this transport has never moved data
        Rccl *r = rccl_load();
        if (!r) return fail(NTT_ERR_UNSUPPORTED, "NTT_AMD_DIST_TRANSPORT=rccl: librccl.so could not be loaded");
        std::vector<int> devs;
        for (ntt_ctx *c : g->sub) devs.push_back(c->device);
        dp->comms.assign((size_t)world, nullptr);
        const int e = r->CommInitAll(dp->comms.data(), world, devs.data());
        if (e != 0) {
            dp->comms.clear();
            return fail(NTT_ERR_UNSUPPORTED, std::string("NTT_AMD_DIST_TRANSPORT=rccl: ncclCommInitAll: ") + r->GetErrorString(e));
        }
    }
    undo.armed = false;
    *out = dp.release();
    return NTT_OK;
}

// the column kernel of one rank's piece (`part`: a single-device plan) over columns [col0, col0 + ncols): row i1 is read
// at src[i1], row k1 is written at dst[k1] -- wherever those arrays live (the slot's own memory, a peer's, staging)
template <class MM>
static int launch_part_colpass(ntt_dist_plan *part, bool inverse, bool half, const uint32_t *const *src, uint32_t *const *dst,
                               uint32_t col0, uint32_t ncols, hipStream_t stream)
{
    ntt_plan *cp = part->colp;
    const int G = part->world;
    const uint32_t cols = part->S / (uint32_t)G;
    DistColArgs<MM> a{};
    for (int i = 0; i < G; ++i) {
        a.in[i] = src[i];
        a.out[i] = dst[i];
    }
    a.tw = static_cast<const typename MM::tw_t *>(inverse ? cp->tw_inv[0] : cp->tw_fwd[0]);
    a.use_scale0 = inverse ? 1 : 0;
    if (inverse) std::memcpy(&a.scale0, part->scale.data(), sizeof(a.scale0));
    a.cols = cols;
    a.col0 = col0;
    a.ncols = ncols;
    a.half = (half && !inverse) ? 1 : 0;
    std::memcpy(&a.roots, (inverse ? cp->iroots : cp->roots).data(), sizeof(a.roots));
    int lgG = 0;
    while ((1 << lgG) < G) ++lgG;
    if (!launch_dist_col(cp->kind, cp->lazy, lgG, inverse, &a, cp->mp, stream)) return take_launch_error();
    return NTT_OK;
}

int dist_colpass_tables(ntt_dist_plan *part, bool inverse, bool half, const uint32_t *const *src, uint32_t *const *dst,
                        uint32_t col0, uint32_t ncols, hipStream_t stream)
{
    if (part->world > kMaxDistWorld) return fail(NTT_ERR_UNSUPPORTED, "the pointer-table column kernel serves at most 16 ranks");
    return part->colp->kind == NTT_MODMUL_MONTGOMERY ? launch_part_colpass<Mont32>(part, inverse, half, src, dst, col0, ncols, stream)
                                                      : launch_part_colpass<Barrett32>(part, inverse, half, src, dst, col0, ncols, stream);
}

// Forward: in[r] = x[r * S .. (r + 1) * S) on slot r  ->  out[r][k2] = X[r + world * k2].
// Inverse: the reverse, scaled by L^-1.  Everything is queued; ntt_ctx_synchronize(group) waits for it.
// scrambled: the local transforms leave / take the spectrum in the engine's own order (no permutation pass): the two
// halves of ntt_dist_multiply.
// half (forward only): the blocks of slots >= world / 2 are zero padding -- never read, never exchanged.
static int group_dist_run(ntt_dist_plan *dp, const uint32_t *const *in, uint32_t *const *out, bool inverse, bool scrambled = false,
                          bool half = false)
{
    ntt_ctx *g = dp->ctx;
    const int G = dp->world;
    const uint32_t S = dp->S, cols = S / (uint32_t)G;
    const int C = dp->chunks;
    const uint32_t cc = cols / (uint32_t)C;
    const int nz = (half && !inverse) ? G / 2 : G;            // rows (= blocks) that hold data
    // every slot starts behind the end of the previous call on EVERY slot: peers are about to write into its arrays
    for (int s = 0; s < G; ++s) {
        HIP_TRY(hipSetDevice(g->sub[(size_t)s]->device));
        for (int o = 0; o < G; ++o) HIP_TRY(hipStreamWaitEvent(g->sub[(size_t)s]->stream, dp->gslot[(size_t)o].fin, 0));
    }
    // inverse, first half: the local S-point inverse transforms, in[r] -> rowbuf[r] (in[] stays untouched)
    std::vector<const uint32_t *> rows((size_t)G);
    if (inverse) {
        for (int r = 0; r < G; ++r) {
            uint32_t *rb = static_cast<uint32_t *>(dp->gslot[(size_t)r].rowbuf.ptr);
            if (int rc = transform_device_any(dp->part[(size_t)r]->local, in[r], rb, true, scrambled)) return rc;
            HIP_TRY(hipEventRecord(dp->gslot[(size_t)r].ready, g->sub[(size_t)r]->stream));
            rows[(size_t)r] = rb;
        }
    } else {
        for (int r = 0; r < G; ++r) rows[(size_t)r] = in[r];
    }
    // the column pass of every slot, exchanges included, one column chunk after the other: the exchange of chunk c + 1
    // (copy / RCCL streams) runs beside the kernel of chunk c (the slot's compute stream)
    Rccl *rc_lib = dp->transport == 3 ? rccl_load() : nullptr;
    if (dp->transport == 3 && !rc_lib) return fail(NTT_ERR_INTERNAL, "RCCL transport without librccl");
    for (int h = 0; h < G; ++h) {
        ntt_ctx *c = g->sub[(size_t)h];
        auto &gs = dp->gslot[(size_t)h];
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t sh = c->stream;
        if (inverse)
            for (int r = 0; r < G; ++r)
                if (r != h) HIP_TRY(hipStreamWaitEvent(sh, dp->gslot[(size_t)r].ready, 0));
        if (dp->transport >= 1) {                           // exchanges run on streams of their own, behind everything queued so far
            HIP_TRY(hipEventRecord(gs.start, sh));
            HIP_TRY(hipStreamWaitEvent(gs.copy_in, gs.start, 0));
            if (gs.copy_out) HIP_TRY(hipStreamWaitEvent(gs.copy_out, gs.start, 0));
        }
    }
    for (int ch = 0; ch < C; ++ch) {
        const uint32_t c0 = (uint32_t)ch * cc;
        if (dp->transport == 3) {                           // exchange in: grouped send / recv, every slot's slice of every row
            RCCL_TRY(rc_lib, rc_lib->GroupStart());
            for (int h = 0; h < G; ++h) {
                auto &gs = dp->gslot[(size_t)h];
                HIP_TRY(hipSetDevice(g->sub[(size_t)h]->device));
                uint32_t *held = static_cast<uint32_t *>(gs.held.ptr);
                for (int i = 0; i < G; ++i) {
                    if (h < nz) RCCL_TRY(rc_lib, rc_lib->Send(rows[(size_t)h] + (size_t)i * cols + c0, cc, Rccl::kUint32, i, dp->comms[(size_t)h], gs.copy_in));
                    if (i < nz) RCCL_TRY(rc_lib, rc_lib->Recv(held + (size_t)i * cols + c0, cc, Rccl::kUint32, i, dp->comms[(size_t)h], gs.copy_in));
                }
            }
            RCCL_TRY(rc_lib, rc_lib->GroupEnd());
        }
        for (int h = 0; h < G; ++h) {
            ntt_ctx *c = g->sub[(size_t)h];
            auto &gs = dp->gslot[(size_t)h];
            HIP_TRY(hipSetDevice(c->device));
            hipStream_t sh = c->stream;
            uint32_t *held = static_cast<uint32_t *>(gs.held.ptr), *sendbuf = static_cast<uint32_t *>(gs.sendbuf.ptr);
            const uint32_t *src[kMaxDistWorld];
            uint32_t *dst[kMaxDistWorld];
            for (int i = 0; i < G; ++i) {
                src[i] = i >= nz ? nullptr : dp->transport == 0 ? rows[(size_t)i] + (size_t)h * cols : held + (size_t)i * cols;
                dst[i] = dp->transport >= 2 ? sendbuf + (size_t)i * cols : out[i] + (size_t)h * cols;
            }
            if (dp->transport == 1 || dp->transport == 2)   // exchange in: this chunk's slice of every peer's row
                for (int i = 0; i < nz; ++i)
                    HIP_TRY(hipMemcpyPeerAsync(held + (size_t)i * cols + c0, c->device, rows[(size_t)i] + (size_t)h * cols + c0,
                                               g->sub[(size_t)i]->device, (size_t)cc * sizeof(uint32_t), gs.copy_in));
            if (dp->transport >= 1) {
                HIP_TRY(hipEventRecord(gs.ev_in[(size_t)ch], gs.copy_in));
                HIP_TRY(hipStreamWaitEvent(sh, gs.ev_in[(size_t)ch], 0));
            }
            const int rc = dist_colpass_tables(dp->part[(size_t)h], inverse, half, src, dst, c0, cc, sh);
            if (rc) return rc;
            if (dp->transport >= 2) {                       // exchange out: row k1 of this chunk to slot k1
                HIP_TRY(hipEventRecord(gs.ev_k[(size_t)ch], sh));
                HIP_TRY(hipStreamWaitEvent(gs.copy_out, gs.ev_k[(size_t)ch], 0));
            }
            if (dp->transport == 2)
                for (int i = 0; i < G; ++i)
                    HIP_TRY(hipMemcpyPeerAsync(out[i] + (size_t)h * cols + c0, g->sub[(size_t)i]->device, sendbuf + (size_t)i * cols + c0,
                                               c->device, (size_t)cc * sizeof(uint32_t), gs.copy_out));
        }
        if (dp->transport == 3) {
            RCCL_TRY(rc_lib, rc_lib->GroupStart());
            for (int h = 0; h < G; ++h) {
                auto &gs = dp->gslot[(size_t)h];
                HIP_TRY(hipSetDevice(g->sub[(size_t)h]->device));
                const uint32_t *sendbuf = static_cast<const uint32_t *>(gs.sendbuf.ptr);
                for (int i = 0; i < G; ++i) {
                    RCCL_TRY(rc_lib, rc_lib->Send(sendbuf + (size_t)i * cols + c0, cc, Rccl::kUint32, i, dp->comms[(size_t)h], gs.copy_out));
                    RCCL_TRY(rc_lib, rc_lib->Recv(out[h] + (size_t)i * cols + c0, cc, Rccl::kUint32, i, dp->comms[(size_t)h], gs.copy_out));
                }
            }
            RCCL_TRY(rc_lib, rc_lib->GroupEnd());
        }
    }
    for (int h = 0; h < G; ++h) {
        ntt_ctx *c = g->sub[(size_t)h];
        auto &gs = dp->gslot[(size_t)h];
        HIP_TRY(hipSetDevice(c->device));
        if (dp->transport >= 2) {
            HIP_TRY(hipEventRecord(gs.out_done, gs.copy_out));
            HIP_TRY(hipStreamWaitEvent(c->stream, gs.out_done, 0));
        }
        HIP_TRY(hipEventRecord(gs.done, c->stream));
    }
    HIP_TRY(hipGetLastError());
    // every slot waits for the rows its peers sent; forward: then its local S-point transform, in place in out[r]
    for (int r = 0; r < G; ++r) {
        ntt_ctx *c = g->sub[(size_t)r];
        HIP_TRY(hipSetDevice(c->device));
        for (int h = 0; h < G; ++h)
            if (h != r) HIP_TRY(hipStreamWaitEvent(c->stream, dp->gslot[(size_t)h].done, 0));
        if (!inverse)
            if (int rc = transform_device_any(dp->part[(size_t)r]->local, out[r], out[r], false, scrambled)) return rc;
        HIP_TRY(hipEventRecord(dp->gslot[(size_t)r].fin, c->stream));
    }
    return NTT_OK;
}

// ab = a * b for two polynomials of n = L / 2 coefficients whose blocks of S live on the slots (block r of a and b on
// slot r < world / 2; the upper half of the transform is zero padding): the harness multiply body
// (main-montgomery.cc:326-351) at a size one GPU need not hold -- forward split transform of a and of b (spectra stay in
// the engine's order: no permutation pass), pointwise product on every slot, inverse split transform into ab.
static int group_dist_multiply(ntt_dist_plan *dp, const uint32_t *const *a, const uint32_t *const *b, uint32_t *const *ab)
{
    ntt_ctx *g = dp->ctx;
    const int G = dp->world;
    const size_t bytes = (size_t)dp->S * sizeof(uint32_t);
    std::vector<const uint32_t *> ina((size_t)G), inb((size_t)G), spa_c((size_t)G);
    std::vector<uint32_t *> spa((size_t)G), spb((size_t)G);
    for (int r = 0; r < G; ++r) {
        auto &gs = dp->gslot[(size_t)r];
        ntt_ctx *c = g->sub[(size_t)r];
        HIP_TRY(hipSetDevice(c->device));
        int rc = gs.spec_a.ensure(bytes);
        if (!rc) rc = gs.spec_b.ensure(bytes);
        if (rc) return rc;
        ina[(size_t)r] = r < G / 2 ? a[r] : nullptr;         // the upper half of both operands is zero padding: not read
        inb[(size_t)r] = r < G / 2 ? b[r] : nullptr;
        spa[(size_t)r] = static_cast<uint32_t *>(gs.spec_a.ptr);
        spb[(size_t)r] = static_cast<uint32_t *>(gs.spec_b.ptr);
        spa_c[(size_t)r] = spa[(size_t)r];
    }
    int rc = group_dist_run(dp, ina.data(), spa.data(), false, true, true);
    if (!rc) rc = group_dist_run(dp, inb.data(), spb.data(), false, true, true);
    if (rc) return rc;
    for (int r = 0; r < G; ++r) {                            // behind the slot's own local transforms (stream order)
        ntt_ctx *c = g->sub[(size_t)r];
        ntt_plan *lp = dp->part[(size_t)r]->local;
        HIP_TRY(hipSetDevice(c->device));
        launch_dist_pointwise(lp->kind, lp->lazy, spa[(size_t)r], spb[(size_t)r], dp->S, lp->mp, c->stream);
        // the next run starts behind `fin` of every slot: move the mark behind the pointwise product
        HIP_TRY(hipEventRecord(dp->gslot[(size_t)r].fin, c->stream));
    }
    HIP_TRY(hipGetLastError());
    return group_dist_run(dp, spa_c.data(), ab, true, true);
}

}  // namespace ntt

using namespace ntt;

extern "C" {

int ntt_partition(long total_units, int world, int rank, long *start, long *count)
try {
    if (total_units < 0 || world < 1 || rank < 0 || rank >= world || !start || !count)
        return fail(NTT_ERR_INVALID, "ntt_partition: need total >= 0, 0 <= rank < world and two output pointers");
    partition(total_units, world, rank, start, count);
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_ctx_device_count(const ntt_ctx *ctx) { return !ctx ? 0 : ctx->sub.empty() ? 1 : (int)ctx->sub.size(); }

int ntt_ctx_device(ntt_ctx *ctx, int slot, ntt_ctx **out_slot_ctx, int *out_device)
try {
    if (!ctx) return fail(NTT_ERR_INVALID, "ntt_ctx_device: null context");
    const int nd = ntt_ctx_device_count(ctx);
    if (slot < 0 || slot >= nd) return fail(NTT_ERR_INVALID, "ntt_ctx_device: no such slot");
    ntt_ctx *c = ctx->sub.empty() ? ctx : ctx->sub[(size_t)slot];
    if (out_slot_ctx) *out_slot_ctx = c;
    if (out_device) *out_device = c->device;
    return NTT_OK;
}
NTT_CATCH_STATUS

static int dist_run_entry(ntt_dist_plan *dp, const uint32_t *const *in, uint32_t *const *out, bool inverse, const char *who)
{
    if (!dp || !in || !out) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument");
    if (dp->part.empty())
        return fail(NTT_ERR_INVALID, std::string(who) + ": the plan was made from a single-device context (one rank's pieces: "
                                     "ntt_dist_colpass / ntt_dist_local around the caller's exchanges); make it from a multi-device context");
    ntt_ctx *g = dp->ctx;
    for (int r = 0; r < dp->world; ++r) {
        if (!in[r] || !out[r]) return fail(NTT_ERR_INVALID, std::string(who) + ": null block pointer");
        if (in[r] == out[r]) return fail(NTT_ERR_INVALID, std::string(who) + ": in[r] and out[r] must be different arrays (peers write out[] while in[] is still being read)");
        const int di = pointer_device(in[r]), dout_dev = pointer_device(out[r]), want = g->sub[(size_t)r]->device;
        if (di == -1 || dout_dev == -1) return fail(NTT_ERR_INVALID, std::string(who) + ": blocks must be device arrays");
        if ((di >= 0 && di != want) || (dout_dev >= 0 && dout_dev != want))
            return fail(NTT_ERR_INVALID, std::string(who) + ": block " + std::to_string(r) + " must live on the device of slot " + std::to_string(r));
    }
    std::lock_guard<std::mutex> lk(dp->gmu);
    return group_dist_run(dp, in, out, inverse);
}

int ntt_dist_forward(ntt_dist_plan *plan, const uint32_t *const *in, uint32_t *const *out)
try {
    return dist_run_entry(plan, in, out, false, "ntt_dist_forward");
}
NTT_CATCH_STATUS

int ntt_dist_inverse(ntt_dist_plan *plan, const uint32_t *const *in, uint32_t *const *out)
try {
    return dist_run_entry(plan, in, out, true, "ntt_dist_inverse");
}
NTT_CATCH_STATUS

int ntt_poly_multiply_batched_slots(ntt_plan *pl, const uint32_t *const *a, const uint32_t *const *b, uint32_t *const *ab,
                                    size_t ab_stride, const int *count)
try {
    if (!pl || !a || !b || !ab || !count) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_slots: null argument");
    if (pl->shard.empty()) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_slots: needs a plan made from a multi-device context");
    if (pl->negacyclic) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_slots: not for negacyclic plans");
    ntt_ctx *g = pl->ctx;
    const int nd = (int)g->sub.size();
    for (int s = 0; s < nd; ++s) {
        if (count[s] < 0 || count[s] > pl->shard[(size_t)s]->batch)
            return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_slots: count[" + std::to_string(s) + "] exceeds the slot's share of the plan's capacity (ceil(batch / slots))");
        if (count[s] && (!a[s] || !b[s] || !ab[s])) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_slots: null array for a slot with work");
        int rc;
        if (count[s] && ((rc = check_reachable(g, a[s], s, "ntt_poly_multiply_batched_slots: a")) || (rc = check_reachable(g, b[s], s, "ntt_poly_multiply_batched_slots: b")) ||
                         (rc = check_reachable(g, ab[s], s, "ntt_poly_multiply_batched_slots: ab"))))
            return rc;
    }
    return for_each_slot(nd, [&](int s) -> int {
        if (!count[s]) return NTT_OK;
        return ntt_poly_multiply_batched(pl->shard[(size_t)s], a[s], b[s], ab[s], ab_stride, count[s]);
    });
}
NTT_CATCH_STATUS

int ntt_dist_multiply(ntt_dist_plan *dp, const uint32_t *const *a, const uint32_t *const *b, uint32_t *const *ab)
try {
    const char *who = "ntt_dist_multiply";
    if (!dp || !a || !b || !ab) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument");
    if (dp->part.empty()) return fail(NTT_ERR_INVALID, std::string(who) + ": needs a plan made from a multi-device context");
    ntt_ctx *g = dp->ctx;
    for (int r = 0; r < dp->world; ++r) {
        const int want = g->sub[(size_t)r]->device;
        const void *ptrs[3] = {r < dp->world / 2 ? a[r] : ab[r], r < dp->world / 2 ? b[r] : ab[r], ab[r]};
        for (const void *p : ptrs) {
            if (!p) return fail(NTT_ERR_INVALID, std::string(who) + ": null block pointer");
            const int d = pointer_device(p);
            if (d == -1) return fail(NTT_ERR_INVALID, std::string(who) + ": blocks must be device arrays");
            if (d >= 0 && d != want) return fail(NTT_ERR_INVALID, std::string(who) + ": block " + std::to_string(r) + " must live on the device of slot " + std::to_string(r));
        }
    }
    std::lock_guard<std::mutex> lk(dp->gmu);
    return group_dist_multiply(dp, a, b, ab);
}
NTT_CATCH_STATUS

int ntt_dist_plan_transport(const ntt_dist_plan *plan, int *transport, int *chunks)
try {
    if (!plan) return fail(NTT_ERR_INVALID, "ntt_dist_plan_transport: null plan");
    if (transport) *transport = plan->part.empty() ? -1 : plan->transport;
    if (chunks) *chunks = plan->part.empty() ? 0 : plan->chunks;
    return NTT_OK;
}
NTT_CATCH_STATUS

}  // extern "C"
