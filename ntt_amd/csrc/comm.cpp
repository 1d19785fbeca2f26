// comm.cpp -- ntt_comm: the exchanges of the sharded workloads for a host program that keeps ONE PROCESS PER GPU.
//
// The reference's distributed program is exactly that shape (main-mpi.cc:645-649: MPI_Init, rank, size; :685-688 every
// rank receives the case by MPI_Bcast; :657-660 rank r owns ceil(4 / size) primes; :770 multiplies their residues;
// :772-791 MPI_Send / MPI_Recv of the residue polynomials to rank 0; :793-796 rank 0 merges).  A maintainer who keeps
// that program and its launcher (mpirun -np N) links this library into every rank: each rank makes an ordinary
// single-device context on its GPU plus one ntt_comm, and the collective entry points below do what multidev.cpp does
// for one process over all GPUs -- the same kernels, the same data layouts -- with the exchange carried by
//   * RCCL (ntt_comm_create_rccl): grouped ncclSend / ncclRecv on device arrays over xGMI; the host program only moves
//     the 128-byte unique id from rank 0 to the others (MPI_Bcast).  The split transform is cut into column chunks so
//     that the exchange of chunk c + 1 (the communicator's stream) runs beside the column kernel of chunk c;
//   * or the host program's own all-to-all on host memory (ntt_comm_create_host; MPI_Alltoall in one line): the library
//     stages through pinned buffers.  The compatibility route -- it is what the reference's MPI_Send of host vectors
//     amounts to -- and the one a one-GPU box can rehearse with several ranks (RCCL refuses two ranks on one device).
#include <cstring>
#include <memory>

#include "../../include/ntt_amd.h"
#include "engine.hpp"
#include "rccl_dl.hpp"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct ntt_comm {
    ntt_ctx *ctx = nullptr;
    int world = 1, rank = 0;
    bool rccl = false;
    ntt_alltoall_fn fn = nullptr;
    void *user = nullptr;
    void *nccl = nullptr;                                  // ncclComm_t
    hipStream_t xs = nullptr;                              // exchange stream (RCCL mode)
    hipEvent_t start = nullptr, out_done = nullptr;
    std::vector<hipEvent_t> ev_in, ev_k;                   // per column chunk
    ntt::DeviceBuffer row, held, sent;                     // split transform: [S] each
    ntt::DeviceBuffer spec_a, spec_b;                      // split product: the two spectra of this rank
    void *host_send = nullptr, *host_recv = nullptr;       // pinned staging (host mode)
    size_t host_bytes = 0;
    int crt_n = 0;                                         // large-modulus product: buffers sized for this n
    ntt::DeviceBuffer crt_send, crt_recv, crt_mine, crt_full;
    std::mutex mu;
};

namespace ntt {

static constexpr int kCommChunks = 4;

static int ensure_host(ntt_comm *cm, size_t bytes)
{
    if (bytes <= cm->host_bytes) return NTT_OK;
    if (cm->host_send) (void)hipHostFree(cm->host_send);
    if (cm->host_recv) (void)hipHostFree(cm->host_recv);
    cm->host_send = cm->host_recv = nullptr;
    cm->host_bytes = 0;
    HIP_TRY(hipHostMalloc(&cm->host_send, bytes, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&cm->host_recv, bytes, hipHostMallocDefault));
    cm->host_bytes = bytes;
    return NTT_OK;
}

static int call_alltoall(ntt_comm *cm, size_t bytes_per_rank)
{
    const int e = cm->fn(cm->user, cm->host_send, cm->host_recv, bytes_per_rank);
    if (e) return fail(NTT_ERR_IO, "ntt_comm: the host program's all-to-all returned " + std::to_string(e));
    return NTT_OK;
}

// recv[i][c0 .. c0 + cc) <- columns [c0, c0 + cc) of block `rank` of rank i's `send`, for two [world][cols] u32 arrays:
// the all-to-all of the distributed 4-step (include/ntt_amd.h, ntt_dist_*).  RCCL: queued on `stream`.  Host mode:
// through the pinned buffers and the callback, synchronously; the copy back to the device is queued on `stream`.
// nz < world: only ranks below nz hold data (the zero-padded half of a product's forward transform) -- the others send
// nothing, and nothing is received from them (their rows of recv are left as they are: the HALF kernel does not read them).
static int exchange_blocks(ntt_comm *cm, const uint32_t *send, uint32_t *recv, size_t cols, size_t c0, size_t cc, hipStream_t stream, int nz)
{
    const int W = cm->world;
    const bool sender = cm->rank < nz;
    if (cm->rccl) {
        Rccl *r = rccl_load();
        if (!r) return fail(NTT_ERR_INTERNAL, "ntt_comm: librccl went away");
        RCCL_TRY(r, r->GroupStart());
        for (int i = 0; i < W; ++i) {
            if (sender) RCCL_TRY(r, r->Send(send + (size_t)i * cols + c0, cc, Rccl::kUint32, i, cm->nccl, stream));
            if (i < nz) RCCL_TRY(r, r->Recv(recv + (size_t)i * cols + c0, cc, Rccl::kUint32, i, cm->nccl, stream));
        }
        RCCL_TRY(r, r->GroupEnd());
        return NTT_OK;
    }
    const size_t bytes = cc * sizeof(uint32_t);
    if (int rc = ensure_host(cm, bytes * (size_t)W)) return rc;
    if (sender) HIP_TRY(hipMemcpy2DAsync(cm->host_send, bytes, send + c0, cols * sizeof(uint32_t), bytes, (size_t)W, hipMemcpyDeviceToHost, stream));
    else std::memset(cm->host_send, 0, bytes * (size_t)W);   // the host program's all-to-all moves every block
    HIP_TRY(hipStreamSynchronize(stream));
    if (int rc = call_alltoall(cm, bytes)) return rc;
    HIP_TRY(hipMemcpy2DAsync(recv + c0, cols * sizeof(uint32_t), cm->host_recv, bytes, bytes, (size_t)nz, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));                   // host_recv is free again for the next exchange
    return NTT_OK;
}

// scrambled: the local transforms leave / take the spectrum in the engine's own order; half (forward only): the blocks of
// ranks >= world / 2 are zero padding (`in` is not read there) -- the two halves of ntt_comm_dist_multiply
static int comm_dist_run(ntt_comm *cm, ntt_dist_plan *dp, const uint32_t *in, uint32_t *out, bool inverse, bool scrambled = false,
                         bool half = false)
{
    ntt_ctx *cx = cm->ctx;
    const int W = cm->world;
    const uint32_t S = dp->S, cols = S / (uint32_t)W;
    const size_t bytes = (size_t)S * sizeof(uint32_t);
    const int nz = (half && !inverse) ? W / 2 : W;          // ranks whose block holds data
    HIP_TRY(hipSetDevice(cx->device));
    int rc = cm->held.ensure(bytes);
    if (!rc) rc = cm->sent.ensure(bytes);
    if (!rc && inverse) rc = cm->row.ensure(bytes);
    if (rc) return rc;
    hipStream_t cs = cx->stream;
    uint32_t *held = static_cast<uint32_t *>(cm->held.ptr), *sent = static_cast<uint32_t *>(cm->sent.ptr);
    const uint32_t *first = in;
    if (inverse) {                                           // local S-point inverse transforms first: in -> row
        uint32_t *rowp = static_cast<uint32_t *>(cm->row.ptr);
        if ((rc = transform_device_any(dp->local, in, rowp, true, scrambled))) return rc;
        first = rowp;
    }
    const uint32_t *src[kMaxDistWorld];
    uint32_t *dst[kMaxDistWorld];
    for (int i = 0; i < W; ++i) {
        src[i] = held + (size_t)i * cols;
        dst[i] = sent + (size_t)i * cols;
    }
    int C = cm->rccl ? kCommChunks : 1;                      // host mode is synchronous: nothing to overlap
    while (C > 1 && (cols % ((uint32_t)C * 256u)) != 0) --C;
    const uint32_t cc = cols / (uint32_t)C;
    if (!cm->rccl) {
        if ((rc = exchange_blocks(cm, first, held, cols, 0, cols, cs, nz))) return rc;
        if ((rc = dist_colpass_tables(dp, inverse, half, src, dst, 0, cols, cs))) return rc;
        if ((rc = exchange_blocks(cm, sent, out, cols, 0, cols, cs, W))) return rc;
    } else {
        // exchange stream xs beside compute stream cs: in(0) | in(1), colpass(0) | out(0), in(2), colpass(1) | ...
        hipStream_t xs = cm->xs;
        HIP_TRY(hipEventRecord(cm->start, cs));
        HIP_TRY(hipStreamWaitEvent(xs, cm->start, 0));
        if ((rc = exchange_blocks(cm, first, held, cols, 0, cc, xs, nz))) return rc;
        HIP_TRY(hipEventRecord(cm->ev_in[0], xs));
        for (int ch = 0; ch < C; ++ch) {
            const uint32_t c0 = (uint32_t)ch * cc;
            if (ch + 1 < C) {
                if ((rc = exchange_blocks(cm, first, held, cols, c0 + cc, cc, xs, nz))) return rc;
                HIP_TRY(hipEventRecord(cm->ev_in[(size_t)ch + 1], xs));
            }
            HIP_TRY(hipStreamWaitEvent(cs, cm->ev_in[(size_t)ch], 0));
            if ((rc = dist_colpass_tables(dp, inverse, half, src, dst, c0, cc, cs))) return rc;
            HIP_TRY(hipEventRecord(cm->ev_k[(size_t)ch], cs));
            HIP_TRY(hipStreamWaitEvent(xs, cm->ev_k[(size_t)ch], 0));
            if ((rc = exchange_blocks(cm, sent, out, cols, c0, cc, xs, W))) return rc;
        }
        HIP_TRY(hipEventRecord(cm->out_done, xs));
        HIP_TRY(hipStreamWaitEvent(cs, cm->out_done, 0));
    }
    HIP_TRY(hipGetLastError());
    if (!inverse) return transform_device_any(dp->local, out, out, false, scrambled);
    return NTT_OK;
}

static int comm_crt_buffers(ntt_comm *cm, int n)
{
    if (cm->crt_n == n) return NTT_OK;
    const int W = cm->world, mpp = (4 + W - 1) / W;
    const size_t sl = (2 * (size_t)n - 1 + (size_t)W - 1) / (size_t)W;
    cm->crt_n = 0;
    HIP_TRY(hipSetDevice(cm->ctx->device));
    int rc = cm->crt_send.ensure((size_t)mpp * W * sl * sizeof(uint32_t));
    if (!rc) rc = cm->crt_recv.ensure((size_t)std::max(4, W * mpp) * sl * sizeof(uint32_t));
    if (!rc) rc = cm->crt_mine.ensure(sl * sizeof(uint64_t));
    if (!rc) rc = cm->crt_full.ensure((size_t)W * sl * sizeof(uint64_t));
    if (rc) return rc;
    // a residue product fills 2n - 1 words of its [world * sl] row: the zero tail is written once, here
    HIP_TRY(hipMemsetAsync(cm->crt_send.ptr, 0, (size_t)mpp * W * sl * sizeof(uint32_t), cm->ctx->stream));
    cm->crt_n = n;
    return NTT_OK;
}

static int comm_crt_multiply(ntt_comm *cm, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64)
{
    ntt_ctx *cx = cm->ctx;
    const int W = cm->world, mpp = (4 + W - 1) / W;
    const size_t out_len = 2 * (size_t)n - 1, sl = (out_len + (size_t)W - 1) / (size_t)W, row = (size_t)W * sl;
    int rc = comm_crt_buffers(cm, n);
    if (rc) return rc;
    hipStream_t cs = cx->stream;
    uint32_t *send = static_cast<uint32_t *>(cm->crt_send.ptr), *recv = static_cast<uint32_t *>(cm->crt_recv.ptr);
    uint64_t *mine = static_cast<uint64_t *>(cm->crt_mine.ptr), *full = static_cast<uint64_t *>(cm->crt_full.ptr);
    long start, count;
    partition(4, W, cm->rank, &start, &count);               // main-mpi.cc:657-660
    // step 1 (main-mpi.cc:770): the residue products of this rank's primes, written in exchange order
    for (long j = 0; j < count; ++j)
        if ((rc = ntt_crt_residue_multiply(cx, a, b, send + (size_t)j * row, n, (int)(start + j)))) return rc;
    HIP_TRY(hipSetDevice(cx->device));
    // step 2 (instead of :772-791's gather to rank 0): slice h of every residue polynomial goes to rank h
    const bool ab_dev = is_device_ptr(ab);
    if (cm->rccl) {
        Rccl *r = rccl_load();
        if (!r) return fail(NTT_ERR_INTERNAL, "ntt_comm: librccl went away");
        RCCL_TRY(r, r->GroupStart());
        for (int h = 0; h < W; ++h) {
            for (long j = 0; j < count; ++j)
                RCCL_TRY(r, r->Send(send + (size_t)j * row + (size_t)h * sl, sl, Rccl::kUint32, h, cm->nccl, cs));
            long so, co;
            partition(4, W, h, &so, &co);
            for (long j = 0; j < co; ++j)
                RCCL_TRY(r, r->Recv(recv + (size_t)(so + j) * sl, sl, Rccl::kUint32, h, cm->nccl, cs));
        }
        RCCL_TRY(r, r->GroupEnd());
    } else {
        // pinned send array [h][j][sl]; what comes back is [owner][j][sl] = prime owner * mpp + j at that index
        const size_t blk = (size_t)mpp * sl * sizeof(uint32_t);
        if ((rc = ensure_host(cm, std::max(blk * (size_t)W, (size_t)W * sl * sizeof(uint64_t))))) return rc;
        if (count < mpp) std::memset(cm->host_send, 0, blk * (size_t)W);
        for (long j = 0; j < count; ++j)
            HIP_TRY(hipMemcpy2DAsync(static_cast<char *>(cm->host_send) + (size_t)j * sl * sizeof(uint32_t), blk, send + (size_t)j * row,
                                     sl * sizeof(uint32_t), sl * sizeof(uint32_t), (size_t)W, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipStreamSynchronize(cs));
        if ((rc = call_alltoall(cm, blk))) return rc;
        HIP_TRY(hipMemcpyAsync(recv, cm->host_recv, 4 * sl * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
    }
    // step 3 (main-mpi.cc:793-796, on every rank instead of rank 0): merge this rank's slice
    if ((rc = ntt_crt_merge(cx, recv, recv + sl, recv + 2 * sl, recv + 3 * sl, mine, sl, p64))) return rc;
    HIP_TRY(hipSetDevice(cx->device));
    // step 4: every rank ends with the whole product (the reference: rank 0 only)
    if (cm->rccl) {
        Rccl *r = rccl_load();
        RCCL_TRY(r, r->GroupStart());
        for (int h = 0; h < W; ++h) {
            RCCL_TRY(r, r->Send(mine, 2 * sl, Rccl::kUint32, h, cm->nccl, cs));
            RCCL_TRY(r, r->Recv(full + (size_t)h * sl, 2 * sl, Rccl::kUint32, h, cm->nccl, cs));
        }
        RCCL_TRY(r, r->GroupEnd());
        HIP_TRY(hipMemcpyAsync(ab, full, out_len * sizeof(uint64_t), hipMemcpyDefault, cs));
        if (!ab_dev) HIP_TRY(hipStreamSynchronize(cs));
    } else {
        const size_t blk = sl * sizeof(uint64_t);
        HIP_TRY(hipMemcpyAsync(cm->host_send, mine, blk, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipStreamSynchronize(cs));
        for (int h = 1; h < W; ++h) std::memcpy(static_cast<char *>(cm->host_send) + (size_t)h * blk, cm->host_send, blk);
        if ((rc = call_alltoall(cm, blk))) return rc;
        if (ab_dev) {
            HIP_TRY(hipMemcpyAsync(ab, cm->host_recv, out_len * sizeof(uint64_t), hipMemcpyHostToDevice, cs));
            HIP_TRY(hipStreamSynchronize(cs));               // host_recv is reused by the next call
        } else {
            std::memcpy(ab, cm->host_recv, out_len * sizeof(uint64_t));
        }
    }
    return NTT_OK;
}

static int comm_new(ntt_ctx *ctx, int world, int rank, const char *who, std::unique_ptr<ntt_comm> &cm)
{
    if (!ctx) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument");
    if (ctx->is_group()) return fail(NTT_ERR_INVALID, std::string(who) + ": a communicator joins single-device contexts of separate processes; a multi-device context exchanges by itself");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(NTT_ERR_INVALID, std::string(who) + ": need 1 <= world <= 64 and 0 <= rank < world");
    cm.reset(new ntt_comm());
    cm->ctx = ctx;
    cm->world = world;
    cm->rank = rank;
    return NTT_OK;
}

static void comm_release(ntt_comm *cm)
{
    (void)hipSetDevice(cm->ctx->device);
    (void)hipStreamSynchronize(cm->ctx->stream);
    if (cm->xs) { (void)hipStreamSynchronize(cm->xs); (void)hipStreamDestroy(cm->xs); }
    for (hipEvent_t e : {cm->start, cm->out_done}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : cm->ev_in) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : cm->ev_k) if (e) (void)hipEventDestroy(e);
    if (cm->nccl)
        if (Rccl *r = rccl_load()) (void)r->CommDestroy(cm->nccl);
    cm->row.release(); cm->held.release(); cm->sent.release();
    cm->spec_a.release(); cm->spec_b.release();
    cm->crt_send.release(); cm->crt_recv.release(); cm->crt_mine.release(); cm->crt_full.release();
    if (cm->host_send) (void)hipHostFree(cm->host_send);
    if (cm->host_recv) (void)hipHostFree(cm->host_recv);
}

}  // namespace ntt

using namespace ntt;

extern "C" {

int ntt_comm_rccl_id(void *id)
try {
    if (!id) return fail(NTT_ERR_INVALID, "ntt_comm_rccl_id: null argument");
    Rccl *r = rccl_load();
    if (!r) return fail(NTT_ERR_UNSUPPORTED, "ntt_comm_rccl_id: librccl.so could not be loaded");
    static_assert(sizeof(RcclUniqueId) == NTT_COMM_ID_BYTES, "unique id size");
    RcclUniqueId uid{};
    RCCL_TRY(r, r->GetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_comm_create_rccl(ntt_ctx *ctx, const void *id, int world, int rank, ntt_comm **out)
try {
    if (!id || !out) return fail(NTT_ERR_INVALID, "ntt_comm_create_rccl: null argument");
    *out = nullptr;
    std::unique_ptr<ntt_comm> cm;
    if (int rc = comm_new(ctx, world, rank, "ntt_comm_create_rccl", cm)) return rc;
    Rccl *r = rccl_load();
    if (!r) return fail(NTT_ERR_UNSUPPORTED, "ntt_comm_create_rccl: librccl.so could not be loaded");
    struct Undo {
        ntt_comm *cm;
        bool armed = true;
        ~Undo() { if (armed) comm_release(cm); }
    } undo{cm.get()};
    cm->rccl = true;
    HIP_TRY(hipSetDevice(ctx->device));
    RcclUniqueId uid{};
    std::memcpy(&uid, id, sizeof(uid));
    const int e = r->CommInitRank(&cm->nccl, world, uid, rank);
    if (e != 0) {
        cm->nccl = nullptr;
        return fail(NTT_ERR_UNSUPPORTED, std::string("ntt_comm_create_rccl: ncclCommInitRank: ") + r->GetErrorString(e));
    }
    HIP_TRY(hipStreamCreateWithFlags(&cm->xs, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&cm->start, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&cm->out_done, hipEventDisableTiming));
    cm->ev_in.assign((size_t)kCommChunks, nullptr);
    cm->ev_k.assign((size_t)kCommChunks, nullptr);
    for (int c = 0; c < kCommChunks; ++c) {
        HIP_TRY(hipEventCreateWithFlags(&cm->ev_in[(size_t)c], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&cm->ev_k[(size_t)c], hipEventDisableTiming));
    }
    undo.armed = false;
    *out = cm.release();
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_comm_create_host(ntt_ctx *ctx, ntt_alltoall_fn alltoall, void *user, int world, int rank, ntt_comm **out)
try {
    if (!alltoall || !out) return fail(NTT_ERR_INVALID, "ntt_comm_create_host: null argument");
    *out = nullptr;
    std::unique_ptr<ntt_comm> cm;
    if (int rc = comm_new(ctx, world, rank, "ntt_comm_create_host", cm)) return rc;
    cm->fn = alltoall;
    cm->user = user;
    *out = cm.release();
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_comm_destroy(ntt_comm *comm)
try {
    if (!comm) return NTT_OK;
    comm_release(comm);
    delete comm;
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_comm_size(const ntt_comm *comm, int *world, int *rank, int *kind)
try {
    if (!comm) return fail(NTT_ERR_INVALID, "ntt_comm_size: null communicator");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    if (kind) *kind = comm->rccl ? 1 : 0;
    return NTT_OK;
}
NTT_CATCH_STATUS

static int comm_dist_entry(ntt_comm *cm, ntt_dist_plan *dp, const uint32_t *in, uint32_t *out, bool inverse, const char *who)
{
    if (!cm || !dp || !in || !out) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument");
    if (!dp->part.empty() || dp->ctx != cm->ctx)
        return fail(NTT_ERR_INVALID, std::string(who) + ": the plan must come from ntt_dist_plan_create on the communicator's own single-device context");
    if (dp->world != cm->world || dp->rank != cm->rank)
        return fail(NTT_ERR_INVALID, std::string(who) + ": the plan's world / rank differ from the communicator's");
    if (dp->world > kMaxDistWorld) return fail(NTT_ERR_UNSUPPORTED, std::string(who) + ": at most 16 ranks");
    if (in == out) return fail(NTT_ERR_INVALID, std::string(who) + ": in and out must be different arrays");
    if (!is_device_ptr(in) || !is_device_ptr(out)) return fail(NTT_ERR_INVALID, std::string(who) + ": in and out must be device arrays of S = L / world words");
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_dist_run(cm, dp, in, out, inverse);
}

int ntt_comm_dist_forward(ntt_comm *comm, ntt_dist_plan *plan, const uint32_t *in, uint32_t *out)
try {
    return comm_dist_entry(comm, plan, in, out, false, "ntt_comm_dist_forward");
}
NTT_CATCH_STATUS

int ntt_comm_dist_inverse(ntt_comm *comm, ntt_dist_plan *plan, const uint32_t *in, uint32_t *out)
try {
    return comm_dist_entry(comm, plan, in, out, true, "ntt_comm_dist_inverse");
}
NTT_CATCH_STATUS

int ntt_comm_dist_multiply(ntt_comm *cm, ntt_dist_plan *dp, const uint32_t *a, const uint32_t *b, uint32_t *ab)
try {
    const char *who = "ntt_comm_dist_multiply";
    if (!cm || !dp || !ab) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument");
    if (!dp->part.empty() || dp->ctx != cm->ctx)
        return fail(NTT_ERR_INVALID, std::string(who) + ": the plan must come from ntt_dist_plan_create on the communicator's own single-device context");
    if (dp->world != cm->world || dp->rank != cm->rank) return fail(NTT_ERR_INVALID, std::string(who) + ": the plan's world / rank differ from the communicator's");
    if (dp->world > kMaxDistWorld) return fail(NTT_ERR_UNSUPPORTED, std::string(who) + ": at most 16 ranks");
    const bool holds = cm->rank < cm->world / 2;             // the upper half of both operands is the zero padding
    if (holds && (!a || !b)) return fail(NTT_ERR_INVALID, std::string(who) + ": ranks below world / 2 pass their block of a and of b");
    if ((holds && (!is_device_ptr(a) || !is_device_ptr(b))) || !is_device_ptr(ab))
        return fail(NTT_ERR_INVALID, std::string(who) + ": blocks must be device arrays of S = L / world words");
    std::lock_guard<std::mutex> lk(cm->mu);
    HIP_TRY(hipSetDevice(cm->ctx->device));
    const size_t bytes = (size_t)dp->S * sizeof(uint32_t);
    int rc = cm->spec_a.ensure(bytes);
    if (!rc) rc = cm->spec_b.ensure(bytes);
    if (rc) return rc;
    uint32_t *spa = static_cast<uint32_t *>(cm->spec_a.ptr), *spb = static_cast<uint32_t *>(cm->spec_b.ptr);
    if ((rc = comm_dist_run(cm, dp, holds ? a : nullptr, spa, false, true, true))) return rc;
    if ((rc = comm_dist_run(cm, dp, holds ? b : nullptr, spb, false, true, true))) return rc;
    ntt_plan *lp = dp->local;
    HIP_TRY(hipSetDevice(cm->ctx->device));
    launch_dist_pointwise(lp->kind, lp->lazy, spa, spb, dp->S, lp->mp, cm->ctx->stream);     // behind the local transforms (stream order)
    HIP_TRY(hipGetLastError());
    return comm_dist_run(cm, dp, spa, ab, true, true);
}
NTT_CATCH_STATUS

int ntt_comm_crt_multiply(ntt_comm *cm, const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64)
try {
    if (!cm || !a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_comm_crt_multiply: null argument");
    if (n <= 0) return fail(NTT_ERR_INVALID, "ntt_comm_crt_multiply: n must be positive");
    // the reference's constraints (ntt_crt_multiply checks the same): every rank must reject together, before any exchange
    if ((p64 & 1) == 0 || p64 < 3 || p64 >= (1ULL << 62)) return fail(NTT_ERR_UNSUPPORTED, "ntt_comm_crt_multiply: p64 must be odd and below 2^62");
    if (n > (1 << 22)) return fail(NTT_ERR_UNSUPPORTED, "ntt_comm_crt_multiply: n too large for the 2^23-adic prime 998244353");
    {
        unsigned __int128 M = 1;
        for (int k = 0; k < 4; ++k) { uint32_t m = 0; ntt_crt_prime(k, &m); M *= m; }
        const unsigned __int128 sq = (unsigned __int128)(p64 - 1) * (p64 - 1);
        if (sq > M / (unsigned)n) return fail(NTT_ERR_UNSUPPORTED, "ntt_comm_crt_multiply: n * p64^2 exceeds the product of the four CRT primes");
    }
    std::lock_guard<std::mutex> lk(cm->mu);
    return comm_crt_multiply(cm, a, b, ab, n, p64);
}
NTT_CATCH_STATUS

}  // extern "C"
