// engine.cpp -- contexts, plans (root search, twiddle tables, work buffers) and the launch
// sequences of the NTT multiply path.  Host C++ over the HIP runtime; the kernels live in
// kernels_*.hip.  See include/ntt_amd.h for the reference interface each entry point replaces.
#include "engine.hpp"

#include <algorithm>
#include <exception>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "../../include/ntt_amd.h"

// ---- scratch of ntt_poly_multiply_batched_noalloc -----------------------------------------------------
// The one entry point that allocates nothing keeps the intermediates of the 4-step passes (the role of
// ntt_plan::work0 / work1) in this file-scope device array instead of hipMalloc'ed buffers.  Every GPU the
// library runs on holds its own copy, at its own address, until the process ends.  In host code the
// array's NAME is only a stand-in for the symbol: its device address is asked for once per device
// (hipGetSymbolAddress in ntt_ctx_create, that device current) and kept in ntt_ctx::scratch; kernels,
// like every other use of the memory, get that pointer, never the name.
static constexpr size_t kScratchBytes = (size_t)64 << 20;
static constexpr size_t kZeroCopyBytes = (size_t)1 << 20;     // per operand: pinned host arrays up to this size are used in place
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
// Every context on a GPU shares that GPU's copy of the array.  Calls are ordered per DEVICE, not per context: one
// event per device marks the end of the last noalloc call there (created in ntt_ctx_create, so the call itself still
// creates nothing); a call makes its stream wait on it before the first kernel touches the array and records it behind
// the last one.  g_scratch_mu keeps the wait / launch / record sequences of two host threads from interleaving.
static constexpr int kMaxScratchDevices = 64;
static std::mutex g_scratch_mu;
static hipEvent_t g_scratch_last[kMaxScratchDevices] = {};

namespace ntt {

const char *const kKernelNames[K_COUNT] = {"col_fwd", "row_mul", "col_inv", "row_fwd", "row_inv",
                                           "small", "permute", "crt_reduce", "crt_merge",
                                           "col64", "lds64"};

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// The C ABI lets no exception out (NTT_CATCH_STATUS closes every entry point): called inside a catch block,
// turns the exception in flight into a status code and a message.
int exception_status() noexcept
{
    try {
        try {
            throw;
        } catch (const std::bad_alloc &) {
            return fail(NTT_ERR_NOMEM, "out of host memory");
        } catch (const std::exception &e) {
            return fail(NTT_ERR_INTERNAL, std::string("unexpected exception: ") + e.what());
        } catch (...) {
            return fail(NTT_ERR_INTERNAL, "unexpected exception");
        }
    } catch (...) {               // not even the message could be stored
        return NTT_ERR_NOMEM;
    }
}
const char *last_error_cstr() { return g_err.c_str(); }

static thread_local std::string g_bad_launch;
void note_unsupported_launch(const char *what, int value)
{
    if (g_bad_launch.empty()) g_bad_launch = std::string(what) + " " + std::to_string(value);
}
int take_launch_error()
{
    if (g_bad_launch.empty()) return NTT_OK;
    std::string msg;
    msg.swap(g_bad_launch);
    return fail(NTT_ERR_INTERNAL, "kernel launch skipped: " + msg);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

// Wait for the context's stream.  (Round 3 tried polling hipStreamQuery for a while before blocking, to shorten the
// harness's one-product-and-wait loop, main-montgomery.cc:336-346: every query costs microseconds of its own and the
// n = 4 case went from 13.6 to 30 us -- hipStreamSynchronize already waits actively; profiles/latency_r03.txt.)
int sync_stream(ntt_ctx *cx)
{
    HIP_TRY(hipStreamSynchronize(cx->stream));
    return NTT_OK;
}

int DeviceBuffer::ensure(size_t need)
{
    if (need <= bytes) return NTT_OK;
    if (ptr) {
        HIP_TRY(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    hipError_t e = hipMalloc(&ptr, need);
    if (e != hipSuccess) {
        ptr = nullptr;
        return fail(e == hipErrorOutOfMemory ? NTT_ERR_NOMEM : NTT_ERR_HIP,
                    std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    bytes = need;
    return NTT_OK;
}
void DeviceBuffer::release()
{
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}
int PinnedBuffer::ensure(size_t need)
{
    if (need <= bytes) return NTT_OK;
    release();
    hipError_t e = hipHostMalloc(&ptr, need, hipHostMallocDefault);
    if (e != hipSuccess) {
        ptr = nullptr;
        return fail(e == hipErrorOutOfMemory ? NTT_ERR_NOMEM : NTT_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    bytes = need;
    return NTT_OK;
}
void PinnedBuffer::release()
{
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

// Device or host?  hipPointerGetAttributes knows the allocations of THIS HIP runtime.  A pointer
// from another HIP runtime in the same process (PyTorch bundles its own libamdhip64; both sit on
// one HSA runtime) is unknown to it, so fall back to the HSA pointer registry, which is shared.
// device memory by the HSA runtime's registry: what another HIP runtime in the process allocated is known there
static bool hsa_device_ptr(const void *p)
{
    hsa_amd_pointer_info_t info{};
    info.size = sizeof(info);
    if (hsa_amd_pointer_info(const_cast<void *>(p), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS)
        return false;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return false;
    hsa_device_type_t dt = HSA_DEVICE_TYPE_CPU;
    if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS) return false;
    return dt == HSA_DEVICE_TYPE_GPU;
}

bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t at{};
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e == hipSuccess && (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged)) return true;
    if (e != hipSuccess) (void)hipGetLastError();   // unregistered memory reports an error: clear it
    return hsa_device_ptr(p);
}

// Page-locked host memory this HIP runtime knows (hipHostMalloc / hipHostRegister): device-accessible at the same address
bool is_pinned_host_ptr(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
}

// ---- table construction -------------------------------------------------------------------------
template <class MM>
static int upload_table(ntt_plan *pl, const std::vector<typename MM::tw_t> &h, void **dptr)
{
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, h.size() * sizeof(typename MM::tw_t)));
    pl->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(typename MM::tw_t), hipMemcpyHostToDevice));
    *dptr = d;
    return NTT_OK;
}

// T[k][i] = c * w^(i*k), k < rows, i < cols
template <class MM>
static std::vector<typename MM::tw_t> power_table(uint32_t w, uint32_t rows, uint32_t cols, uint32_t c,
                                                  const ModParams &mp)
{
    std::vector<typename MM::tw_t> t((size_t)rows * cols);
    uint32_t wk = 1;                                   // w^k
    for (uint32_t k = 0; k < rows; ++k) {
        uint32_t cur = c;
        for (uint32_t i = 0; i < cols; ++i) {
            t[(size_t)k * cols + i] = MM::make_tw(cur, mp);
            cur = h_mulmod(cur, wk, mp.p);
        }
        wk = h_mulmod(wk, w, mp.p);
    }
    return t;
}

// The second table of a fused forward row pass (reg_ntt16_tw, device_ntt.hpp): for the q-th output of the difference
// side, k = 4, 5, 6, 7, 12, 13, 14, 15, entry [q][i] is the pair { w^(i*k), rot * w^(i*k) } with rot = i4 for k < 8 and
// -i4 above (i4 = the primitive fourth root the transform uses), both as CENTRED Montgomery forms in (-p / 2, p / 2]:
// one 8-byte load per output.
static std::vector<uint32_t> rotated_table(uint32_t w, uint32_t cols, uint32_t i4, const ModParams &mp)
{
    std::vector<uint32_t> t((size_t)16 * cols);
    for (uint32_t q = 0; q < 8; ++q) {
        const uint32_t k = q < 4 ? q + 4 : q + 8;
        const uint32_t wk = h_powmod(w, k, mp.p), rot = q < 4 ? i4 : mp.p - i4;
        uint32_t cur = 1;
        for (uint32_t i = 0; i < cols; ++i) {
            t[2 * ((size_t)q * cols + i)] = (uint32_t)Mont32::center_tw(Mont32::make_tw(cur, mp), mp);
            t[2 * ((size_t)q * cols + i) + 1] = (uint32_t)Mont32::center_tw(Mont32::make_tw(h_mulmod(cur, rot, mp.p), mp), mp);
            cur = h_mulmod(cur, wk, mp.p);
        }
    }
    return t;
}

// The pair table of the second fused INVERSE pass: its twiddles used to follow the exchange (reader t multiplied its
// k-th value by w^(k * t)); in front of it, thread (b, i) multiplies its output j by w^(b * (16 j + i)).  Entry
// (b, q, i) is the pair for the q-th output of the difference side, j = k_q.
static std::vector<uint32_t> rotated_table_inv(uint32_t w, uint32_t i4, const ModParams &mp)
{
    std::vector<uint32_t> t((size_t)16 * 8 * 16 * 2);
    for (uint32_t b = 0; b < 16; ++b)
        for (uint32_t q = 0; q < 8; ++q) {
            const uint32_t k = q < 4 ? q + 4 : q + 8, rot = q < 4 ? i4 : mp.p - i4;
            for (uint32_t i = 0; i < 16; ++i) {
                const uint32_t e = h_powmod(w, (uint64_t)b * (16 * k + i), mp.p);
                const size_t at = 2 * (((size_t)b * 8 + q) * 16 + i);
                t[at] = (uint32_t)Mont32::center_tw(Mont32::make_tw(e, mp), mp);
                t[at + 1] = (uint32_t)Mont32::center_tw(Mont32::make_tw(h_mulmod(e, rot, mp.p), mp), mp);
            }
        }
    return t;
}

template <class MM>
static std::vector<unsigned char> bytes_of_tw(uint32_t v, const ModParams &mp)
{
    typename MM::tw_t t = MM::make_tw(v, mp);
    std::vector<unsigned char> b(sizeof(t));
    std::memcpy(b.data(), &t, sizeof(t));
    return b;
}

template <class MM>
static int build_tables(ntt_plan *pl)
{
    const ModParams &mp = pl->mp;
    const uint32_t p = pl->p, L = pl->L;
    const uint32_t wL = h_powmod(pl->g, (p - 1) / L, p);
    const uint32_t iwL = h_powmod(wL, p - 2, p);
    const uint32_t invL = h_powmod(L % p, p - 2, p);
    // Montgomery's pointwise product carries R^-1: fold R into the multiply-path scale
    const uint32_t scale_mul = MM::kMontgomery ? h_mulmod(invL, mp.r1, p) : invL;
    pl->scale_mul = bytes_of_tw<MM>(scale_mul, mp);
    pl->scale_plain = bytes_of_tw<MM>(invL, mp);

    if (pl->negacyclic) {
        // psi = g^((p-1)/2L), psi^2 = wL: a * psi^i, cyclic transform of length L = n, result * psi^-i
        const uint32_t psi = h_powmod(pl->g, (p - 1) / (2 * L), p), ipsi = h_powmod(psi, p - 2, p);
        int rc = upload_table<MM>(pl, power_table<MM>(psi, 2, L, 1, mp), &pl->twist_fwd);      // row 1 = psi^i
        if (rc) return rc;
        pl->twist_fwd = static_cast<typename MM::tw_t *>(pl->twist_fwd) + L;
        rc = upload_table<MM>(pl, power_table<MM>(ipsi, 2, L, 1, mp), &pl->twist_inv);
        if (rc) return rc;
        pl->twist_inv = static_cast<typename MM::tw_t *>(pl->twist_inv) + L;
        rc = upload_table<MM>(pl, power_table<MM>(ipsi, 2, L, scale_mul, mp), &pl->small_itwist);
        if (rc) return rc;
        pl->small_itwist = static_cast<typename MM::tw_t *>(pl->small_itwist) + L;
    }
    if (pl->logL <= kRowLog) {
        std::vector<typename MM::tw_t> w(std::max<uint32_t>(L / 2, 1)), iw(std::max<uint32_t>(L / 2, 1));
        uint32_t c = 1, ic = 1;
        for (uint32_t j = 0; j < L / 2; ++j) {
            w[j] = MM::make_tw(c, mp);
            iw[j] = MM::make_tw(ic, mp);
            c = h_mulmod(c, wL, p);
            ic = h_mulmod(ic, iwL, p);
        }
        int rc = upload_table<MM>(pl, w, &pl->small_w);
        if (rc) return rc;
        rc = upload_table<MM>(pl, iw, &pl->small_iw);
        if (rc || pl->logL < kRowLog) return rc;
        // L = 4096: products go through the row kernel (row_product_kernel), which needs the roots
        // and row tables below; ntt_forward / ntt_inverse keep the small kernel
    }

    // roots of the 64-point register transform
    Roots64<MM> r{}, ir{};
    const uint32_t w64 = h_powmod(wL, L / 64, p), iw64 = h_powmod(iwL, L / 64, p);
    uint32_t c = 1, ic = 1;
    for (int j = 0; j < 32; ++j) {
        r.w[j] = MM::make_tw(c, mp);
        ir.w[j] = MM::make_tw(ic, mp);
        c = h_mulmod(c, w64, p);
        ic = h_mulmod(ic, iw64, p);
    }
    pl->roots.resize(sizeof(r));
    pl->iroots.resize(sizeof(ir));
    std::memcpy(pl->roots.data(), &r, sizeof(r));
    std::memcpy(pl->iroots.data(), &ir, sizeof(ir));

    // column levels
    for (int j = 0; j < pl->nlevels; ++j) {
        const uint32_t M = 1u << pl->level_logM[j], N1 = 1u << pl->level_log[j], S = M / N1;
        const uint32_t wM = h_powmod(wL, L / M, p), iwM = h_powmod(iwL, L / M, p);
        const bool innermost = (j == pl->nlevels - 1);
        if (pl->level_log[j] >= 5) {                     // inner twiddles of the LDS-staged column kernel
            const uint32_t wN = h_powmod(wL, L / N1, p), iwN = h_powmod(iwL, L / N1, p);
            typename MM::tw_t f[256] = {}, b[256] = {};
            for (uint32_t r = 0; r < std::max(4u, N1 / 16); ++r)
                for (uint32_t ka = 0; ka < 16; ++ka) {
                    f[r * 16 + ka] = MM::make_tw(h_powmod(wN, (uint64_t)r * ka, p), mp);
                    b[r * 16 + ka] = MM::make_tw(h_powmod(iwN, (uint64_t)r * ka, p), mp);
                }
            pl->tw_in_fwd[j].assign((unsigned char *)f, (unsigned char *)f + sizeof(f));
            pl->tw_in_inv[j].assign((unsigned char *)b, (unsigned char *)b + sizeof(b));
        }
        int rc = upload_table<MM>(pl, power_table<MM>(wM, N1, S, 1, mp), &pl->tw_fwd[j]);
        if (rc) return rc;
        rc = upload_table<MM>(pl, power_table<MM>(iwM, N1, S, innermost ? scale_mul : 1, mp), &pl->tw_inv[j]);
        if (rc) return rc;
        if (innermost) {
            if (scale_mul == invL) pl->tw_inv_plain = pl->tw_inv[j];
            else {
                rc = upload_table<MM>(pl, power_table<MM>(iwM, N1, S, invL, mp), &pl->tw_inv_plain);
                if (rc) return rc;
            }
        }
    }
    if (pl->fused) {
        // seq_product_kernel: row k1 of the forward column level as a dot product over the column's nonzero inputs
        // i = i1 * 4096 + i2 with (psi * w_L^k1)^i -- N1-point root, 4-step twiddle and negacyclic twist in one entry
        const uint32_t N1 = 1u << pl->level_log[0], cols = (pl->negacyclic ? N1 : N1 / 2) * (uint32_t)kRowLen;
        const uint32_t psi = pl->negacyclic ? h_powmod(pl->g, (p - 1) / (2 * L), p) : 1u;
        std::vector<typename MM::tw_t> t((size_t)N1 * cols);
        uint32_t base = psi;
        for (uint32_t k1 = 0; k1 < N1; ++k1) {
            uint32_t cur = 1;
            for (uint32_t i = 0; i < cols; ++i) {
                t[(size_t)k1 * cols + i] = MM::make_tw(cur, mp);
                cur = h_mulmod(cur, base, p);
            }
            base = h_mulmod(base, wL, p);
        }
        int rc = upload_table<MM>(pl, t, &pl->seq_tw);
        if (rc) return rc;
    }
    // row tables
    const uint32_t w4096 = h_powmod(wL, L / kRowLen, p), iw4096 = h_powmod(iwL, L / kRowLen, p);
    const uint32_t w256 = h_powmod(w4096, 16, p), iw256 = h_powmod(iw4096, 16, p);
    int rc = upload_table<MM>(pl, power_table<MM>(w4096, 16, 256, 1, mp), &pl->twA);
    if (rc) return rc;
    rc = upload_table<MM>(pl, power_table<MM>(w256, 16, 16, 1, mp), &pl->twB);
    if (rc) return rc;
    rc = upload_table<MM>(pl, power_table<MM>(iw4096, 16, 256, 1, mp), &pl->itwA);
    if (rc) return rc;
    rc = upload_table<MM>(pl, power_table<MM>(iw256, 16, 16, 1, mp), &pl->itwB);
    if (rc) return rc;
    if constexpr (MM::kMontgomery) {
        const uint32_t i4 = h_powmod(wL, L / 4, p);
        rc = upload_table<Mont32>(pl, rotated_table(w4096, 256, i4, mp), &pl->twA2);
        if (rc) return rc;
        rc = upload_table<Mont32>(pl, rotated_table(w256, 16, i4, mp), &pl->twB2);
        if (rc) return rc;
        const uint32_t ii4 = h_powmod(iwL, L / 4, p);
        rc = upload_table<Mont32>(pl, rotated_table(iw256, 16, ii4, mp), &pl->itwB2);
        if (rc) return rc;
        rc = upload_table<Mont32>(pl, rotated_table_inv(iw4096, ii4, mp), &pl->itwA2);
    }
    return rc;
}

// ---- launch sequences -----------------------------------------------------------------------------
template <class MM>
static void fill_roots(Roots64<MM> &dst, const std::vector<unsigned char> &src)
{
    std::memcpy(&dst, src.data(), sizeof(dst));
}

template <class MM>
static ColArgs<MM> fill_col(ntt_plan *pl, int level, bool inverse, const uint32_t *in0, const uint32_t *in1,
                            uint32_t *out0, uint32_t *out1, size_t in_stride, size_t out_stride,
                            uint32_t in_len, uint32_t out_len, const void *tw, bool use_scale0,
                            const std::vector<unsigned char> &scale0, bool canon, const void *twist = nullptr)
{
    ColArgs<MM> a{};
    a.twist = static_cast<const typename MM::tw_t *>(twist);
    a.in[0] = in0; a.in[1] = in1;
    a.out[0] = out0; a.out[1] = out1;
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.in_len = in_len; a.out_len = out_len;
    a.logM = pl->level_logM[level];
    a.logS = pl->level_logM[level] - pl->level_log[level];
    a.tw = static_cast<const typename MM::tw_t *>(tw);
    a.use_scale0 = use_scale0 ? 1 : 0;
    if (use_scale0) std::memcpy(&a.scale0, scale0.data(), sizeof(a.scale0));
    a.canon = canon ? 1 : 0;
    fill_roots<MM>(a.roots, inverse ? pl->iroots : pl->roots);
    if (pl->level_log[level] >= 5) {
        const auto &src = inverse ? pl->tw_in_inv[level] : pl->tw_in_fwd[level];
        std::memcpy(a.tw_in, src.data(), std::min(src.size(), sizeof(a.tw_in)));
    }
    return a;
}

template <class MM>
static void col_pass(ntt_plan *pl, int level, bool inverse, bool half_in, const uint32_t *in0,
                     const uint32_t *in1, uint32_t *out0, uint32_t *out1, size_t in_stride,
                     size_t out_stride, uint32_t in_len, uint32_t out_len, const void *tw,
                     bool use_scale0, const std::vector<unsigned char> &scale0, bool canon,
                     unsigned batch, unsigned nops, const void *twist = nullptr)
{
    ColArgs<MM> a = fill_col<MM>(pl, level, inverse, in0, in1, out0, out1, in_stride, out_stride, in_len,
                                 out_len, tw, use_scale0, scale0, canon, twist);
    ntt_ctx *cx = pl->ctx;
    // 128 and 256 points exist only in the LDS-staged kernel; 32 and 64 in both
    const bool col16 = pl->level_log[level] >= 7 || ((inverse ? pl->col16_inv : pl->col16_fwd) && pl->level_log[level] >= 5);
    cx->begin(inverse ? K_COL_INV : K_COL_FWD);
    if (col16)
        launch_col16(pl->kind, pl->lazy, pl->level_log[level], inverse, half_in, &a, pl->mp,
                     pl->L >> pl->level_log[level], batch, nops, cx->cur);
    else
        launch_col(pl->kind, pl->lazy_mode(), pl->level_log[level], inverse, half_in, &a, pl->mp,
                   pl->L >> pl->level_log[level], batch, nops, cx->cur);
    cx->end();
}

template <class MM>
static RowArgs<MM> fill_row(ntt_plan *pl, const uint32_t *a0, const uint32_t *b0, uint32_t *out)
{
    RowArgs<MM> a{};
    a.a = a0; a.b = b0; a.out = out;
    a.twA = static_cast<const typename MM::tw_t *>(pl->twA);
    a.twB = static_cast<const typename MM::tw_t *>(pl->twB);
    a.itwA = static_cast<const typename MM::tw_t *>(pl->itwA);
    a.itwB = static_cast<const typename MM::tw_t *>(pl->itwB);
    a.twA2 = static_cast<const uint2 *>(pl->twA2);
    a.twB2 = static_cast<const uint2 *>(pl->twB2);
    a.itwA2 = static_cast<const uint2 *>(pl->itwA2);
    a.itwB2 = static_cast<const uint2 *>(pl->itwB2);
    fill_roots<MM>(a.roots, pl->roots);
    fill_roots<MM>(a.iroots, pl->iroots);
    return a;
}

template <class MM>
static void row_pass(ntt_plan *pl, int which, const uint32_t *a0, const uint32_t *b0, uint32_t *out,
                     unsigned rows)
{
    RowArgs<MM> a = fill_row<MM>(pl, a0, b0, out);
    ntt_ctx *cx = pl->ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    static const bool ext_events = [] { const char *e = std::getenv("NTT_AMD_EXT_EVENTS"); return !(e && e[0] == '0'); }();
    if (which == 0 && ext_events) {
        const int st = cx->begin_ext(K_ROW_MUL, &e0, &e1);     // 1: this launch carries events; -1: skipped by the sampling stride
        if (st == 1) {
            launch_row_mul_timed(pl->kind, pl->lazy_mode(), &a, pl->mp, rows, cx->cur, e0, e1);
            return;
        }
        if (st == -1) {
            launch_row_mul(pl->kind, pl->lazy_mode(), &a, pl->mp, rows, cx->cur);
            return;
        }
    }
    cx->begin(which == 0 ? K_ROW_MUL : which == 1 ? K_ROW_FWD : K_ROW_INV);
    if (which == 0) launch_row_mul(pl->kind, pl->lazy_mode(), &a, pl->mp, rows, cx->cur);
    else if (which == 1) launch_row_fwd(pl->kind, pl->lazy_mode(), &a, pl->mp, rows, cx->cur);
    else launch_row_inv(pl->kind, pl->lazy_mode(), &a, pl->mp, rows, cx->cur);
    cx->end();
}

template <class MM>
static void small_pass(ntt_plan *pl, int mode, const uint32_t *a0, const uint32_t *b0, uint32_t *out,
                       size_t in_stride, size_t out_stride, uint32_t in_len, uint32_t out_len,
                       unsigned batch)
{
    SmallArgs<MM> a{};
    a.a = a0; a.b = b0; a.out = out;
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.in_len = in_len; a.out_len = out_len;
    a.logL = pl->logL;
    a.w = static_cast<const typename MM::tw_t *>(pl->small_w);
    a.iw = static_cast<const typename MM::tw_t *>(pl->small_iw);
    std::memcpy(&a.scale, (mode == 0 ? pl->scale_mul : pl->scale_plain).data(), sizeof(a.scale));
    a.mode = mode;
    if (mode == 0 && pl->negacyclic) {
        a.twist = static_cast<const typename MM::tw_t *>(pl->twist_fwd);
        a.itwist = static_cast<const typename MM::tw_t *>(pl->small_itwist);
    }
    ntt_ctx *cx = pl->ctx;
    cx->begin(K_SMALL);
    launch_small(pl->kind, pl->lazy, &a, pl->mp, batch, cx->cur);
    cx->end();
}

// Whatever ends a function that spreads its chunks over the stream lanes (a failed HIP call between fork and join
// included), the context gets its own stream back as the launch target and every side lane that may hold work is
// joined into it, so that the next call on the context never runs on a half-ordered stream.
struct LaneJoin {
    ntt_ctx *cx;
    int forked = 0;                                   // side lanes 1 .. forked-1 were made to wait on the fork event
    int join()
    {
        cx->cur = cx->stream;
        int rc = NTT_OK;
        for (int l = 1; l < forked; ++l) {
            hipError_t e = hipEventRecord(cx->lane_done[l], cx->lane[l]);
            if (e == hipSuccess) e = hipStreamWaitEvent(cx->stream, cx->lane_done[l], 0);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(cx->lane[l]);     // last resort: order by waiting on the host
                rc = fail(NTT_ERR_HIP, std::string("joining stream lanes: ") + hipGetErrorString(e));
            }
        }
        forked = 0;
        return rc;
    }
    ~LaneJoin() { if (forked || cx->cur != cx->stream) (void)join(); }
};

// a, b, ab: DEVICE pointers.  area == nullptr: the plan's work buffers and lanes.
template <class MM>
static int multiply_device(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                           size_t ab_stride, int batch, const WorkArea *area = nullptr)
{
    // negacyclic plans: L = n, no zero padding, n outputs, operands and result twisted by psi^(+-i)
    const bool neg = pl->negacyclic;
    const uint32_t n = (uint32_t)pl->n, L = pl->L, out_len = neg ? n : 2 * n - 1;
    if (pl->wide) return multiply_device<MM>(pl->wide, a, b, ab, ab_stride, batch, area);
    pl->ctx->cur = pl->ctx->stream;
    if (pl->logL == kRowLog && pl->twA) {
        Row4096Args<MM> ra{};
        ra.row = fill_row<MM>(pl, a, b, ab);
        ra.in_stride = n; ra.out_stride = ab_stride;
        ra.in_len = n; ra.out_len = pl->fold2048 ? n : out_len;
        ra.fold = pl->fold2048 ? 1 : 0;
        std::memcpy(&ra.scale, pl->scale_mul.data(), sizeof(ra.scale));
        if (neg) {
            ra.twist = static_cast<const typename MM::tw_t *>(pl->twist_fwd);
            ra.itwist = static_cast<const typename MM::tw_t *>(pl->small_itwist);
        }
        ntt_ctx *cx = pl->ctx;
        cx->begin(K_ROW_MUL);
        launch_row_product(pl->kind, pl->lazy_mode(), &ra, pl->mp, (unsigned)batch, cx->cur);
        cx->end();
        return NTT_OK;
    }
    if (pl->logL <= kRowLog) {
        small_pass<MM>(pl, 0, a, b, ab, n, ab_stride, n, out_len, (unsigned)batch);
        return NTT_OK;
    }
    if (pl->fused) {                                       // L = 8192 / 16384: one kernel, rows one after the other
        SeqArgs<MM> sa{};
        sa.r.row = fill_row<MM>(pl, a, b, ab);
        sa.r.in_stride = n; sa.r.out_stride = ab_stride;
        sa.r.in_len = n; sa.r.out_len = out_len;
        if (neg) sa.r.itwist = static_cast<const typename MM::tw_t *>(pl->twist_inv);
        sa.tw_in = static_cast<const typename MM::tw_t *>(pl->seq_tw);
        sa.tw_inv = static_cast<const typename MM::tw_t *>(pl->tw_inv[0]);
        sa.plain_row0 = neg ? 0 : 1;
        ntt_ctx *cx = pl->ctx;
        cx->begin(K_ROW_MUL);
        launch_seq_product(pl->kind, pl->lazy, pl->level_log[0], !neg, &sa, pl->mp, (unsigned)batch, cx->cur);
        cx->end();
        return NTT_OK;
    }
    const int m = pl->nlevels;
    ntt_ctx *cx = pl->ctx;
    const int chunk = area ? area->chunk : pl->chunk;
    uint32_t *const base0 = area ? area->w0 : static_cast<uint32_t *>(pl->work0.ptr);
    uint32_t *const base1 = area ? area->w1 : static_cast<uint32_t *>(pl->work1.ptr);
    const int nchunks = (batch + chunk - 1) / chunk;
    int lanes = area ? 1 : std::min(std::min(pl->lanes, cx->nlanes), nchunks);
    if (cx->lane_limit > 0) lanes = std::min(lanes, cx->lane_limit);
    LaneJoin lanes_guard{cx};
    if (lanes > 1) {                                      // fork: side lanes start after prior work
        // ordering between the lanes is what keeps the results right: a failure here is an error, not a warning
        HIP_TRY(hipEventRecord(cx->fork_ev, cx->stream));
        for (int l = 1; l < lanes; ++l) {
            lanes_guard.forked = l + 1;
            HIP_TRY(hipStreamWaitEvent(cx->lane[l], cx->fork_ev, 0));
        }
    }
    int ci = 0;
    for (int c0 = 0; c0 < batch; c0 += chunk, ++ci) {
        const int l = ci % lanes;
        cx->cur = l == 0 ? cx->stream : cx->lane[l];
        uint32_t *w0 = base0 + (size_t)l * chunk * L;
        uint32_t *w1 = base1 + (size_t)l * chunk * L;
        const unsigned cb = (unsigned)std::min(chunk, batch - c0);
        const uint32_t *ca = a + (size_t)c0 * n, *cbp = b + (size_t)c0 * n;
        // forward column levels (both operands per launch); level 0 reads the caller's arrays with
        // implicit zero padding (n <= L/2: the upper half of every column is zero)
        if (area && area->u64_in) {                          // CRT path: u64 operands, reduced by the pass itself
            if constexpr (MM::kMontgomery) {
                const uint64_t *a64 = reinterpret_cast<const uint64_t *>(a) + (size_t)c0 * n;
                const uint64_t *b64 = reinterpret_cast<const uint64_t *>(b) + (size_t)c0 * n;
                ColArgs<MM> ca64 = fill_col<MM>(pl, 0, false, reinterpret_cast<const uint32_t *>(a64), reinterpret_cast<const uint32_t *>(b64),
                                                w0, w1, n, L, n, L, pl->tw_fwd[0], false, {}, false);
                ca64.in_one = area->u64_in->one;
                ca64.in_two32 = area->u64_in->two32;
                cx->begin(K_COL_FWD);
                const bool ok = launch_col_u64in(pl->lazy, pl->level_log[0], &ca64, pl->mp, pl->L >> pl->level_log[0], cb, 2, cx->cur);
                cx->end();
                if (!ok) return fail(NTT_ERR_INTERNAL, "no u64-input column kernel for this plan");
            } else {
                return fail(NTT_ERR_INTERNAL, "u64 operands need the Montgomery kernels");
            }
        } else
        col_pass<MM>(pl, 0, false, !neg, ca, cbp, w0, w1, n, L, n, L, pl->tw_fwd[0], false, {}, false, cb, 2,
                     neg ? pl->twist_fwd : nullptr);
        for (int j = 1; j < m; ++j)
            col_pass<MM>(pl, j, false, false, w0, w1, w0, w1, L, L, L, L, pl->tw_fwd[j], false, {}, false, cb, 2);
        // rows: forward a, forward b, pointwise, inverse
        row_pass<MM>(pl, 0, w0, w1, w0, cb * (L >> kRowLog));
        // inverse column levels, innermost first; the outermost writes the caller's ab
        for (int j = m - 1; j >= 1; --j)
            col_pass<MM>(pl, j, true, false, w0, nullptr, w0, nullptr, L, L, L, L, pl->tw_inv[j],
                         j == m - 1, pl->scale_mul, false, cb, 1);
        if (area && area->stop_before_outer_inverse) continue;
        col_pass<MM>(pl, 0, true, false, w0, nullptr, ab + (size_t)c0 * ab_stride, nullptr, L, ab_stride,
                     L, out_len, pl->tw_inv[0], m == 1, pl->scale_mul, true, cb, 1, neg ? pl->twist_inv : nullptr);
    }
    return lanes_guard.join();
}

int multiply_device_any(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                        size_t ab_stride, int batch, const WorkArea *area)
{
    const int rc = pl->kind == NTT_MODMUL_MONTGOMERY ? multiply_device<Mont32>(pl, a, b, ab, ab_stride, batch, area)
                                                     : multiply_device<Barrett32>(pl, a, b, ab, ab_stride, batch, area);
    const int lrc = take_launch_error();                  // a dispatcher without a kernel for this shape launched nothing
    return rc ? rc : lrc;
}

// Tile description for the LDS-staged digit reversal: dst[R(i)] = src[i] where i has the bit fields
// `w` (most significant first) and R reverses their order.  False if no tile of the wanted shape exists
// (never for L >= 8192; the caller then falls back to the plain kernel).
static bool make_perm_tile(const std::vector<int> &w, int logL, PermTileArgs &pa)
{
    const int nf = (int)w.size();
    if (nf > 8) return false;
    // field f occupies bits [lo[f], lo[f] + w[f]) of i and [klo[f], ...) of R(i)
    std::vector<int> lo(nf), klo(nf);
    int acc = logL;
    for (int f = 0; f < nf; ++f) { acc -= w[f]; lo[f] = acc; }
    acc = 0;
    for (int f = 0; f < nf; ++f) { klo[f] = acc; acc += w[f]; }
    if (acc != logL) return false;
    pa.logL = logL;
    pa.n_rev = nf;
    for (int f = 0; f < nf; ++f) pa.rev[f] = BitRun{lo[f], w[f], klo[f]};
    // 32 x 64 tiles (128-byte output runs, 256-byte input runs) measured best at L = 2^26: 217 us per
    // array; 64 x 128 tiles 282 us (profiles/sweep_permute_r01e.txt)
    int a = 5, b = 6;
    // bits of i that R sends to output bits [0, a): they must not meet the low b bits of i
    for (;; ) {
        uint32_t tmask = 0;
        pa.n_top = 0;
        for (int f = 0; f < nf && klo[f] < a; ++f) {
            const int width = std::min(w[f], a - klo[f]);
            pa.top[pa.n_top++] = BitRun{klo[f], width, lo[f]};
            tmask |= ((1u << width) - 1u) << lo[f];
        }
        const uint32_t lmask = (1u << b) - 1u;
        if ((tmask & lmask) == 0) {
            // remaining bits of i, lowest first, as runs fed from the block index
            uint32_t rest = ~(tmask | lmask) & (logL == 32 ? 0xffffffffu : ((1u << logL) - 1u));
            pa.n_mid = 0;
            int from = 0;
            for (int bit = 0; bit < logL; ) {
                if (!((rest >> bit) & 1u)) { ++bit; continue; }
                int width = 0;
                while (bit + width < logL && ((rest >> (bit + width)) & 1u)) ++width;
                if (pa.n_mid >= 12) return false;
                pa.mid[pa.n_mid++] = BitRun{from, width, bit};
                from += width;
                bit += width;
            }
            pa.a = a;
            pa.b = b;
            return from == logL - a - b;
        }
        if (b > 5) --b;
        else if (a > 5) --a;
        else return false;                                   // tile sizes below 2^10 have no kernel
    }
}

static void permute_pass(ntt_plan *pl, const uint32_t *src, uint32_t *dst, bool to_natural, unsigned batch)
{
    {
        std::vector<int> w;
        for (int j = 0; j < pl->nlevels; ++j) w.push_back(pl->level_log[j]);
        w.push_back(4); w.push_back(4); w.push_back(4);
        if (!to_natural) std::reverse(w.begin(), w.end());       // dst[pos] = src[k]: the fields of k, top first
        PermTileArgs ta{};
        if (make_perm_tile(w, pl->logL, ta)) {
            ta.src = src; ta.dst = dst;
            ta.canon_p = (to_natural && pl->lazy) ? pl->p : 0;   // forward spectrum leaves the rows in [0, 2p)
            ntt_ctx *cx = pl->ctx;
            cx->begin(K_PERMUTE);
            launch_permute_tiled(ta, batch, cx->cur);
            cx->end();
            return;
        }
    }
    PermArgs pa{};
    pa.src = src; pa.dst = dst;
    pa.logL = pl->logL;
    int d = 0;
    for (int j = 0; j < pl->nlevels; ++j) pa.level_log[d++] = pl->level_log[j];
    pa.level_log[d++] = 4; pa.level_log[d++] = 4; pa.level_log[d++] = 4;
    pa.nlevels = d;
    pa.to_natural = to_natural ? 1 : 0;
    pa.canon_p = (to_natural && pl->lazy) ? pl->p : 0;      // forward spectrum leaves the rows in [0, 2p)
    ntt_ctx *cx = pl->ctx;
    cx->begin(K_PERMUTE);
    launch_permute(pa, batch, cx->cur);
    cx->end();
}

// src, data: DEVICE pointers, [batch][L]; natural order in and out.  src == data transforms in place; otherwise src
// is only read (the first pass of either direction takes it as its input) and the result lands in data.
// scrambled: the spectrum stays in the engine's own (mixed-radix digit-reversed) order and range -- forward without the
// final permutation, inverse without the initial one -- which is all a pointwise product between the two needs
// (L > 4096 only; shorter transforms have no permutation pass and keep natural order either way).
template <class MM>
static int transform_device(ntt_plan *pl, const uint32_t *src, uint32_t *data, int batch, bool inverse, bool scrambled = false)
{
    const uint32_t L = pl->L;
    pl->ctx->cur = pl->ctx->stream;
    if (pl->logL <= kRowLog) {
        small_pass<MM>(pl, inverse ? 2 : 1, src, nullptr, data, L, L, L, L, (unsigned)batch);
        return NTT_OK;
    }
    const int m = pl->nlevels;
    // a batch of transforms alternates its chunks between the stream lanes like a batch of products (round 3): the
    // memory-bound column and reordering passes of one chunk run beside the row pass of the other
    ntt_ctx *cx = pl->ctx;
    const int nchunks = (batch + pl->chunk - 1) / pl->chunk;
    int lanes = std::min(std::min(pl->lanes, cx->nlanes), nchunks);
    if (cx->lane_limit > 0) lanes = std::min(lanes, cx->lane_limit);
    if (const char *e = std::getenv("NTT_AMD_TRANSFORM_LANES")) lanes = std::max(1, std::min(lanes, std::atoi(e)));
    LaneJoin lanes_guard{cx};
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(cx->fork_ev, cx->stream));
        for (int l = 1; l < lanes; ++l) {
            lanes_guard.forked = l + 1;
            HIP_TRY(hipStreamWaitEvent(cx->lane[l], cx->fork_ev, 0));
        }
    }
    int ci = 0;
    for (int c0 = 0; c0 < batch; c0 += pl->chunk, ++ci) {
        const int l = ci % lanes;
        cx->cur = l == 0 ? cx->stream : cx->lane[l];
        uint32_t *w0 = static_cast<uint32_t *>(pl->work0.ptr) + (size_t)l * pl->chunk * L;
        const unsigned cb = (unsigned)std::min(pl->chunk, batch - c0);
        const uint32_t *s = src + (size_t)c0 * L;
        uint32_t *d = data + (size_t)c0 * L;
        if (!inverse) {
            for (int j = 0; j < m; ++j)
                col_pass<MM>(pl, j, false, false, j == 0 ? s : d, nullptr, d, nullptr, L, L, L, L, pl->tw_fwd[j], false,
                             {}, false, cb, 1);
            if (scrambled) {
                row_pass<MM>(pl, 1, d, nullptr, d, cb * (L >> kRowLog));     // in place: a row is read before it is written
            } else {
                row_pass<MM>(pl, 1, d, nullptr, w0, cb * (L >> kRowLog));
                permute_pass(pl, w0, d, true, cb);
            }
        } else {
            if (scrambled) row_pass<MM>(pl, 2, s, nullptr, w0, cb * (L >> kRowLog));
            else {
                permute_pass(pl, s, w0, false, cb);
                row_pass<MM>(pl, 2, w0, nullptr, w0, cb * (L >> kRowLog));
            }
            for (int j = m - 1; j >= 0; --j)
                col_pass<MM>(pl, j, true, false, w0, nullptr, j == 0 ? d : w0, nullptr, L, L, L, L,
                             j == m - 1 ? pl->tw_inv_plain : pl->tw_inv[j], j == m - 1, pl->scale_plain,
                             j == 0, cb, 1);
        }
    }
    return lanes_guard.join();
}

// One transform of the plan's length from src to dst (DEVICE pointers on the plan's GPU, src == dst allowed), queued
// on the context's stream: the local S-point transforms of the in-library distributed transform (multidev.cpp).
int transform_device_any(ntt_plan *pl, const uint32_t *src, uint32_t *dst, bool inverse, bool scrambled)
{
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    const int rc = pl->kind == NTT_MODMUL_MONTGOMERY ? transform_device<Mont32>(pl, src, dst, 1, inverse, scrambled)
                                                     : transform_device<Barrett32>(pl, src, dst, 1, inverse, scrambled);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return take_launch_error();
}

}  // namespace ntt

// ================================================================================================
// ntt_ctx
// ================================================================================================
hipEvent_t ntt_ctx::grab_event()
{
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;      // begin() then leaves this launch untimed
    return e;
}
void ntt_ctx::begin(int id)
{
    stamp_open = timing && (timing_only < 0 || timing_only == id);
    if (stamp_open && timing_stride > 1 && (timing_tick++ % timing_stride) != 0) stamp_open = false;
    if (!stamp_open) return;
    Stamp s{id, grab_event(), grab_event()};
    if (!s.e0 || !s.e1 || hipEventRecord(s.e0, cur ? cur : stream) != hipSuccess) {   // timing is diagnostic only
        if (s.e0) event_pool.push_back(s.e0);
        if (s.e1) event_pool.push_back(s.e1);
        stamp_open = false;
        return;
    }
    stamps.push_back(s);
}
int ntt_ctx::begin_ext(int id, hipEvent_t *e0, hipEvent_t *e1)
{
    stamp_open = false;
    if (!(timing && (timing_only < 0 || timing_only == id))) return 0;
    if (timing_stride > 1 && (timing_tick++ % timing_stride) != 0) return -1;
    Stamp s{id, grab_event(), grab_event()};
    if (!s.e0 || !s.e1) {
        if (s.e0) event_pool.push_back(s.e0);
        if (s.e1) event_pool.push_back(s.e1);
        return -1;
    }
    stamps.push_back(s);
    *e0 = s.e0;
    *e1 = s.e1;
    return 1;
}
void ntt_ctx::end()
{
    if (!stamp_open) return;
    (void)hipEventRecord(stamps.back().e1, cur ? cur : stream);
    stamp_open = false;
}
int ntt_ctx::fold()
{
    if (stamps.empty()) return NTT_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    for (auto &s : stamps) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.e0, s.e1));
        acc_ms[s.id] += ms;
        acc_n[s.id] += 1;
        event_pool.push_back(s.e0);
        event_pool.push_back(s.e1);
    }
    stamps.clear();
    return NTT_OK;
}

using namespace ntt;

extern "C" {

const char *ntt_last_error(void) { return ntt::last_error_cstr(); }

int ntt_ctx_create(const int *device_ids, int ndev, ntt_ctx **out_ctx)
try {
    if (!out_ctx) return fail(NTT_ERR_INVALID, "ntt_ctx_create: out_ctx is null");
    *out_ctx = nullptr;
    if (ndev < 1) return fail(NTT_ERR_INVALID, "ntt_ctx_create: ndev must be at least 1");
    if (ndev > 1) return group_ctx_create(device_ids, ndev, out_ctx);        // one slot per entry of device_ids (multidev.cpp)
    int dev = device_ids ? device_ids[0] : 0, count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (dev < 0 || dev >= count) return fail(NTT_ERR_INVALID, "ntt_ctx_create: no such device");
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop{};
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(NTT_ERR_UNSUPPORTED, std::string("ntt_ctx_create: kernels are built for gfx950, device is ") + prop.gcnArchName);
    auto *cx = new ntt_ctx();
    cx->device = dev;
    {   // this device's copy of the scratch array (device `dev` is current): its address, never the array's name.
        // A failed lookup leaves cx->scratch null: only ntt_poly_multiply_batched_noalloc then reports unsupported.
        void *sp = nullptr;
        const hipError_t se = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
        if (se != hipSuccess) (void)hipGetLastError();
        std::lock_guard<std::mutex> sl(g_scratch_mu);
        if (se == hipSuccess && sp && dev < kMaxScratchDevices &&
            (g_scratch_last[dev] || hipEventCreateWithFlags(&g_scratch_last[dev], hipEventDisableTiming) == hipSuccess)) {
            cx->scratch = sp;
            cx->scratch_bytes = kScratchBytes;
        } else {
            (void)hipGetLastError();
        }
    }
    hipError_t e = hipStreamCreateWithFlags(&cx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete cx;
        return fail(NTT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    cx->stream = cx->own_stream;
    cx->cur = cx->stream;
    int nl = 2;
    if (const char *ev = std::getenv("NTT_AMD_STREAMS")) nl = std::atoi(ev);
    cx->nlanes = std::max(1, std::min(nl, (int)ntt_ctx::kMaxLanes));
    for (int l = 1; l < cx->nlanes; ++l) {
        if (hipStreamCreateWithFlags(&cx->lane[l], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&cx->lane_done[l], hipEventDisableTiming) != hipSuccess) {
            cx->nlanes = l;
            break;
        }
    }
    if (hipEventCreateWithFlags(&cx->fork_ev, hipEventDisableTiming) != hipSuccess) {
        cx->fork_ev = nullptr;
        cx->nlanes = 1;                                    // no fork event: single lane
    }
    *out_ctx = cx;
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_ctx_set_stream(ntt_ctx *ctx, void *hip_stream)
try {
    if (!ctx) return fail(NTT_ERR_INVALID, "null context");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "ntt_ctx_set_stream: a multi-device context has one stream per slot: set it on ntt_ctx_device(ctx, slot)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    // (void*)1 = hipStreamLegacy names the null stream; the runtime's own handling of that handle in
    // hipEventRecord / hipStreamWaitEvent faults (ROCm 7.0 runtime bundled with PyTorch), its handle 0 does not
    if (hip_stream == reinterpret_cast<void *>(1)) ctx->stream = nullptr;
    else ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    ctx->cur = ctx->stream;
    return NTT_OK;
}
NTT_CATCH_STATUS
void *ntt_ctx_stream(ntt_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

int ntt_ctx_synchronize(ntt_ctx *ctx)
try {
    if (!ctx) return fail(NTT_ERR_INVALID, "null context");
    if (ctx->is_group()) return group_ctx_synchronize(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    return sync_stream(ctx);
}
NTT_CATCH_STATUS

int ntt_ctx_enable_kernel_timing(ntt_ctx *ctx, int enable)
try {
    if (!ctx) return fail(NTT_ERR_INVALID, "null context");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "kernel timing is per slot: use ntt_ctx_device(ctx, slot)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx->fold();
    ctx->timing = enable != 0;
    for (int i = 0; i < K_COUNT; ++i) { ctx->acc_ms[i] = 0; ctx->acc_n[i] = 0; }
    return rc;
}
NTT_CATCH_STATUS

int ntt_ctx_kernel_timing_select(ntt_ctx *ctx, int index)
try {
    if (!ctx || index < -1 || index >= K_COUNT) return fail(NTT_ERR_INVALID, "ntt_ctx_kernel_timing_select: bad index");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "kernel timing is per slot: use ntt_ctx_device(ctx, slot)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx->fold();
    ctx->timing_only = index;
    return rc;
}
NTT_CATCH_STATUS

int ntt_ctx_kernel_timing_stride(ntt_ctx *ctx, int every)
try {
    if (!ctx || every < 1) return fail(NTT_ERR_INVALID, "ntt_ctx_kernel_timing_stride: every must be >= 1");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "kernel timing is per slot: use ntt_ctx_device(ctx, slot)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx->fold();
    ctx->timing_stride = every;
    ctx->timing_tick = 0;
    return rc;
}
NTT_CATCH_STATUS

int ntt_ctx_kernel_time(ntt_ctx *ctx, int index, const char **name, double *total_ms, long *launches)
try {
    if (!ctx || index < 0 || index >= K_COUNT) return fail(NTT_ERR_INVALID, "ntt_ctx_kernel_time: bad index");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "kernel timing is per slot: use ntt_ctx_device(ctx, slot)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx->fold();
    if (rc) return rc;
    if (name) *name = kKernelNames[index];
    if (total_ms) *total_ms = ctx->acc_ms[index];
    if (launches) *launches = ctx->acc_n[index];
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_plan_destroy(ntt_plan *pl)
try {
    if (!pl) return NTT_OK;
    if (!pl->shard.empty()) return group_plan_destroy(pl);
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    for (void *d : pl->owned) (void)hipFree(d);
    pl->work0.release();
    pl->work1.release();
    if (pl->wide) ntt_plan_destroy(pl->wide);
    delete pl;
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_ctx_destroy(ntt_ctx *ctx)
try {
    if (!ctx) return NTT_OK;
    if (ctx->is_group()) return group_ctx_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->plan_cache) ntt_plan_destroy(kv.second);
    ctx->plan_cache.clear();
    ntt::crt_release(ctx);
    ntt::crt_direct_release(ctx);
    ntt::stream_release(ctx);
    ctx->stage_a.release();
    ctx->pin_a.release();
    ctx->pin_b.release();
    ctx->pin_out.release();
    ctx->stage_b.release();
    ctx->stage_out.release();
    for (auto &s : ctx->stamps) { (void)hipEventDestroy(s.e0); (void)hipEventDestroy(s.e1); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int l = 1; l < ctx->nlanes; ++l) {
        if (ctx->lane[l]) (void)hipStreamDestroy(ctx->lane[l]);
        if (ctx->lane_done[l]) (void)hipEventDestroy(ctx->lane_done[l]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return NTT_OK;
}
NTT_CATCH_STATUS

static int plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, bool negacyclic, ntt_plan **out_plan,
                       int force_logL = 0);

int ntt_plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, ntt_plan **out_plan)
try {
    return plan_create(ctx, p, n, batch, modmul_kind, false, out_plan);
}
NTT_CATCH_STATUS

int ntt_negacyclic_plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, ntt_plan **out_plan)
try {
    if (n < 2 || (n & (n - 1)) != 0) return fail(NTT_ERR_INVALID, "ntt_negacyclic_plan_create: n must be a power of two >= 2");
    return plan_create(ctx, p, n, batch, modmul_kind, true, out_plan);
}
NTT_CATCH_STATUS

static int plan_create(ntt_ctx *ctx, uint32_t p, int n, int batch, int modmul_kind, bool negacyclic, ntt_plan **out_plan,
                       int force_logL)
{
    if (!ctx || !out_plan) return fail(NTT_ERR_INVALID, "ntt_plan_create: null argument");
    *out_plan = nullptr;
    if (n <= 0 || batch <= 0) return fail(NTT_ERR_INVALID, "ntt_plan_create: n and batch must be positive");
    if (modmul_kind != NTT_MODMUL_MONTGOMERY && modmul_kind != NTT_MODMUL_BARRETT)
        return fail(NTT_ERR_INVALID, "ntt_plan_create: unknown modmul_kind");
    if (p >= (1u << 31) || p < 3 || !h_is_prime(p))
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan_create: p must be an odd prime below 2^31");
    if (n > (1 << 25)) return fail(NTT_ERR_UNSUPPORTED, "ntt_plan_create: n too large (transform length is capped at 2^26)");
    if (ctx->is_group() && force_logL == 0) return group_plan_create(ctx, p, n, batch, modmul_kind, negacyclic, out_plan);
    int logL = 1;
    while ((1u << logL) < 2u * (uint32_t)n) ++logL;
    if (negacyclic) --logL;                               // x^n + 1: transform length n, 2n-th root psi
    if (force_logL > logL) logL = force_logL;             // companion plan: longer zero-padded transform
    const uint32_t L = 1u << logL;
    if ((p - 1) % (negacyclic ? 2 * (uint64_t)L : (uint64_t)L) != 0)
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan_create: p - 1 is not divisible by the transform length (p is not NTT-friendly for this n)");
    HIP_TRY(hipSetDevice(ctx->device));

    std::unique_ptr<ntt_plan> pl(new ntt_plan());
    pl->ctx = ctx;
    pl->p = p;
    pl->g = h_primitive_root(p);
    pl->n = n;
    pl->logL = logL;
    pl->L = L;
    pl->batch = batch;
    pl->kind = modmul_kind;
    pl->negacyclic = negacyclic;
    pl->mp = make_mod_params(p);
    pl->lazy = p < (1u << 30);
    if (const char *e = std::getenv("NTT_AMD_FORCE_STRICT"))
        if (e[0] == '1') pl->lazy = false;
    // 8p <= 2^32: the wide-range transforms exist, and since the radix-4 steps of the [0, 2p) Montgomery kernels
    // (device_ntt.hpp: acc4_step, reg_ntt16_tw) they are the slower choice there (p = 469762049: 533 k against 591 k
    // products/s), so only the Barrett kernels, which have no such steps, take them by default.  NTT_AMD_WIDE=1 / 0
    // forces / forbids them (A/B runs and the tests of both).
    pl->wide8 = pl->lazy && p < (1u << 29) && modmul_kind != NTT_MODMUL_MONTGOMERY;
    if (const char *e = std::getenv("NTT_AMD_WIDE")) {
        if (e[0] == '0') pl->wide8 = false;
        if (e[0] == '1') pl->wide8 = pl->lazy && p < (1u << 29);
    }
    // forward column pass: the register kernel wins in a full launch, the LDS-staged one (four times the
    // workgroups, a quarter of the work per thread) when a launch holds a few products only
    // (single product: 9.7 -> 8.2 us, tools/time_single.py).  The Barrett kernels always take the LDS-staged one: their
    // 8-byte twiddles put the 64-point register kernel's root table beyond the SGPR file (150+ spilled), and its forward
    // pass takes 1.46 us per product where the staged kernel takes 0.80 (391 k -> 435 k products/s, bench.py --barrett)
    pl->col16_fwd = batch <= 4 || modmul_kind == NTT_MODMUL_BARRETT;
    if (const char *e = std::getenv("NTT_AMD_COL16")) {
        const int v = std::atoi(e);
        pl->col16_fwd = v & 1;
        pl->col16_inv = v & 2;
    }
    if (logL > kRowLog) {
        const int c = logL - kRowLog;
        // 256-point levels pay for zero-padded (linear) products only: 10.4 vs 12.8 us at L = 2^20; with a
        // full-length first level (negacyclic) two levels of 16 points are faster (14.5 vs 16.3 us)
        int maxlog = negacyclic ? kMaxColLog - 1 : kMaxColLog;
        if (const char *e = std::getenv("NTT_AMD_COL_MAXLOG")) maxlog = std::max(4, std::min(kMaxColLog, std::atoi(e)));
        const int m = (c + maxlog - 1) / maxlog;
        int logM = logL;
        for (int j = 0; j < m; ++j) {
            pl->level_log[j] = c / m + (j < c % m ? 1 : 0);
            pl->level_logM[j] = logM;
            logM -= pl->level_log[j];
        }
        pl->nlevels = m;
        // L = 8192 and the linear L = 16384: one kernel does the column level around the rows, which it takes one after
        // the other (kernels_fused.hip; 0.077 -> 0.044, 0.137 -> 0.113, negacyclic 8192 0.083 -> 0.054 us per product,
        // profiles/time_medium_r02.txt).  The full-length L = 16384 only ties and stays on the three kernels, which
        // NTT_AMD_FUSED=0 keeps everywhere
        // (=2 sends the full-length L = 16384 through the one kernel as well)
        int fused_mode = 1;
        if (const char *e = std::getenv("NTT_AMD_FUSED")) fused_mode = std::atoi(e);
        pl->fused = fused_mode && m == 1 && (pl->level_log[0] == 1 || (pl->level_log[0] == 2 && (!negacyclic || fused_mode == 2)));
        // work buffers: two operands per product, `chunk` products per launch and per lane.
        // Measured (profiles/sweep_chunk_r01.txt, profiles/sweep_chunk_r01d.txt): throughput rises
        // with the chunk up to ~40 products (fewer launches, smaller tail rounds), is flat from 40
        // to 56 and drops 5 % at 64 and beyond, where the intermediates of the two lanes no longer
        // fit the 256 MiB Infinity Cache together: the default was 96 MiB per lane (48 products at
        // L = 2^18).  With the shorter kernels of round 3 the launches count for more: 56 products +0.65 %, 64
        // +0.85 %, 72 and more -9 % (profiles/sweep_chunk_r03h.txt, sweep_chunk_r03k.txt): 112 MiB per lane (56 products),
        // one step away from the edge.
        size_t per = (size_t)L * sizeof(uint32_t);
        long chunk = (long)((112u << 20) / (2 * per));
        if (const char *e = std::getenv("NTT_AMD_CHUNK")) chunk = std::atol(e);
        const bool chunk_forced = std::getenv("NTT_AMD_CHUNK") != nullptr;
        chunk = std::max<long>(1, std::min<long>(chunk, batch));
        chunk = std::min<long>(chunk, 65535);
        // a few chunks only: equal shares, so that the last launch is not a small one (64 products: 32 + 32
        // instead of 48 + 16, 14.3 -> 13.6 us per CRT product); with many chunks the tail does not matter and
        // full-size chunks measured 1.5 % better
        if (!chunk_forced) {
            long nch = (batch + chunk - 1) / chunk;
            if (nch > 1 && nch <= 4 && (nch & 1) && ctx->nlanes == 2) ++nch;     // the same number of chunks on both lanes
            if (nch <= 4) chunk = (batch + nch - 1) / nch;
        }
        pl->chunk = (int)chunk;
        pl->lanes = (int)std::min<long>(ctx->nlanes, (batch + chunk - 1) / chunk);
        const long slots = pl->lanes;
        int rc = pl->work0.ensure(per * chunk * slots);
        if (rc == NTT_OK) rc = pl->work1.ensure(per * chunk * slots);
        if (rc) { ntt_plan *raw = pl.release(); ntt_plan_destroy(raw); return rc; }
    } else {
        pl->chunk = batch;
    }
    int rc = modmul_kind == NTT_MODMUL_MONTGOMERY ? build_tables<Mont32>(pl.get())
                                                  : build_tables<Barrett32>(pl.get());
    if (rc) { ntt_plan *raw = pl.release(); ntt_plan_destroy(raw); return rc; }
    // x^2048 + 1: the same trick through the linear product of L = 4096 (no twist), whose coefficients i and
    // i + 2048 meet in one thread of the row kernel and are subtracted at the store: 0.045 -> 0.022 us
    if (negacyclic && n == 2048 && force_logL == 0) {
        bool want = true;
        if (const char *e = std::getenv("NTT_AMD_WIDE2048")) want = e[0] != '0';
        if (want) {
            rc = plan_create(ctx, p, n, batch, modmul_kind, false, &pl->wide, kRowLog);
            if (rc) { ntt_plan *raw = pl.release(); ntt_plan_destroy(raw); return rc; }
            pl->wide->fold2048 = true;
        }
    }
    if (!negacyclic && logL == kRowLog - 1 && force_logL == 0 && (p - 1) % (1u << kRowLog) == 0) {
        bool want = true;
        if (const char *e = std::getenv("NTT_AMD_WIDE2048")) want = e[0] != '0';
        if (want) {
            rc = plan_create(ctx, p, n, batch, modmul_kind, false, &pl->wide, kRowLog);
            if (rc) { ntt_plan *raw = pl.release(); ntt_plan_destroy(raw); return rc; }
        }
    }
    *out_plan = pl.release();
    return NTT_OK;
}

int ntt_plan_length(const ntt_plan *plan) { return plan ? (int)plan->L : 0; }

static int multiply_entry(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab, size_t ab_stride,
                          int batch, bool negacyclic);

int ntt_poly_multiply_batched(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                              size_t ab_stride, int batch)
try {
    return multiply_entry(pl, a, b, ab, ab_stride, batch, false);
}
NTT_CATCH_STATUS

int ntt_negacyclic_multiply_batched(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *c,
                                    size_t c_stride, int batch)
try {
    return multiply_entry(pl, a, b, c, c_stride, batch, true);
}
NTT_CATCH_STATUS

static int multiply_entry(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab, size_t ab_stride,
                          int batch, bool negacyclic)
{
    if (!pl || !a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: null argument");
    if (pl->negacyclic != negacyclic)
        return fail(NTT_ERR_INVALID, negacyclic ? "ntt_negacyclic_multiply_batched: the plan is not a negacyclic plan"
                                                : "ntt_poly_multiply_batched: negacyclic plans multiply through ntt_negacyclic_multiply_batched");
    if (!pl->shard.empty()) {                              // plan of a multi-device context: rows sharded over the slots
        const size_t row_len = negacyclic ? (size_t)pl->n : 2 * (size_t)pl->n - 1;
        if (ab_stride < row_len) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: output row stride below the row length");
        return group_multiply(pl, a, b, ab, ab_stride, batch, negacyclic);
    }
    if (batch <= 0) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: batch must be positive");
    if (batch > pl->batch) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: batch exceeds the plan's capacity");
    const size_t n = (size_t)pl->n, out_len = negacyclic ? n : 2 * n - 1;
    if (ab_stride < out_len) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched: output row stride below the row length");
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    const size_t in_bytes = (size_t)batch * n * 4, out_elems = ((size_t)batch - 1) * ab_stride + out_len;
    // Page-locked host arrays of a SMALL call (ntt_host_alloc / hipHostMalloc: mapped into the device's address space)
    // are used in place: the first pass reads the operands and the last pass writes the result over PCIe, which costs
    // less than three staging copies with a launch latency each when there is one product to move (93 -> 6x us per
    // n = 2^17 product, profiles/latency_r03.txt).  Larger calls stage as before (DMA beats loads over the link).
    // PAGEABLE arrays of such a call (the harness's own: main-montgomery.cc:297 keeps a, b, ab in static arrays) go the
    // same way through page-locked buffers of the context and a plain CPU memcpy each way: 13 us for both operands and
    // 11 us for the result of an n = 2^17 product, where hipMemcpy from and to pageable memory takes 43 and 63 us
    // (profiles/ubench_stage_r03.txt).  NTT_AMD_PAGEABLE_STAGE=0 (read per call) keeps the hipMemcpy staging.
    const bool zero_copy = in_bytes <= kZeroCopyBytes;
    const bool own_pinned = zero_copy && [] { const char *e = std::getenv("NTT_AMD_PAGEABLE_STAGE"); return !(e && e[0] == '0'); }();
    // one runtime query per array (each costs about a microsecond, more for memory the runtime does not know)
    enum class Mem { Device, Pinned, Pageable };
    auto classify = [](const void *p) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) return Mem::Device;
            if (at.type == hipMemoryTypeHost && at.devicePointer != nullptr) return Mem::Pinned;
        } else {
            (void)hipGetLastError();                       // unregistered memory reports an error: clear it
        }
        return hsa_device_ptr(p) ? Mem::Device : Mem::Pageable;   // memory of another runtime in the process
    };
    Mem ma = classify(a), mb = classify(b), mab = classify(ab);
    const uint32_t *pa = a, *pb = b;
    uint32_t *pab = ab;
    bool copy_back = false;
    if (own_pinned) {
        if (ma == Mem::Pageable) {
            if (int rc = cx->pin_a.ensure(in_bytes)) return rc;
            std::memcpy(cx->pin_a.ptr, a, in_bytes);
            pa = static_cast<const uint32_t *>(cx->pin_a.ptr);
            ma = Mem::Pinned;
        }
        if (mb == Mem::Pageable) {
            if (int rc = cx->pin_b.ensure(in_bytes)) return rc;
            std::memcpy(cx->pin_b.ptr, b, in_bytes);
            pb = static_cast<const uint32_t *>(cx->pin_b.ptr);
            mb = Mem::Pinned;
        }
        if (mab == Mem::Pageable) {
            if (int rc = cx->pin_out.ensure(out_elems * 4)) return rc;
            pab = static_cast<uint32_t *>(cx->pin_out.ptr);
            if (ab_stride != out_len) std::memcpy(pab, ab, out_elems * 4);      // keep the caller's bytes between rows
            copy_back = true;
            mab = Mem::Pinned;
        }
    }
    const bool da = ma == Mem::Device || (zero_copy && ma == Mem::Pinned), db = mb == Mem::Device || (zero_copy && mb == Mem::Pinned);
    const bool host_out_in_place = zero_copy && mab == Mem::Pinned;
    const bool dab = mab == Mem::Device || host_out_in_place;
    const bool host_in_place = host_out_in_place || ma != Mem::Device || mb != Mem::Device;
    if (!da) {
        int rc = cx->stage_a.ensure(in_bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(cx->stage_a.ptr, pa, in_bytes, hipMemcpyHostToDevice, cx->stream));
        pa = static_cast<const uint32_t *>(cx->stage_a.ptr);
    }
    if (!db) {
        int rc = cx->stage_b.ensure(in_bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(cx->stage_b.ptr, pb, in_bytes, hipMemcpyHostToDevice, cx->stream));
        pb = static_cast<const uint32_t *>(cx->stage_b.ptr);
    }
    uint32_t *const host_dst = pab;                        // where the result goes on the host side (the caller's array, or pin_out)
    if (!dab) {
        int rc = cx->stage_out.ensure(out_elems * 4);
        if (rc) return rc;
        pab = static_cast<uint32_t *>(cx->stage_out.ptr);
        if (ab_stride != out_len)      // keep the caller's bytes between rows: entries >= 2n-1 are not ours
            HIP_TRY(hipMemcpyAsync(pab, host_dst, out_elems * 4, hipMemcpyHostToDevice, cx->stream));
    }
    int rc = multiply_device_any(pl, pa, pb, pab, ab_stride, batch);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    if (int lrc_ = ntt::take_launch_error()) return lrc_;
    if (!dab) {
        HIP_TRY(hipMemcpyAsync(host_dst, pab, out_elems * 4, hipMemcpyDeviceToHost, cx->stream));
        if (int src_ = sync_stream(cx)) return src_;
    } else if (!da || !db || host_in_place) {
        if (int src_ = sync_stream(cx)) return src_;   // host arrays: the call returns when they may be reused / are complete
    }
    if (copy_back) std::memcpy(ab, cx->pin_out.ptr, out_elems * 4);
    return NTT_OK;
}

// Products under several moduli in one call (include/ntt_amd.h): limb i whole on stream lane i % lanes, through plan i's own
// work buffers -- one fork before the first limb and one join behind the last instead of a fork / join per limb.
int ntt_poly_multiply_limbs(ntt_plan *const *plans, int nlimbs, const uint32_t *const *a, const uint32_t *const *b,
                            uint32_t *const *ab, size_t ab_stride, int batch)
try {
    const char *who = "ntt_poly_multiply_limbs";
    if (!plans || !a || !b || !ab || nlimbs <= 0) return fail(NTT_ERR_INVALID, std::string(who) + ": null argument or no limbs");
    if (batch <= 0) return fail(NTT_ERR_INVALID, std::string(who) + ": batch must be positive");
    ntt_ctx *cx = plans[0] ? plans[0]->ctx : nullptr;
    for (int i = 0; i < nlimbs; ++i) {
        ntt_plan *pl = plans[i];
        if (!pl || !a[i] || !b[i] || !ab[i]) return fail(NTT_ERR_INVALID, std::string(who) + ": null plan or array");
        if (!pl->shard.empty()) return fail(NTT_ERR_UNSUPPORTED, std::string(who) + ": plans of a multi-device context shard rows, not limbs: call per slot with plans made on ntt_ctx_device(ctx, slot)");
        if (pl->ctx != cx) return fail(NTT_ERR_INVALID, std::string(who) + ": every plan must come from the same context");
        if (batch > pl->batch) return fail(NTT_ERR_INVALID, std::string(who) + ": batch exceeds a plan's capacity");
        if (ab_stride < (pl->negacyclic ? (size_t)pl->n : 2 * (size_t)pl->n - 1))
            return fail(NTT_ERR_INVALID, std::string(who) + ": output row stride below the row length");
        for (int j = 0; j < i; ++j)
            if (plans[j] == pl) return fail(NTT_ERR_INVALID, std::string(who) + ": a plan may appear once (its work buffers serve one limb at a time): give equal moduli one larger batch");
        if (!is_device_ptr(a[i]) || !is_device_ptr(b[i]) || !is_device_ptr(ab[i]))
            return fail(NTT_ERR_INVALID, std::string(who) + ": device arrays only");
    }
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    int lanes = std::min(cx->nlanes, nlimbs);
    if (cx->lane_limit > 0) lanes = std::min(lanes, cx->lane_limit);
    for (int l = 1; l < lanes; ++l)
        if (!cx->lane[l]) lanes = l;
    hipStream_t main_stream = cx->stream;
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(cx->fork_ev, main_stream));
        for (int l = 1; l < lanes; ++l) HIP_TRY(hipStreamWaitEvent(cx->lane[l], cx->fork_ev, 0));
    }
    int rc = NTT_OK;
    {
        struct Restore {                                   // whatever ends the loop, the context gets its stream back
            ntt_ctx *c;
            hipStream_t s;
            ~Restore() { c->stream = s; c->cur = s; }
        } restore{cx, main_stream};
        for (int i = 0; i < nlimbs && !rc; ++i) {
            ntt_plan *pl = plans[i];
            const int l = i % lanes;
            cx->stream = l == 0 ? main_stream : cx->lane[l];    // the plan's launches follow ctx->stream
            cx->cur = cx->stream;
            const ntt_plan *shape = pl->wide ? pl->wide : pl;
            // the limb has its plan's work buffers to itself (both lanes' halves): chunks up to lanes x chunk products
            int chunk = shape->chunk * std::max(1, shape->lanes);
            if (const char *e = std::getenv("NTT_AMD_LIMB_CHUNK")) chunk = std::max(1, std::min(chunk, std::atoi(e)));
            const int nch = (batch + chunk - 1) / chunk;
            chunk = (batch + nch - 1) / nch;                     // equal shares: no small last launch
            WorkArea area{static_cast<uint32_t *>(shape->work0.ptr), static_cast<uint32_t *>(shape->work1.ptr), chunk};
            const bool needs_area = shape->logL > kRowLog && !shape->fused;
            rc = multiply_device_any(pl, a[i], b[i], ab[i], ab_stride, batch, needs_area ? &area : nullptr);
        }
    }
    // join even after a failure: the next call must not find the lanes unordered
    for (int l = 1; l < lanes; ++l) {
        hipError_t e = hipEventRecord(cx->lane_done[l], cx->lane[l]);
        if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, cx->lane_done[l], 0);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(cx->lane[l]);
            if (!rc) rc = fail(NTT_ERR_HIP, std::string("joining stream lanes: ") + hipGetErrorString(e));
        }
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return NTT_OK;
}
NTT_CATCH_STATUS

// The variant of ntt_poly_multiply_batched that allocates nothing (include/ntt_amd.h): device pointers only,
// intermediates in this device's copy of the static scratch array, one stream lane, no kernel-timing events.
int ntt_poly_multiply_batched_noalloc(ntt_plan *pl, const uint32_t *a, const uint32_t *b, uint32_t *ab,
                                      size_t ab_stride, int batch)
try {
    if (!pl || !a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: null argument");
    if (pl->negacyclic) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: not for negacyclic plans");
    if (!pl->shard.empty()) return fail(NTT_ERR_UNSUPPORTED, "ntt_poly_multiply_batched_noalloc: not for plans of a multi-device context");
    if (batch <= 0) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: batch must be positive");
    if (batch > 65535) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: batch above 65535");
    const size_t n = (size_t)pl->n, out_len = 2 * n - 1;
    if (ab_stride < out_len) return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: output row stride below the row length");
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    if (!is_device_ptr(a) || !is_device_ptr(b) || !is_device_ptr(ab))
        return fail(NTT_ERR_INVALID, "ntt_poly_multiply_batched_noalloc: a, b and ab must be device pointers (staging a host array would allocate)");
    WorkArea area{nullptr, nullptr, batch};
    const ntt_plan *shape = pl->wide ? pl->wide : pl;
    if (shape->logL > kRowLog && !shape->fused) {          // the shapes whose passes hand intermediates over in memory
        const size_t per = 2 * (size_t)shape->L * sizeof(uint32_t);          // both operands of one product
        if (!cx->scratch || per > cx->scratch_bytes)
            return fail(NTT_ERR_UNSUPPORTED, "ntt_poly_multiply_batched_noalloc: transform length " + std::to_string(shape->L) +
                                                 " needs " + std::to_string(per) + " bytes of scratch per product, the static scratch array holds " +
                                                 std::to_string(cx->scratch_bytes) + " (n up to 2^22)");
        area.chunk = (int)std::min<size_t>((size_t)batch, cx->scratch_bytes / per);
        area.w0 = static_cast<uint32_t *>(cx->scratch);
        area.w1 = area.w0 + (size_t)area.chunk * shape->L;
    }
    struct TimingOff {                                     // stamping a launch may create events: not on this path
        ntt_ctx *c;
        bool was;
        explicit TimingOff(ntt_ctx *cx_) : c(cx_), was(cx_->timing) { c->timing = false; }
        ~TimingOff() { c->timing = was; }
    } timing_off(cx);
    // the scratch array belongs to the device: order this call behind the last one that used it there, whichever context
    // and stream that was, and leave the mark for the next
    const bool uses_scratch = area.w0 != nullptr;
    std::unique_lock<std::mutex> sl(g_scratch_mu, std::defer_lock);
    if (uses_scratch) {
        sl.lock();
        HIP_TRY(hipStreamWaitEvent(cx->stream, g_scratch_last[cx->device], 0));
    }
    const int rc = pl->kind == NTT_MODMUL_MONTGOMERY ? multiply_device<Mont32>(pl, a, b, ab, ab_stride, batch, &area)
                                                     : multiply_device<Barrett32>(pl, a, b, ab, ab_stride, batch, &area);
    if (uses_scratch) {                                    // also after a failure: what was launched still uses the array
        const hipError_t re = hipEventRecord(g_scratch_last[cx->device], cx->stream);
        if (re != hipSuccess && rc == NTT_OK) return fail(NTT_ERR_HIP, std::string("hipEventRecord(scratch order): ") + hipGetErrorString(re));
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return take_launch_error();
}
NTT_CATCH_STATUS

static int transform_entry(ntt_plan *pl, uint32_t *data, int batch, bool inverse)
{
    if (!pl || !data) return fail(NTT_ERR_INVALID, "ntt_forward/inverse: null argument");
    if (!pl->shard.empty()) return group_transform(pl, data, batch, inverse);
    if (batch <= 0) return fail(NTT_ERR_INVALID, "ntt_forward/inverse: batch must be positive");
    if (batch > pl->batch) return fail(NTT_ERR_INVALID, "ntt_forward/inverse: batch exceeds the plan's capacity");
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    const bool dev = is_device_ptr(data);
    uint32_t *d = data;
    const size_t bytes = (size_t)batch * pl->L * 4;
    if (!dev) {
        int rc = cx->stage_a.ensure(bytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(cx->stage_a.ptr, data, bytes, hipMemcpyHostToDevice, cx->stream));
        d = static_cast<uint32_t *>(cx->stage_a.ptr);
    }
    int rc = pl->kind == NTT_MODMUL_MONTGOMERY ? transform_device<Mont32>(pl, d, d, batch, inverse)
                                               : transform_device<Barrett32>(pl, d, d, batch, inverse);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    if (int lrc_ = ntt::take_launch_error()) return lrc_;
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, cx->stream));
        HIP_TRY(hipStreamSynchronize(cx->stream));
    }
    return NTT_OK;
}

int ntt_forward(ntt_plan *plan, uint32_t *data, int batch)
try {
    return transform_entry(plan, data, batch, false);
}
NTT_CATCH_STATUS
int ntt_inverse(ntt_plan *plan, uint32_t *data, int batch)
try {
    return transform_entry(plan, data, batch, true);
}
NTT_CATCH_STATUS

}  // extern "C"

// ================================================================================================
// Distributed single transform: the pieces one rank runs between the all-to-alls
// ================================================================================================
namespace ntt {

template <class MM>
static int build_dist_tables(ntt_dist_plan *dp, uint32_t p, uint32_t g)
{
    ntt_plan *pl = dp->colp;
    const ModParams &mp = pl->mp;
    const uint64_t L = 1ULL << dp->lgL;
    const uint32_t G = (uint32_t)dp->world, cols = dp->S / G, off = (uint32_t)dp->rank * cols;
    const uint32_t wL = h_powmod(g, (p - 1) / L, p), iwL = h_powmod(wL, p - 2, p);
    const uint32_t invG = h_powmod(G % p, p - 2, p);
    dp->scale = bytes_of_tw<MM>(invG, mp);
    Roots64<MM> r{}, ir{};
    const uint32_t w64 = h_powmod(wL, L / 64, p), iw64 = h_powmod(iwL, L / 64, p);
    uint32_t c = 1, ic = 1;
    for (int j = 0; j < 32; ++j) {
        r.w[j] = MM::make_tw(c, mp);
        ir.w[j] = MM::make_tw(ic, mp);
        c = h_mulmod(c, w64, p);
        ic = h_mulmod(ic, iw64, p);
    }
    pl->roots.assign((unsigned char *)&r, (unsigned char *)&r + sizeof(r));
    pl->iroots.assign((unsigned char *)&ir, (unsigned char *)&ir + sizeof(ir));
    if (pl->level_log[0] >= 5) {
        const uint32_t wN = h_powmod(wL, L / G, p), iwN = h_powmod(iwL, L / G, p);
        typename MM::tw_t f[64], b[64];
        for (uint32_t rr = 0; rr < 4; ++rr)
            for (uint32_t ka = 0; ka < 16; ++ka) {
                f[rr * 16 + ka] = MM::make_tw(h_powmod(wN, (uint64_t)rr * ka, p), mp);
                b[rr * 16 + ka] = MM::make_tw(h_powmod(iwN, (uint64_t)rr * ka, p), mp);
            }
        pl->tw_in_fwd[0].assign((unsigned char *)f, (unsigned char *)f + sizeof(f));
        pl->tw_in_inv[0].assign((unsigned char *)b, (unsigned char *)b + sizeof(b));
    }
    // tw[k1][j] = w_L^((off + j) * k1); the inverse table carries world^-1
    std::vector<typename MM::tw_t> tf((size_t)G * cols), ti((size_t)G * cols);
    for (uint32_t k1 = 0; k1 < G; ++k1) {
        const uint32_t wk = h_powmod(wL, k1, p), iwk = h_powmod(iwL, k1, p);
        uint32_t cur = h_powmod(wk, off, p), icur = h_mulmod(h_powmod(iwk, off, p), invG, p);
        for (uint32_t j = 0; j < cols; ++j) {
            tf[(size_t)k1 * cols + j] = MM::make_tw(cur, mp);
            ti[(size_t)k1 * cols + j] = MM::make_tw(icur, mp);
            cur = h_mulmod(cur, wk, p);
            icur = h_mulmod(icur, iwk, p);
        }
    }
    int rc = upload_table<MM>(pl, tf, &pl->tw_fwd[0]);
    if (rc) return rc;
    return upload_table<MM>(pl, ti, &pl->tw_inv[0]);
}

}  // namespace ntt

extern "C" {

int ntt_dist_plan_destroy(ntt_dist_plan *dp)
try {
    if (!dp) return NTT_OK;
    if (!dp->part.empty() || !dp->gslot.empty()) group_dist_plan_release(dp);
    if (dp->local) ntt_plan_destroy(dp->local);
    if (dp->colp) ntt_plan_destroy(dp->colp);
    delete dp;
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_dist_plan_create(ntt_ctx *ctx, uint32_t p, int lgL, int world, int rank, int modmul_kind,
                         ntt_dist_plan **out)
try {
    if (!ctx || !out) return fail(NTT_ERR_INVALID, "ntt_dist_plan_create: null argument");
    *out = nullptr;
    if (ctx->is_group()) return group_dist_plan_create(ctx, p, lgL, world, modmul_kind, out);    // every rank's pieces; `rank` is not used
    if (world < 2 || world > 64 || (world & (world - 1)) || rank < 0 || rank >= world)
        return fail(NTT_ERR_INVALID, "ntt_dist_plan_create: world must be a power of two in 2..64, 0 <= rank < world");
    int lgG = 0;
    while ((1 << lgG) < world) ++lgG;
    if (lgL > 30 || lgL < 8 + 2 * lgG)
        return fail(NTT_ERR_UNSUPPORTED, "ntt_dist_plan_create: need 256 * world^2 <= L <= 2^30");
    if (p >= (1u << 31) || p < 3 || !h_is_prime(p) || ((uint64_t)(p - 1) % (1ULL << lgL)) != 0)
        return fail(NTT_ERR_UNSUPPORTED, "ntt_dist_plan_create: p must be a prime < 2^31 with 2^lgL | p - 1");
    if (modmul_kind != NTT_MODMUL_MONTGOMERY && modmul_kind != NTT_MODMUL_BARRETT)
        return fail(NTT_ERR_INVALID, "ntt_dist_plan_create: unknown modmul_kind");
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<ntt_dist_plan> dp(new ntt_dist_plan());
    dp->ctx = ctx;
    dp->lgL = lgL;
    dp->world = world;
    dp->rank = rank;
    dp->S = 1u << (lgL - lgG);
    // local S-point transforms: same generator, so its root w_S equals w_L^world
    int rc = ntt_plan_create(ctx, p, (int)(dp->S / 2), 1, modmul_kind, &dp->local);
    if (rc) return rc;
    ntt_plan *cp = new ntt_plan();
    dp->colp = cp;
    cp->ctx = ctx;
    cp->p = p;
    cp->g = dp->local->g;
    cp->n = (int)(dp->S / 2);
    cp->logL = lgL - lgG;                 // the column pass sees one block = the whole local array
    cp->L = dp->S;
    cp->batch = 1;
    cp->kind = modmul_kind;
    cp->mp = dp->local->mp;
    cp->lazy = dp->local->lazy;
    cp->nlevels = 1;
    cp->level_log[0] = lgG;
    cp->level_logM[0] = lgL - lgG;
    cp->col16_fwd = cp->col16_inv = lgG >= 5;
    rc = modmul_kind == NTT_MODMUL_MONTGOMERY ? build_dist_tables<Mont32>(dp.get(), p, cp->g)
                                              : build_dist_tables<Barrett32>(dp.get(), p, cp->g);
    if (rc) { ntt_dist_plan_destroy(dp.release()); return rc; }
    *out = dp.release();
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_dist_colpass(ntt_dist_plan *dp, uint32_t *data, int inverse)
try {
    if (!dp || !data) return fail(NTT_ERR_INVALID, "ntt_dist_colpass: null argument");
    if (!dp->part.empty()) return fail(NTT_ERR_INVALID, "ntt_dist_colpass: plan of a multi-device context (its exchanges run inside ntt_dist_forward / ntt_dist_inverse)");
    ntt_ctx *cx = dp->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    if (!is_device_ptr(data)) return fail(NTT_ERR_INVALID, "ntt_dist_colpass: data must be a device pointer");
    ntt_plan *pl = dp->colp;
    cx->cur = cx->stream;
    const uint32_t S = dp->S;
    const bool inv = inverse != 0;
    if (pl->kind == NTT_MODMUL_MONTGOMERY)
        col_pass<Mont32>(pl, 0, inv, false, data, nullptr, data, nullptr, S, S, S, S,
                         inv ? pl->tw_inv[0] : pl->tw_fwd[0], inv, dp->scale, true, 1, 1);
    else
        col_pass<Barrett32>(pl, 0, inv, false, data, nullptr, data, nullptr, S, S, S, S,
                            inv ? pl->tw_inv[0] : pl->tw_fwd[0], inv, dp->scale, true, 1, 1);
    HIP_TRY(hipGetLastError());
    if (int lrc_ = ntt::take_launch_error()) return lrc_;
    return NTT_OK;
}
NTT_CATCH_STATUS

int ntt_dist_local(ntt_dist_plan *dp, uint32_t *row, int inverse)
try {
    if (!dp) return fail(NTT_ERR_INVALID, "ntt_dist_local: null argument");
    if (!dp->part.empty()) return fail(NTT_ERR_INVALID, "ntt_dist_local: plan of a multi-device context (use ntt_dist_forward / ntt_dist_inverse)");
    return inverse ? ntt_inverse(dp->local, row, 1) : ntt_forward(dp->local, row, 1);
}
NTT_CATCH_STATUS

int ntt_dist_local_length(const ntt_dist_plan *dp) { return dp ? (int)dp->S : 0; }

}  // extern "C"
