// direct64.cpp -- plans and launch sequence of the direct 64-bit path (kernels_u64.hip): polynomial
// product modulo ONE NTT-friendly prime p64 < 2^62 with 64-bit Montgomery lanes.
//
// Reference counterpart: the single-modulus branch of main-openmp.cc:678-692 (toMontgomery,
// NTT_iterative x2, pointwise ModMul, inverse, fromMontgomery) with `montgomery m(R = 2^60, p_)`
// (:650-651), which the reference only takes for p_ < 2^50 and sends larger moduli through the
// 4-prime CRT (:693-711, here ntt_crt_multiply).  This is the same branch with lanes wide enough
// for the 51-bit modulus of the reference's fifth test case, so that case can be computed two ways.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/ntt_amd.h"
#include "engine.hpp"

using namespace ntt;
typedef unsigned __int128 u128;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ntt::fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct ntt_plan64 {
    ntt_ctx *ctx = nullptr;
    uint64_t p = 0, g = 0;
    int n = 0, logL = 0, batch = 0;
    uint32_t L = 0;
    ModParams64 mp{};
    int nlevels = 0;
    int level_log[8] = {0}, level_logM[8] = {0};
    uint64_t *tw_fwd[8] = {nullptr}, *tw_inv[8] = {nullptr};
    uint64_t *row_w = nullptr, *row_iw = nullptr;             // lds64_kernel (L <= 4096)
    uint64_t *twA = nullptr, *twB = nullptr, *itwA = nullptr, *itwB = nullptr;   // row64_mul_kernel
    Roots16x64 roots{}, iroots{};
    std::vector<uint64_t> tw_in_fwd[8], tw_in_inv[8];         // per level of 32 / 64 points: w_N1^(+-r*ka)
    uint64_t scale = 0;                     // L^-1 * R (cancels the pointwise R^-1), Montgomery form
    bool negacyclic = false;                // products modulo x^n + 1: L = n, psi twist
    bool fold2048 = false;                  // x^2048 + 1 through the linear product of L = 4096, folded at the store
    uint64_t *twist_fwd = nullptr, *twist_inv = nullptr, *small_itwist = nullptr;
    DeviceBuffer work0, work1, stage_a, stage_b, stage_out;
    std::vector<void *> owned;
};

namespace {

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((u128)a * b % p); }
uint64_t powmod(uint64_t b, uint64_t e, uint64_t p)
{
    uint64_t r = 1 % p;
    b %= p;
    for (; e; e >>= 1, b = mulmod(b, b, p))
        if (e & 1) r = mulmod(r, b, p);
    return r;
}
// deterministic Miller-Rabin for 64-bit integers (the first twelve primes as bases)
bool is_prime64_impl(uint64_t n)
{
    if (n < 2) return false;
    for (uint64_t q : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n % q == 0) return n == q;
    }
    uint64_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s && comp; ++i) {
            x = mulmod(x, x, n);
            if (x == n - 1) comp = false;
        }
        if (comp) return false;
    }
    return true;
}
// table entry for the residue x: Montgomery form, or the residue itself as a double (F64 policy)
uint64_t to_mont(uint64_t x, const ModParams64 &mp)
{
    if (mp.fp) {
        const double d = (double)x;                     // x < p < 2^51: exact
        uint64_t bits;
        std::memcpy(&bits, &d, 8);
        return bits;
    }
    return (uint64_t)((((u128)x) << 64) % mp.p);
}

ModParams64 make_params(uint64_t p)
{
    ModParams64 m{};
    m.p = p;
    m.p2 = 2 * p;
    uint64_t inv = 1;                                   // Newton: p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    m.nprime = 0 - inv;
    m.r1 = (uint64_t)((((u128)1) << 64) % p);
    m.r2 = mulmod(m.r1, m.r1, p);
    // p < 2^51: exact integer arithmetic in double-precision lanes (kernels_u64.hip, F64); NTT_AMD_PLAN64_INT=1
    // keeps such a prime on the Montgomery lanes (A/B runs, tests of both policies on one prime)
    const char *e = std::getenv("NTT_AMD_PLAN64_INT");
    // NTT_AMD_PLAN64_INT=2 keeps a prime below 2^50 on the general FP64 policy (tests of all three on one prime)
    const int force = e ? std::atoi(e) : 0;
    m.fp = force == 1 ? 0 : p < (1ull << 50) && force != 2 ? 2 : p < (1ull << 51) ? 1 : 0;
    m.pd = (double)p;
    m.pd2 = 2.0 * (double)p;
    m.pinv = 1.0 / (double)p;
    if (m.fp) m.r1 = 1;                                 // no Montgomery factor to cancel in the scale
    return m;
}

// T[k][i] = c * w^(i*k) in Montgomery form, k < rows, i < cols
std::vector<uint64_t> power_table(uint64_t w, uint32_t rows, uint32_t cols, uint64_t c, const ModParams64 &mp)
{
    std::vector<uint64_t> t((size_t)rows * cols);
    uint64_t wk = 1;
    for (uint32_t k = 0; k < rows; ++k) {
        uint64_t cur = c;
        for (uint32_t i = 0; i < cols; ++i) {
            t[(size_t)k * cols + i] = to_mont(cur, mp);
            cur = mulmod(cur, wk, mp.p);
        }
        wk = mulmod(wk, w, mp.p);
    }
    return t;
}

int upload(ntt_plan64 *pl, const std::vector<uint64_t> &h, uint64_t **dptr)
{
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, h.size() * 8));
    pl->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    *dptr = static_cast<uint64_t *>(d);
    return NTT_OK;
}

int build_tables(ntt_plan64 *pl)
{
    const ModParams64 &mp = pl->mp;
    const uint64_t p = pl->p;
    const uint32_t L = pl->L;
    const uint64_t wL = powmod(pl->g, (p - 1) / L, p), iwL = powmod(wL, p - 2, p);
    const uint64_t invL = powmod(L % p, p - 2, p);
    const uint64_t scale = mulmod(invL, mp.r1, p);                  // L^-1 * R
    pl->scale = to_mont(scale, mp);
    if (pl->negacyclic && !pl->fold2048) {      // psi = g^((p-1)/2L), psi^2 = wL; row 1 of a 2-row power table is psi^i
        const uint64_t psi = powmod(pl->g, (p - 1) / (2ull * L), p), ipsi = powmod(psi, p - 2, p);
        int rc = upload(pl, power_table(psi, 2, L, 1, mp), &pl->twist_fwd);
        if (!rc) rc = upload(pl, power_table(ipsi, 2, L, 1, mp), &pl->twist_inv);
        if (!rc) rc = upload(pl, power_table(ipsi, 2, L, scale, mp), &pl->small_itwist);
        if (rc) return rc;
        pl->twist_fwd += L;
        pl->twist_inv += L;
        pl->small_itwist += L;
    }
    const int rowlog = std::min(pl->logL, kRowLog);
    const uint32_t len = 1u << rowlog;
    // row tables: w_len^j, j < len/2
    {
        const uint64_t wr = powmod(wL, L / len, p), iwr = powmod(iwL, L / len, p);
        std::vector<uint64_t> w(std::max<uint32_t>(len / 2, 1)), iw(std::max<uint32_t>(len / 2, 1));
        uint64_t c = 1, ic = 1;
        for (uint32_t j = 0; j < std::max<uint32_t>(len / 2, 1); ++j) {
            w[j] = to_mont(c, mp);
            iw[j] = to_mont(ic, mp);
            c = mulmod(c, wr, p);
            ic = mulmod(ic, iwr, p);
        }
        int rc = upload(pl, w, &pl->row_w);
        if (rc) return rc;
        if ((rc = upload(pl, iw, &pl->row_iw))) return rc;
    }
    if (pl->nlevels == 0 && pl->logL != kRowLog) return NTT_OK;
    {   // 4096-point row kernel: pass twiddles w_4096^(k*t), w_256^(k*i) and their inverses
        const uint64_t w4096 = powmod(wL, L / kRowLen, p), iw4096 = powmod(iwL, L / kRowLen, p);
        const uint64_t w256 = powmod(w4096, 16, p), iw256 = powmod(iw4096, 16, p);
        int rc = upload(pl, power_table(w4096, 16, 256, 1, mp), &pl->twA);
        if (!rc) rc = upload(pl, power_table(w256, 16, 16, 1, mp), &pl->twB);
        if (!rc) rc = upload(pl, power_table(iw4096, 16, 256, 1, mp), &pl->itwA);
        if (!rc) rc = upload(pl, power_table(iw256, 16, 16, 1, mp), &pl->itwB);
        if (rc) return rc;
    }
    const uint64_t w16 = powmod(wL, L / 16, p), iw16 = powmod(iwL, L / 16, p);
    uint64_t c = 1, ic = 1;
    for (int j = 0; j < 8; ++j) {
        pl->roots.w[j] = to_mont(c, mp);
        pl->iroots.w[j] = to_mont(ic, mp);
        c = mulmod(c, w16, p);
        ic = mulmod(ic, iw16, p);
    }
    for (int j = 0; j < pl->nlevels; ++j) {
        const uint32_t M = 1u << pl->level_logM[j], N1 = 1u << pl->level_log[j], S = M / N1;
        const uint64_t wM = powmod(wL, L / M, p), iwM = powmod(iwL, L / M, p);
        const bool innermost = (j == pl->nlevels - 1);
        if (pl->level_log[j] >= 5) {                     // inner twiddles of the LDS-staged level
            const uint64_t wN = powmod(wL, L / N1, p), iwN = powmod(iwL, L / N1, p);
            pl->tw_in_fwd[j].assign(128, 0);
            pl->tw_in_inv[j].assign(128, 0);
            for (uint32_t r = 0; r < std::max(4u, N1 / 16); ++r)
                for (uint32_t ka = 0; ka < 16; ++ka) {
                    pl->tw_in_fwd[j][r * 16 + ka] = to_mont(powmod(wN, (uint64_t)r * ka, p), mp);
                    pl->tw_in_inv[j][r * 16 + ka] = to_mont(powmod(iwN, (uint64_t)r * ka, p), mp);
                }
        }
        int rc = upload(pl, power_table(wM, N1, S, 1, mp), &pl->tw_fwd[j]);
        if (rc) return rc;
        if ((rc = upload(pl, power_table(iwM, N1, S, innermost ? scale : 1, mp), &pl->tw_inv[j]))) return rc;
    }
    return NTT_OK;
}

Col64Args fill_col(ntt_plan64 *pl, int level, bool inverse, const uint64_t *in0, const uint64_t *in1, uint64_t *out0,
                   uint64_t *out1, size_t in_stride, size_t out_stride, uint32_t in_len, uint32_t out_len, bool canon)
{
    Col64Args a{};
    a.in[0] = in0; a.in[1] = in1;
    a.out[0] = out0; a.out[1] = out1;
    a.in_stride = in_stride; a.out_stride = out_stride;
    a.in_len = in_len; a.out_len = out_len;
    a.logM = pl->level_logM[level];
    a.logS = pl->level_logM[level] - pl->level_log[level];
    a.tw = inverse ? pl->tw_inv[level] : pl->tw_fwd[level];
    a.use_scale0 = (inverse && level == pl->nlevels - 1) ? 1 : 0;
    a.scale0 = pl->scale;
    a.canon = canon ? 1 : 0;
    a.ext_in = (!inverse && level == 0) ? 1 : 0;
    a.roots = inverse ? pl->iroots : pl->roots;
    a.twist = (pl->negacyclic && !pl->fold2048 && level == 0) ? (inverse ? pl->twist_inv : pl->twist_fwd) : nullptr;
    if (pl->level_log[level] >= 5) {
        const auto &src = inverse ? pl->tw_in_inv[level] : pl->tw_in_fwd[level];
        std::memcpy(a.tw_in, src.data(), sizeof(a.tw_in));
    }
    return a;
}

// device pointers only; caller holds ctx->mu
int multiply_device(ntt_plan64 *pl, const uint64_t *a, const uint64_t *b, uint64_t *ab, size_t ab_stride, int batch)
{
    ntt_ctx *cx = pl->ctx;
    const bool neg = pl->negacyclic && !pl->fold2048;
    const uint32_t n = (uint32_t)pl->n, L = pl->L, out_len = pl->negacyclic ? n : 2 * n - 1;
    hipStream_t s = cx->stream;
    cx->cur = s;
    if (pl->nlevels == 0 && pl->logL == kRowLog && pl->twA) {      // L = 4096: the row kernel does the whole product
        Row64ProdArgs r{};
        r.row.a = a; r.row.b = b; r.row.out = ab;
        r.row.twA = pl->twA; r.row.twB = pl->twB; r.row.itwA = pl->itwA; r.row.itwB = pl->itwB;
        r.row.roots = pl->roots; r.row.iroots = pl->iroots;
        r.in_stride = n; r.out_stride = ab_stride;
        r.in_len = n; r.out_len = out_len;
        r.scale = pl->scale;
        r.fold = pl->fold2048 ? 1 : 0;
        if (neg) { r.twist = pl->twist_fwd; r.itwist = pl->small_itwist; }
        cx->begin(K_LDS64);
        launch_row64_product(r, pl->mp, (unsigned)batch, s);
        cx->end();
        HIP_TRY(hipGetLastError());
        if (int lrc_ = ntt::take_launch_error()) return lrc_;
        return NTT_OK;
    }
    if (pl->nlevels == 0) {
        Lds64Args r{};
        r.a = a; r.b = b; r.out = ab;
        r.in_stride = n; r.out_stride = ab_stride;
        r.in_len = n; r.out_len = out_len;
        r.blocks_per_row = 1; r.logLen = pl->logL;
        r.w = pl->row_w; r.iw = pl->row_iw;
        r.scale = pl->scale; r.use_scale = 1; r.canon = 1; r.mode = 0; r.ext_in = 1;
        if (neg) { r.twist = pl->twist_fwd; r.itwist = pl->small_itwist; }
        cx->begin(K_LDS64);
        launch_lds64(r, pl->mp, (unsigned)batch, s);
        cx->end();
        HIP_TRY(hipGetLastError());
        if (int lrc_ = ntt::take_launch_error()) return lrc_;
        return NTT_OK;
    }
    // equal chunks of at most 96 MiB per work buffer (48 products at L = 2^18): the intermediates of a launch
    // sequence stay inside the 256 MiB Infinity Cache (tools/sweep_batch64.py: 5.25 us per product at 32-48 per
    // launch, 5.6 at 128), and the work buffers do not grow with the batch
    // Round 3: the chunks alternate between the context's two stream lanes, as in the 32-bit path -- the memory-bound
    // column kernels of one chunk run beside the FP64-bound row kernel of the other (NTT_AMD_LANES64=1, read per call,
    // keeps one lane).  Each lane has its own pair of work buffers; a lane's chunk is capped lower so that the
    // intermediates of both lanes still fit the cache (NTT_AMD_CHUNK64 overrides the cap, for sweeps).
    int want_lanes = std::min(cx->nlanes, 2);
    if (cx->lane_limit > 0) want_lanes = std::min(want_lanes, cx->lane_limit);
    if (want_lanes > 1 && !cx->lane[1]) want_lanes = 1;
    if (const char *e = std::getenv("NTT_AMD_LANES64")) want_lanes = std::max(1, std::min(want_lanes, std::atoi(e)));
    size_t cap_bytes = (size_t)(want_lanes > 1 ? 48 : 96) << 20;
    int cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)pl->batch, cap_bytes / ((size_t)L * 8)));
    if (const char *e = std::getenv("NTT_AMD_CHUNK64")) cap = std::max(1, std::min(pl->batch, std::atoi(e)));
    int nchunks = (batch + cap - 1) / cap;
    const int lanes = std::min(want_lanes, nchunks);
    if (lanes == 2 && (nchunks & 1) && nchunks <= 5) ++nchunks;          // a few chunks only: the same number on both lanes
    const int chunk = (batch + nchunks - 1) / nchunks;                   // equal shares, none above cap
    const size_t per_lane = (size_t)cap * L * 8;
    int rc = pl->work0.ensure(per_lane * (size_t)want_lanes);
    if (!rc) rc = pl->work1.ensure(per_lane * (size_t)want_lanes);
    if (rc) return rc;
    const int m = pl->nlevels;
    struct LaneJoin {                                                    // whatever ends the loop: lane 1 is joined, the context gets its stream back
        ntt_ctx *cx;
        hipStream_t main;
        bool forked = false;
        int join()
        {
            cx->cur = main;
            if (!forked) return NTT_OK;
            forked = false;
            hipError_t e = hipEventRecord(cx->lane_done[1], cx->lane[1]);
            if (e == hipSuccess) e = hipStreamWaitEvent(main, cx->lane_done[1], 0);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(cx->lane[1]);
                return fail(NTT_ERR_HIP, std::string("joining stream lanes: ") + hipGetErrorString(e));
            }
            return NTT_OK;
        }
        ~LaneJoin() { (void)join(); }
    } lanes_guard{cx, s};
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(cx->fork_ev, s));
        lanes_guard.forked = true;
        HIP_TRY(hipStreamWaitEvent(cx->lane[1], cx->fork_ev, 0));
    }
    const hipStream_t main_stream = s;
    int ci = 0;
    for (int c0 = 0; c0 < batch; c0 += chunk, ++ci) {
        const int lane = ci % lanes;
        s = lane ? cx->lane[1] : main_stream;
        cx->cur = s;
        uint64_t *w0 = reinterpret_cast<uint64_t *>(static_cast<char *>(pl->work0.ptr) + (size_t)lane * per_lane);
        uint64_t *w1 = reinterpret_cast<uint64_t *>(static_cast<char *>(pl->work1.ptr) + (size_t)lane * per_lane);
        const unsigned cb = (unsigned)std::min(chunk, batch - c0);
        const uint64_t *ca = a + (size_t)c0 * n, *cbp = b + (size_t)c0 * n;
        uint64_t *cout = ab + (size_t)c0 * ab_stride;
        // forward column levels, outermost first (operands a and b in one grid)
        for (int j = 0; j < m; ++j) {
            const Col64Args ca_ = j == 0 ? fill_col(pl, 0, false, ca, cbp, w0, w1, n, L, n, L, false)
                                         : fill_col(pl, j, false, w0, w1, w0, w1, L, L, L, L, false);
            cx->begin(K_COL64);
            (pl->level_log[j] >= 5 ? launch_col64x : launch_col64)(pl->level_log[j], false, j == 0 && !neg, ca_, pl->mp,
                                                                   L >> pl->level_log[j], cb, 2, s);
            cx->end();
        }
        // rows: forward a, forward b, pointwise, inverse
        {
            Row64Args r{};
            r.a = w0; r.b = w1; r.out = w0;
            r.twA = pl->twA; r.twB = pl->twB; r.itwA = pl->itwA; r.itwB = pl->itwB;
            r.roots = pl->roots; r.iroots = pl->iroots;
            cx->begin(K_LDS64);
            launch_row64_mul(r, pl->mp, cb * (L >> kRowLog), s);
            cx->end();
        }
        // inverse column levels, innermost first; the outermost writes the caller's ab
        for (int j = m - 1; j >= 0; --j) {
            const Col64Args ca_ = j == 0 ? fill_col(pl, 0, true, w0, nullptr, cout, nullptr, L, ab_stride, L, out_len, true)
                                         : fill_col(pl, j, true, w0, nullptr, w0, nullptr, L, L, L, L, false);
            cx->begin(K_COL64);
            (pl->level_log[j] >= 5 ? launch_col64x : launch_col64)(pl->level_log[j], true, false, ca_, pl->mp,
                                                                   L >> pl->level_log[j], cb, 1, s);
            cx->end();
        }
    }
    if ((rc = lanes_guard.join())) return rc;
    HIP_TRY(hipGetLastError());
    if (int lrc_ = ntt::take_launch_error()) return lrc_;
    return NTT_OK;
}

}  // namespace

namespace ntt {
bool is_prime64(uint64_t n) { return is_prime64_impl(n); }
}
using ntt::is_prime64;

static int plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, bool negacyclic, ntt_plan64 **out);

extern "C" int ntt_plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, ntt_plan64 **out)
try {
    return plan64_create(ctx, p64, n, batch, false, out);
}
NTT_CATCH_STATUS

extern "C" int ntt_negacyclic_plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, ntt_plan64 **out)
try {
    if (n < 2 || (n & (n - 1)) != 0) return fail(NTT_ERR_INVALID, "ntt_negacyclic_plan64_create: n must be a power of two >= 2");
    return plan64_create(ctx, p64, n, batch, true, out);
}
NTT_CATCH_STATUS

static int plan64_create(ntt_ctx *ctx, uint64_t p64, int n, int batch, bool negacyclic, ntt_plan64 **out)
{
    if (!ctx || !out) return fail(NTT_ERR_INVALID, "ntt_plan64_create: null argument");
    if (ctx->is_group()) return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: 64-bit plans live on one slot: use ntt_ctx_device(ctx, slot)");
    if (n <= 0 || batch <= 0) return fail(NTT_ERR_INVALID, "ntt_plan64_create: n and batch must be positive");
    if (n > (1 << 23)) return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: n above 2^23");
    if (p64 < 3 || p64 >= (1ULL << 62) || !is_prime64(p64))
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: p64 must be a prime below 2^62");
    int lg = 1;
    while ((1ull << lg) < 2ull * (uint64_t)n) ++lg;                // smallest power of two >= 2n, at least 2
    // x^2048 + 1: the linear product of L = 4096 (no twist), coefficients i and i + 2048 subtracted at the store
    const bool fold2048 = negacyclic && n == 2048 && (p64 - 1) % (1ull << kRowLog) == 0;
    if (fold2048) negacyclic = false;                              // ... for the rest of the plan's construction
    if (negacyclic) --lg;                                          // x^n + 1: transform length n, 2n-th root psi
    // a linear product of L = 2048 runs faster as one of L = 4096 in the row kernel than at its own length in the
    // radix-2 LDS kernel (0.046 vs 0.054 us): pad further when the prime allows
    if (!negacyclic && lg == kRowLog - 1 && (p64 - 1) % (1ull << kRowLog) == 0) lg = kRowLog;
    if ((p64 - 1) % (1ull << (negacyclic ? lg + 1 : lg)) != 0)
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: p64 - 1 is not divisible by the transform length");
    if ((uint64_t)batch << lg > (1ull << 29))
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: batch * L above 2^29 elements");
    if (lg > kRowLog && batch > 65535)                            // the column kernels index the batch with blockIdx.y
        return fail(NTT_ERR_UNSUPPORTED, "ntt_plan64_create: batch above 65535 for a transform longer than 4096");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    auto *pl = new ntt_plan64();
    pl->ctx = ctx;
    pl->p = p64;
    pl->n = n;
    pl->batch = batch;
    pl->negacyclic = negacyclic || fold2048;
    pl->fold2048 = fold2048;
    pl->logL = lg;
    pl->L = 1u << lg;
    pl->mp = make_params(p64);
    // a quadratic non-residue g has g^((p-1)/L) of exact order L for every power of two L | p-1
    // (the reference's primes all take g = 3, main-openmp.cc:635; so does 1337006139375617)
    uint64_t g = 2;
    while (powmod(g, (p64 - 1) / 2, p64) != p64 - 1) ++g;
    pl->g = g;
    if (lg > kRowLog) {
        // column levels of at most 64 points: 5 or 6 bits go through the LDS-staged kernel (one pass over
        // memory), up to 4 bits through the register kernel
        int maxlog = 7;
        if (const char *e = std::getenv("NTT_AMD_COL64_MAXLOG")) maxlog = std::max(1, std::min(7, std::atoi(e)));
        const int c = lg - kRowLog, m = (c + maxlog - 1) / maxlog;
        int logM = lg;
        for (int j = 0; j < m; ++j) {
            pl->level_log[j] = c / m + (j < c % m ? 1 : 0);
            pl->level_logM[j] = logM;
            logM -= pl->level_log[j];
        }
        pl->nlevels = m;
    }
    int rc = build_tables(pl);
    if (rc) {
        ntt_plan64_destroy(pl);
        return rc;
    }
    *out = pl;
    return NTT_OK;
}

extern "C" int ntt_plan64_destroy(ntt_plan64 *pl)
try {
    if (!pl) return NTT_OK;
    (void)hipSetDevice(pl->ctx->device);
    (void)hipStreamSynchronize(pl->ctx->stream);
    for (void *d : pl->owned) (void)hipFree(d);
    pl->work0.release(); pl->work1.release();
    pl->stage_a.release(); pl->stage_b.release(); pl->stage_out.release();
    delete pl;
    return NTT_OK;
}
NTT_CATCH_STATUS

extern "C" int ntt_plan64_length(const ntt_plan64 *pl, uint32_t *L)
try {
    if (!pl || !L) return fail(NTT_ERR_INVALID, "ntt_plan64_length: null argument");
    *L = pl->L;
    return NTT_OK;
}
NTT_CATCH_STATUS

static int multiply64_entry(ntt_plan64 *pl, const uint64_t *a, const uint64_t *b, uint64_t *ab, size_t ab_stride,
                            int batch, bool negacyclic);

extern "C" int ntt_poly_multiply64_batched(ntt_plan64 *pl, const uint64_t *a, const uint64_t *b, uint64_t *ab,
                                           size_t ab_stride, int batch)
try {
    return multiply64_entry(pl, a, b, ab, ab_stride, batch, false);
}
NTT_CATCH_STATUS

extern "C" int ntt_negacyclic_multiply64_batched(ntt_plan64 *pl, const uint64_t *a, const uint64_t *b, uint64_t *c,
                                                 size_t c_stride, int batch)
try {
    return multiply64_entry(pl, a, b, c, c_stride, batch, true);
}
NTT_CATCH_STATUS

static int multiply64_entry(ntt_plan64 *pl, const uint64_t *a, const uint64_t *b, uint64_t *ab, size_t ab_stride,
                            int batch, bool negacyclic)
{
    if (!pl || !a || !b || !ab) return fail(NTT_ERR_INVALID, "ntt_poly_multiply64_batched: null argument");
    if (pl->negacyclic != negacyclic)
        return fail(NTT_ERR_INVALID, "ntt_poly_multiply64_batched / ntt_negacyclic_multiply64_batched: wrong kind of plan");
    if (batch <= 0 || batch > pl->batch)
        return fail(NTT_ERR_INVALID, "ntt_poly_multiply64_batched: batch must be in 1..plan capacity");
    const size_t n = (size_t)pl->n, out_len = negacyclic ? n : 2 * n - 1;
    if (ab_stride < out_len) return fail(NTT_ERR_INVALID, "ntt_poly_multiply64_batched: output row stride below the row length");
    ntt_ctx *cx = pl->ctx;
    std::lock_guard<std::mutex> lk(cx->mu);
    HIP_TRY(hipSetDevice(cx->device));
    const bool dev_a = is_device_ptr(a), dev_b = is_device_ptr(b), dev_o = is_device_ptr(ab);
    const uint64_t *da = a, *db = b;
    uint64_t *dout = ab;
    const size_t in_bytes = (size_t)batch * n * 8, out_elems = ((size_t)batch - 1) * ab_stride + out_len;
    int rc;
    if (!dev_a) {
        if ((rc = pl->stage_a.ensure(in_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(pl->stage_a.ptr, a, in_bytes, hipMemcpyHostToDevice, cx->stream));
        da = static_cast<const uint64_t *>(pl->stage_a.ptr);
    }
    if (!dev_b) {
        if ((rc = pl->stage_b.ensure(in_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(pl->stage_b.ptr, b, in_bytes, hipMemcpyHostToDevice, cx->stream));
        db = static_cast<const uint64_t *>(pl->stage_b.ptr);
    }
    if (!dev_o) {
        if ((rc = pl->stage_out.ensure(out_elems * 8))) return rc;
        dout = static_cast<uint64_t *>(pl->stage_out.ptr);
        if (ab_stride != out_len)          // keep the caller's bytes between rows: entries >= 2n-1 are not ours
            HIP_TRY(hipMemcpyAsync(dout, ab, out_elems * 8, hipMemcpyHostToDevice, cx->stream));
    }
    if ((rc = multiply_device(pl, da, db, dout, ab_stride, batch))) return rc;
    if (!dev_o) {
        HIP_TRY(hipMemcpyAsync(ab, dout, out_elems * 8, hipMemcpyDeviceToHost, cx->stream));
        HIP_TRY(hipStreamSynchronize(cx->stream));
    } else if (!dev_a || !dev_b) {
        HIP_TRY(hipStreamSynchronize(cx->stream));
    }
    return NTT_OK;
}
