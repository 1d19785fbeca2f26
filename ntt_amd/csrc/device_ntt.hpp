// device_ntt.hpp -- register-resident butterflies shared by the gfx950 kernels.
//
// Replaces the inner loop of NTT_iterative (main-montgomery.cc:270-282 / main-barrett.cc:206-217):
// there every butterfly does two modular multiplications (value * twiddle and the running
// twiddle update) and two `%` divisions; here twiddles are constants (SGPR operands or table
// loads), a butterfly is one lazy modular multiplication, one add and one subtract, and a
// thread keeps 2^k points in VGPRs for k stages before anything is written back.
#pragma once
#include "kernels.hpp"

namespace ntt {

template <class MM, bool LAZY>
struct Ops {
    using tw_t = typename MM::tw_t;
    // value range invariant: [0, 2p) when LAZY, [0, p) otherwise
    static __device__ __forceinline__ uint32_t bound(const ModParams &m) { return LAZY ? m.p2 : m.p; }
    static __device__ __forceinline__ uint32_t add(uint32_t u, uint32_t v, const ModParams &m)
    {
        return csub(u + v, bound(m));
    }
    // u - v + bound, in (0, 2*bound): valid as the variable operand of mul_tw without reduction
    static __device__ __forceinline__ uint32_t sub_raw(uint32_t u, uint32_t v, const ModParams &m)
    {
        return u - v + bound(m);
    }
    // (u - v) mod bound-range: u - v wraps above 2^32 - bound when negative, so the unsigned
    // minimum of d and d + bound picks the representative in [0, bound)  (3 instructions; the
    // csub(sub_raw()) form costs 4)
    static __device__ __forceinline__ uint32_t sub(uint32_t u, uint32_t v, const ModParams &m)
    {
#if !defined(NTT_CSUB_MIN)                        // borrow-out + select, as csub() in modarith.hpp
        uint32_t d0, r;
        asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, %4, %0\n\tv_cndmask_b32 %1, %0, %1, vcc"
            : "=&v"(d0), "=&v"(r) : "v"(u), "v"(v), "s"(bound(m)) : "vcc");
        return r;
#else
        const uint32_t d = u - v, e = d + bound(m);
        return e < d ? e : d;
#endif
    }
    static __device__ __forceinline__ uint32_t mul_tw(uint32_t x, tw_t w, const ModParams &m)
    {
        uint32_t r = MM::mul_tw(x, w, m);             // [0, 2p)
        return LAZY ? r : csub(r, m.p);
    }
    // (u - v) * w for the nontrivial butterflies
    static __device__ __forceinline__ uint32_t mul_tw_diff(uint32_t u, uint32_t v, tw_t w, const ModParams &m)
    {
        uint32_t r = MM::mul_tw_diff(u, v, w, bound(m), m);
        return LAZY ? r : csub(r, m.p);
    }
    // the same for a wave-uniform twiddle (Mont32::mul_tw_diff_s)
    static __device__ __forceinline__ uint32_t mul_tw_diff_s(uint32_t u, uint32_t v, tw_t w, const ModParams &m)
    {
        uint32_t r = MM::mul_tw_diff_s(u, v, w, bound(m), m);
        return LAZY ? r : csub(r, m.p);
    }
    static __device__ __forceinline__ uint32_t canon(uint32_t x, const ModParams &m)
    {
        return LAZY ? csub(x, m.p) : x;
    }
    // loose [0, 2*bound) -> [0, bound)
    static __device__ __forceinline__ uint32_t tighten(uint32_t x, const ModParams &m)
    {
        return csub(x, bound(m));
    }
    // variable * variable (pointwise product).  Montgomery: result carries a factor R^-1 that
    // the host folds into the inverse scale constant.
    static __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b, const ModParams &m)
    {
        if constexpr (MM::kMontgomery) {
            uint32_t r = MM::mul_raw(a, b, m);        // a*b < 4p^2 <= 2^32 p when LAZY (4p <= 2^32)
            return LAZY ? r : csub(r, m.p);
        } else {
            return MM::mul_raw(canon(a, m), canon(b, m), m);
        }
    }
};

// (u - v) brought back into range by adding `bound` when it is negative: u in [0, bu), v in [0, bound) -> [0, max(bu, bound)).
// Ops::sub with the bound as an argument (borrow-out + select, as csub() in modarith.hpp).
__device__ __forceinline__ uint32_t sub_mod(uint32_t u, uint32_t v, uint32_t bound)
{
#if !defined(NTT_CSUB_MIN)
    uint32_t d0, r;
    asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, %4, %0\n\tv_cndmask_b32 %1, %0, %1, vcc"
        : "=&v"(d0), "=&v"(r) : "v"(u), "v"(v), "s"(bound) : "vcc");
    return r;
#else
    const uint32_t d = u - v, e = d + bound;
    return u < v ? e : d;
#endif
}

// ---- buffer addressing ---------------------------------------------------------------------------
// The column kernels read and write rows that are S elements apart.  Through a buffer descriptor
// (base + size in four SGPRs) a row is reached with a 32-bit per-lane byte offset plus a scalar
// offset, so an access costs at most one 32-bit add instead of a 64-bit address computation, and the
// hardware range check (offset + 4 > size: loads return 0, stores are dropped) replaces the
// explicit index guards that implement zero padding and the 2n-1 output cut.  Anything the range
// check must see goes into the per-lane offset; the scalar offset is used only where every access
// is in range by construction.
using bufrsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ bufrsrc_t make_buf(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_load(bufrsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store(uint32_t v, bufrsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, 0);
}
// streaming variants (cache policy NT): the caller's operands are read once and its result is never read back,
// so they need not displace the intermediates that the next pass re-reads from L2 / Infinity Cache
#ifndef NTT_NT_AUX
#define NTT_NT_AUX 2
#endif
__device__ __forceinline__ uint32_t buf_load_nt(bufrsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, NTT_NT_AUX);
}
__device__ __forceinline__ void buf_store_nt(uint32_t v, bufrsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, NTT_NT_AUX);
}
// twiddle table entry (4 B Montgomery, 8 B Barrett)
template <class TW>
__device__ __forceinline__ TW buf_load_tw(bufrsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (sizeof(TW) == 4) {
        return TW(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
    } else {
        static_assert(sizeof(TW) == 8, "twiddle entries are 4 or 8 bytes");
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
        TW t;
        t.w = v[0];
        t.wq = v[1];
        return t;
    }
}
template <class TW>
constexpr int tw_shift() { return sizeof(TW) == 4 ? 2 : 3; }

__host__ __device__ constexpr int bitrev_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// Range bookkeeping of the WIDE transform below, at compile time.  Values are tracked as multiples of p: a product is
// below 2p, the sum of two values below 2p is kept unreduced (below 4p), a sum that could reach 8p is reduced once by
// 4p; csub[s][i] says whether the sum written to position i in stage s needs that reduction, subb[s][i] which multiple
// of p the twiddle-free difference written to position i adds back when it is negative.
template <int LOGN, bool HALF>
struct WidePlan {
    static constexpr int N = 1 << LOGN;
    static constexpr int ST = LOGN > 0 ? LOGN : 1;
    bool csub[ST][N];
    int subb[ST][N];
    constexpr WidePlan() : csub{}, subb{}
    {
        int b[N] = {};
        for (int i = 0; i < N; ++i) b[i] = 2;
        for (int s = 0; s < LOGN; ++s) {
            const int half = N >> (s + 1);
            for (int blk = 0; blk < N; blk += 2 * half)
                for (int j = 0; j < half; ++j) {
                    const int lo = blk + j, hi = lo + half;
                    if (HALF && s == 0) {                     // v == 0: the sum is u, the difference u or u * w
                        b[hi] = j == 0 ? b[lo] : 2;
                        continue;
                    }
                    if (s == LOGN - 1) continue;              // last stage: loose outputs below 8p
                    const int bu = b[lo], bv = b[hi];
                    if (bu + bv > 4) { csub[s][lo] = true; b[lo] = 4; }
                    else b[lo] = bu + bv;
                    if (j == 0) { subb[s][hi] = bv; b[hi] = bu > bv ? bu : bv; }
                    else b[hi] = 2;
                }
        }
    }
};

// Which stages of reg_ntt run as radix-4 steps (two stages, one reduction per output: see there), at compile time:
// kind[s] = 1: stages s and s + 1 together; 2: the second stage of such a step; 0: a radix-2 stage.  Never the HALF opening
// stage, never a pair whose second stage is a loose closing stage (its radix-2 form is two instructions per butterfly).
#ifndef NTT_ACC4
#define NTT_ACC4 1                       // 2: the sum side through the accumulator as well (measured -0.7 %, see reg_ntt)
#endif
// a + b + c as ONE instruction (v_add3_u32; the compiler splits the C++ form into two adds when c is a scalar)
__device__ __forceinline__ int32_t add3_i32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return (int32_t)r;
}
template <class MM, bool WIDE>
constexpr bool acc4_enabled() { return NTT_ACC4 && MM::kMontgomery && !WIDE; }
template <int LOGN, bool HALF, bool LOOSE>
struct Acc4Plan {
    int kind[LOGN > 0 ? LOGN : 1];
    constexpr Acc4Plan() : kind{}
    {
        int s = 0;
        while (s < LOGN) {
            if (!(HALF && s == 0) && s + 1 < LOGN && !(LOOSE && s + 1 == LOGN - 1)) {
                kind[s] = 1;
                kind[s + 1] = 2;
                s += 2;
            } else {
                kind[s] = 0;
                s += 1;
            }
        }
    }
};

// Stages s and s + 1 of reg_ntt as one radix-4 step (see there).
template <class MM, bool LAZY, int LOGN>
__device__ __forceinline__ void acc4_step(uint32_t (&x)[1 << LOGN], const Roots64<MM> &roots, const ModParams &m, const int s)
{
    using O = Ops<MM, LAZY>;
    constexpr int N = 1 << LOGN;
    const int half = N >> (s + 1);
    const int step = 32 / half;
    const int hq = half / 2;
    auto W = [&](int k) -> int32_t {
        k &= 63;
        const int32_t c = MM::center_tw(roots.w[k & 31], m);
        return (k & 32) ? -c : c;
    };
#pragma unroll
    for (int blk = 0; blk < N; blk += 2 * half) {
#pragma unroll
        for (int j = 0; j < hq; ++j) {
            const int ia = blk + j, ib = ia + hq, ic = ia + half, id = ic + hq;
            const uint32_t a = x[ia], b = x[ib], c = x[ic], d = x[id];
            const int32_t d0 = (int32_t)(a - c), d1 = (int32_t)(b - d);
#if NTT_ACC4 >= 2
            // the sum side through the accumulator as well: e = a + c - bound and f = b + d - bound lie in
            // [-bound, bound) like the differences (one three-operand add each), and
            // x[i] = e * 1 + f * 1, x[i + half / 2] = e w^(2j) - f w^(2j): 10 instructions instead of 13, but 10
            // half-rate ones instead of 3 -- row kernel 1897 -> 1761 instructions, 570 -> 566 k products/s
            // (profiles/ab_acc4b_r03.txt): not the default
            const int32_t e = add3_i32(a, c, 0u - O::bound(m)), f = add3_i32(b, d, 0u - O::bound(m));
            const uint32_t y0 = MM::mul2_acc(e, W(0), f, W(0), m);
            const uint32_t y1 = MM::mul2_acc(e, W(2 * j * step), f, W(2 * j * step + 32), m);
            x[ia] = LAZY ? y0 : csub(y0, m.p);
            x[ib] = LAZY ? y1 : csub(y1, m.p);
#else
            const uint32_t s0 = O::add(a, c, m), s1 = O::add(b, d, m);
            x[ia] = O::add(s0, s1, m);
            x[ib] = j == 0 ? O::sub(s0, s1, m) : O::mul_tw_diff_s(s0, s1, roots.w[2 * j * step], m);
#endif
            const uint32_t y2 = MM::mul2_acc(d0, W(j * step), d1, W((j + hq) * step), m);
            const uint32_t y3 = MM::mul2_acc(d0, W(3 * j * step), d1, W((3 * j + hq) * step + 32), m);
            x[ic] = LAZY ? y2 : csub(y2, m.p);
            x[id] = LAZY ? y3 : csub(y3, m.p);
        }
    }
}

// A 16-point transform (reg_ntt<4>) TOGETHER with the multiplication of output k by a per-lane twiddle t_k that follows
// it in the row passes (output 0 is not multiplied).  Stages 0 and 1 are the radix-4 step above; stages 2 and 3 are four
// 4-point transforms with trivial roots (1, i), and with a, b, c, d the inputs of one of them its outputs are
//   k         : ((a + c) + (b + d)) t_k            k + 8  : ((a + c) - (b + d)) t_(k+8)
//   k + 4     : (a - c) t_(k+4) + (b - d) (i t_(k+4))        k + 12 : (a - c) t_(k+12) + (b - d) (-i t_(k+12))
// so the two outputs of the difference side are two multiply-adds and one reduction each (Mont32::mul2_acc_v) when the
// table also holds the ROTATED entries i t_k, -i t_k: 24 instructions per four points with their twiddles instead of 29.
// tw(k): the plain table entry (Montgomery form in [0, p)) of output k, asked for k = 1, 2, 3, 8 ... 11;
// t2(q): the pair { centred entry, centred rotated entry } of the q-th output of the difference side, q = 0 ... 7 for
// k = 4, 5, 6, 7, 12, 13, 14, 15 (engine.cpp: rotated_table).  Outputs in natural order, below `bound` (output 0 tight).
// TW0: output 0 is multiplied as well (by tw(0)).
template <class MM, bool LAZY, bool TW0 = false, class TW, class T2>
__device__ __forceinline__ void reg_ntt16_tw(uint32_t (&x)[16], const Roots64<MM> &roots, const ModParams &m,
                                             TW &&tw, T2 &&t2)
{
    using O = Ops<MM, LAZY>;
    static_assert(MM::kMontgomery, "the accumulated products are Montgomery products");
    acc4_step<MM, LAZY, 4>(x, roots, m, 0);
#if NTT_FUSE_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    uint32_t y[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int k = bitrev_c(4 * g, 4);                         // 0, 2, 1, 3
        const uint32_t a = x[4 * g], b = x[4 * g + 1], c = x[4 * g + 2], d = x[4 * g + 3];
        const uint32_t s0 = O::add(a, c, m), s1 = O::add(b, d, m);
        const int32_t d0 = (int32_t)(a - c), d1 = (int32_t)(b - d);
        y[k] = (k == 0 && !TW0) ? O::add(s0, s1, m) : O::mul_tw(s0 + s1, tw(k), m);
        y[k + 8] = O::mul_tw_diff(s0, s1, tw(k + 8), m);
        const uint2 ta = t2(k), tb = t2(k + 4);
        const uint32_t y4 = MM::mul2_acc_v(d0, (int32_t)ta.x, d1, (int32_t)ta.y, m);
        const uint32_t y12 = MM::mul2_acc_v(d0, (int32_t)tb.x, d1, (int32_t)tb.y, m);
        y[k + 4] = LAZY ? y4 : csub(y4, m.p);
        y[k + 12] = LAZY ? y12 : csub(y12, m.p);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

// In-register N-point transform, N = 2^LOGN <= 64: natural order in, natural order out,
// X[k] = sum_j x[j] * w^(jk) with w = roots.w[32/(N/2)] (i.e. the primitive N-th root whose
// powers are the stride-(64/N) entries of the w_64 table; forward or inverse table).
// Decimation in frequency; the final bit reversal is a compile-time register renaming.
// HALF: x[j] == 0 for j >= N/2 (zero-padded operand), the first stage then needs no add/sub.
// LOOSE: the last stage skips its conditional subtractions, so outputs lie in [0, 2*bound)
// ([0, 4p) lazy, [0, 2p) strict).  Every consumer either multiplies the value by a twiddle
// (mul_tw takes any u32 and reduces for free) or applies Ops::tighten -- this removes 16 of the
// ~47 conditional subtractions of a 16-point transform.
// WIDE (needs LAZY, LOOSE and 8p <= 2^32, i.e. p < 2^29 -- the reference's three shipped primes 7340033, 104857601,
// 469762049, main-montgomery.cc:301-306, and the CRT prime 469762049, main-openmp.cc:654): inputs in [0, 2p), values
// inside the transform may grow to 4p, a sum is reduced only where it could reach 8p (WidePlan), outputs lie in
// [0, 8p).  16 of the 24 conditional subtractions a 16-point LOOSE transform still has are gone (2 of 7 instructions
// of those butterflies).  The signed multiply-add butterfly stays valid: |u - v| < 4p < 2^31 and |u - v| w < 4p^2 <= p 2^31.
template <class MM, bool LAZY, int LOGN, bool HALF, bool LOOSE = false, bool WIDE = false>
__device__ __forceinline__ void reg_ntt(uint32_t (&x)[1 << LOGN], const Roots64<MM> &roots,
                                        const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    constexpr int N = 1 << LOGN;
    static_assert(!WIDE || (LAZY && LOOSE), "the wide-range transform is a lazy transform with loose outputs");
    if constexpr (LOGN == 0) return;
    constexpr WidePlan<WIDE ? LOGN : 0, HALF> wp{};
    constexpr Acc4Plan<acc4_enabled<MM, WIDE>() ? LOGN : 0, HALF, LOOSE> ap{};
    const uint32_t p4 = m.p2 << 1;                    // used by WIDE only
#pragma unroll
    for (int s = 0; s < LOGN; ++s) {
        const int half = N >> (s + 1);
        const int step = 32 / half;                   // w_(2*half)^j = w_64^(j*step)
        if constexpr (acc4_enabled<MM, WIDE>()) {
            // Stages s and s + 1 as ONE radix-4 step (Montgomery kernels; not the HALF opening stage, and not when stage
            // s + 1 is a loose closing stage, whose radix-2 form is two instructions per butterfly already).  With
            // a, b, c, d = x[i], x[i + half / 2], x[i + half], x[i + 3 half / 2] the two stages compute
            //   x[i]              = (a + c) + (b + d)
            //   x[i + half / 2]   = ((a + c) - (b + d)) w^(2j)
            //   x[i + half]       = (a - c) w^j  + (b - d) w^(j + half / 2)
            //   x[i + 3 half / 2] = (a - c) w^(3j) - (b - d) w^(3j + half / 2)          (w = w_(2 half), j < half / 2)
            // and each of the last two is two multiply-adds into one accumulator and ONE reduction (Mont32::mul2_acc)
            // instead of a reduced product, a modular add / subtract and another reduced product: 23 instructions
            // per four points instead of 28 (22 instead of 25 for j = 0).  Twiddles are the centred forms of the w_64
            // table, computed on the scalar unit; an index of 32 and above is the negated entry.
            if (ap.kind[acc4_enabled<MM, WIDE>() ? s : 0] == 2) continue;             // second stage of a radix-4 step: done
            if (ap.kind[acc4_enabled<MM, WIDE>() ? s : 0] == 1) {
                acc4_step<MM, LAZY, LOGN>(x, roots, m, s);
                continue;
            }
        }
#pragma unroll
        for (int blk = 0; blk < N; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                uint32_t u = x[blk + j], v = x[blk + j + half];
                if (HALF && s == 0) {
                    // v == 0: sum = u, difference = u
                    x[blk + j + half] = (j == 0) ? u : O::mul_tw(u, roots.w[j * step], m);
                } else if (WIDE && s == LOGN - 1) {
                    x[blk + j] = u + v;                              // below 8p
                    x[blk + j + half] = u - v + p4;                  // (0, 8p)
                } else if (WIDE) {
                    const uint32_t sum = u + v;
                    x[blk + j] = wp.csub[WIDE ? s : 0][WIDE ? blk + j : 0] ? csub(sum, p4) : sum;
                    if (j == 0) x[blk + j + half] = sub_mod(u, v, wp.subb[WIDE ? s : 0][WIDE ? blk + j + half : 0] == 4 ? p4 : m.p2);
                    else x[blk + j + half] = MM::mul_tw_diff(u, v, roots.w[j * step], p4, m);   // [0, 2p)
                } else if (LOOSE && s == LOGN - 1) {
                    x[blk + j] = u + v;                              // half == 1: j == 0 only
                    x[blk + j + half] = O::sub_raw(u, v, m);
                } else {
                    x[blk + j] = O::add(u, v, m);
                    if (j == 0) x[blk + j + half] = O::sub(u, v, m);
                    else x[blk + j + half] = O::mul_tw_diff_s(u, v, roots.w[j * step], m); // signed-mad form
                }
            }
        }
    }
    uint32_t y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[bitrev_c(i, LOGN)] = x[i];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = y[i];
}

// loose output of a WIDE transform, [0, 8p), to [0, 2p)
__device__ __forceinline__ uint32_t tighten_wide(uint32_t x, const ModParams &m)
{
    return csub(csub(x, m.p2 << 1), m.p2);
}

}  // namespace ntt
