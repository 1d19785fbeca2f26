// modarith.hpp -- 32-bit modular arithmetic for the NTT engine, host and gfx950 device.
//
// Two interchangeable multiplication policies, both exact in Z_p with canonical results:
//   Mont32    : Montgomery REDC with R = 2^32.  The reference's class montgomery
//               (main-montgomery.cc:133-234, REDC :170-181) uses 64-bit lanes and __uint128_t;
//               here a residue is one u32 lane and REDC is mul_lo + two v_mad_u64_u32.
//   Barrett32 : the reference's class Barrett (main-barrett.cc:130-172) computes
//               q = (x * floor(2^64/p)) >> 64 with a 128-bit product.  Here a multiplication
//               by a precomputed constant w carries its own quotient w' = floor(w * 2^32 / p)
//               (one mul_hi + two mul_lo, no 128-bit product), and variable * variable uses
//               the two-word Barrett reduction with mu = floor(2^64 / p).
//
// Measured on MI355X (tools/ubench_intmul.hip): v_mul_lo_u32, v_mul_hi_u32 and v_mad_u64_u32
// all issue at the same rate as v_min_u32 / v_lshlrev_b32 (half the rate of v_add_u32), so a
// modular multiplication costs three half-rate instructions in either policy.
//
// Lazy ranges: mul_tw() accepts ANY u32 as the variable operand and returns a value in [0, 2p).
// Butterflies keep values in [0, 2p) when 4p <= 2^32 (Lazy = true) and in [0, p) otherwise.
#pragma once
#include <cstdint>
#include <initializer_list>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTT_HD __host__ __device__ __forceinline__
#else
#define NTT_HD inline
#endif

namespace ntt {

// ---- host-side number theory (plan creation only) ------------------------------------------
inline uint32_t h_mulmod(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)((uint64_t)a * b % p); }
inline uint32_t h_powmod(uint32_t b, uint64_t e, uint32_t p)
{
    uint32_t r = 1 % p;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod(r, b, p);
        b = h_mulmod(b, b, p);
    }
    return r;
}
inline uint64_t h_mulmod64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((unsigned __int128)a * b % p); }
inline uint64_t h_powmod64(uint64_t b, uint64_t e, uint64_t p)
{
    uint64_t r = 1 % p;
    b %= p;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod64(r, b, p);
        b = h_mulmod64(b, b, p);
    }
    return r;
}
inline bool h_is_prime(uint32_t n)
{
    if (n < 2) return false;
    for (uint32_t q : {2u, 3u, 5u, 7u, 11u, 13u, 17u, 19u, 23u, 29u, 31u, 37u}) {
        if (n % q == 0) return n == q;
    }
    uint32_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    for (uint32_t a : {2u, 7u, 61u}) {               // deterministic Miller-Rabin for n < 2^32
        uint32_t x = h_powmod(a % n, d, n);
        if (x == 1 || x == n - 1 || a % n == 0) continue;
        bool comp = true;
        for (int i = 1; i < s && comp; ++i) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) comp = false;
        }
        if (comp) return false;
    }
    return true;
}
// Smallest primitive root of the prime p (3 for every prime the reference uses).
inline uint32_t h_primitive_root(uint32_t p)
{
    if (p == 2) return 1;
    uint32_t phi = p - 1, m = phi, fac[32];
    int nf = 0;
    for (uint32_t q = 2; (uint64_t)q * q <= m; ++q) {
        if (m % q == 0) {
            fac[nf++] = q;
            while (m % q == 0) m /= q;
        }
    }
    if (m > 1) fac[nf++] = m;
    for (uint32_t g = 2; g < p; ++g) {
        bool ok = true;
        for (int i = 0; i < nf && ok; ++i) ok = h_powmod(g, phi / fac[i], p) != 1;
        if (ok) return g;
    }
    return 0;
}

// ---- per-prime constants passed by value to kernels ---------------------------------------
struct ModParams {
    uint32_t p;        // modulus, odd prime < 2^31
    uint32_t p2;       // 2p
    uint32_t nprime;   // -p^-1 mod 2^32            (Montgomery)
    uint32_t r1;       // R mod p  = Montgomery form of 1
    uint32_t r2;       // R^2 mod p
    uint32_t kbits;    // k = bit length of p       (Barrett)
    uint32_t mu;       // floor(2^(2k) / p) < 2^(k+1) <= 2^32
};

inline ModParams make_mod_params(uint32_t p)
{
    ModParams m{};
    m.p = p;
    m.p2 = 2 * p;
    uint32_t inv = 1;                                 // Newton iteration for p^-1 mod 2^32
    for (int i = 0; i < 5; ++i) inv *= 2u - p * inv;
    m.nprime = 0u - inv;
    uint64_t R = (1ULL << 32) % p;
    m.r1 = (uint32_t)R;
    m.r2 = (uint32_t)(R * R % p);
    uint32_t k = 0;
    while ((1ULL << k) <= p) ++k;
    m.kbits = k;
    m.mu = (uint32_t)((((unsigned __int128)1) << (2 * k)) / p);
    return m;
}

NTT_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// x in [0, 2m) -> [0, m).  On the device: subtract with borrow-out + select, two full-rate instructions
// (2 + 2 issue cycles per wave64) instead of v_sub_u32 + the half-rate v_min_u32 (2 + 4).  Written as one asm
// block: left to the compiler the C++ form of it becomes v_cmp (half rate) + v_cndmask or picks up s_nop
// hazard padding, which is how round 1 measured this idea 3 % slower; as a pinned pair it is 2 % faster end to
// end (profiles/ab_csub_r02.txt) and 9 % fewer VALU issue cycles in the row kernel.  -DNTT_CSUB_MIN restores
// the subtract + minimum form.
NTT_HD uint32_t csub(uint32_t x, uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTT_CSUB_MIN)
    uint32_t t, r;
    asm("v_subrev_co_u32 %0, vcc, %2, %3\n\tv_cndmask_b32 %1, %0, %3, vcc" : "=&v"(t), "=v"(r) : "s"(m), "v"(x) : "vcc");
    return r;
#else
    uint32_t t = x - m;
    return t < x ? t : x;       // unsigned min(x, x - m): x - m wraps high when x < m
#endif
}

// ---- Montgomery, R = 2^32 ------------------------------------------------------------------
struct Mont32 {
    static constexpr bool kMontgomery = true;
    // Table entries are Montgomery forms w*R mod p, one u32 each.
    using tw_t = uint32_t;
    // REDC of a 64-bit T < 2^32 * p: returns T * R^-1 mod p in [0, 2p).
    static NTT_HD uint32_t redc_lazy(uint64_t T, const ModParams &m)
    {
        uint32_t q = (uint32_t)T * m.nprime;
        uint64_t t = T + (uint64_t)q * m.p;           // low word cancels; fits: T, q*p < 2^32 * p
        return (uint32_t)(t >> 32);
    }
    // x: any u32, w: table entry (w*R mod p) -> x*w mod p in [0, 2p)
    static NTT_HD uint32_t mul_tw(uint32_t x, tw_t w, const ModParams &m)
    {
        return redc_lazy((uint64_t)x * w, m);
    }
    // (u - v) * w mod p in [0, 2p) for u, v in [0, bound), bound <= min(2p, 2^31): the signed difference goes
    // straight into a signed v_mad_i64_i32 whose 64-bit addend p * 2^31 makes the sum non-negative (|u - v| * w <
    // bound * p <= p * 2^31) without changing it modulo p, and keeps it below p * 2^32, so redc_lazy applies.  This
    // saves the separate v_add_u32 of the sub_raw() form, and the addend is ONE wave-uniform constant for every
    // twiddle (round 1 used bound * w: an SGPR pair per twiddle, which kept the 32- and 64-point register
    // transforms on the longer form).
    static NTT_HD uint32_t mul_tw_diff(uint32_t u, uint32_t v, tw_t w, uint32_t bound, const ModParams &m)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTT_NO_MADI64)
        // hipcc expands a signed 32x32+64 C++ expression into five instructions; ask for the one
        (void)bound;
        const uint64_t C = (uint64_t)m.p << 31;
        uint64_t T, carry;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(carry) : "v"(u - v), "v"(w), "s"(C));
        return redc_lazy(T, m);
#else
        return mul_tw(u - v + bound, w, m);
#endif
    }
    // d0 * w0 + d1 * w1 mod p in [0, 2p) with ONE reduction: both signed products go into the same 64-bit accumulator.
    // d0, d1: signed differences of two values in [0, bound), bound <= min(2p, 2^31); w0, w1: CENTRED table entries,
    // |w| <= p / 2 (center_tw).  |d0 w0 + d1 w1| < 2 * 2p * p / 2 = 2p^2 <= p * 2^31 (p < 2^30 lazy; strict: < p^2 < p * 2^31),
    // so with the addend p * 2^31 the sum lies in (0, 2p^2 + p * 2^31) below p * 2^32 and redc_lazy applies.  This is what
    // lets two radix-2 stages run as one radix-4 step: the outputs (a - c) w^j +- i (b - d) w^j of the difference side
    // cost two multiply-adds and one reduction each instead of a reduced product, a modular add and another product.
    // The twiddles are scalar operands, the addend a vector pair (one scalar source per VOP3 on gfx950).
    static NTT_HD uint32_t mul2_acc(int32_t d0, int32_t w0, int32_t d1, int32_t w1, const ModParams &m)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t C = (uint64_t)m.p << 31;
        uint64_t T0, T1, carry;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T0), "=s"(carry) : "v"(d0), "s"(w0), "v"(C));
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T1), "=s"(carry) : "v"(d1), "s"(w1), "v"(T0));
        return redc_lazy(T1, m);
#else
        const int64_t T = (int64_t)d0 * w0 + (int64_t)d1 * w1 + (int64_t)((uint64_t)m.p << 31);
        return redc_lazy((uint64_t)T, m);
#endif
    }
    // the same with per-lane table entries (vector operands; the addend of the first multiply-add is then the scalar pair)
    static NTT_HD uint32_t mul2_acc_v(int32_t d0, int32_t w0, int32_t d1, int32_t w1, const ModParams &m)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t C = (uint64_t)m.p << 31;
        uint64_t T0, T1, carry;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T0), "=s"(carry) : "v"(d0), "v"(w0), "s"(C));
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T1), "=s"(carry) : "v"(d1), "v"(w1), "v"(T0));
        return redc_lazy(T1, m);
#else
        return mul2_acc(d0, w0, d1, w1, m);
#endif
    }
    // table entry w * R mod p in [0, p) -> the representative in (-p / 2, p / 2]
    static NTT_HD int32_t center_tw(tw_t w, const ModParams &m)
    {
        return w > (m.p >> 1) ? (int32_t)(w - m.p) : (int32_t)w;
    }
    // mul_tw_diff for a WAVE-UNIFORM twiddle (the w_64 powers of the register transforms): the twiddle is the scalar
    // operand and the addend a vector pair, so no v_mov copies the twiddle into a vector register first (a VOP3 takes one
    // scalar source; the forward column kernel spent 80 of its 1420 instructions per column on such copies)
    static NTT_HD uint32_t mul_tw_diff_s(uint32_t u, uint32_t v, tw_t w, uint32_t bound, const ModParams &m)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTT_NO_MADI64)
        (void)bound;
        const uint64_t C = (uint64_t)m.p << 31;
        uint64_t T, carry;
        asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(carry) : "v"(u - v), "s"(w), "v"(C));
        return redc_lazy(T, m);
#else
        return mul_tw(u - v + bound, w, m);
#endif
    }
    // a*b*R^-1 mod p in [0, 2p); a any u32, b < p (or a*b < 2^32 * p)
    static NTT_HD uint32_t mul_raw(uint32_t a, uint32_t b, const ModParams &m)
    {
        return redc_lazy((uint64_t)a * b, m);
    }
    static inline tw_t make_tw(uint32_t w, const ModParams &m)      // host: w -> w*R mod p
    {
        return (uint32_t)(((uint64_t)w << 32) % m.p);
    }
};

// ---- Barrett with per-constant quotients (Shoup form) ------------------------------------------
struct Barrett32 {
    static constexpr bool kMontgomery = false;
    struct tw_t { uint32_t w, wq; };                  // w and floor(w * 2^32 / p)
    static NTT_HD uint32_t mul_tw(uint32_t x, tw_t t, const ModParams &m)
    {
        uint32_t q = mulhi32(x, t.wq);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(NTT_NO_MADI64)
        // x*w - q*p mod 2^32 as one multiply-add: q * (2^32 - p) + low(x*w) is the low word of v_mad_u64_u32
        // (hipcc folds the C++ form back into multiply + subtract; ask for the one instruction)
        // (the addend is the full 64-bit x*w: its high word does not reach the low word of the result, and
        // taking it along saves the zero-extension of a 32-bit addend)
        uint64_t T, carry;
        const uint64_t addend = (uint64_t)x * t.w;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(T), "=s"(carry) : "v"(q), "s"(0u - m.p), "v"(addend));
        return (uint32_t)T;
#else
        return x * t.w - q * m.p;                      // exact mod 2^32, value in [0, 2p)
#endif
    }
    static NTT_HD uint32_t mul_tw_diff(uint32_t u, uint32_t v, tw_t t, uint32_t bound, const ModParams &m)
    {
        return mul_tw(u - v + bound, t, m);            // the quotient estimate needs the non-negative value
    }
    static NTT_HD uint32_t mul_tw_diff_s(uint32_t u, uint32_t v, tw_t t, uint32_t bound, const ModParams &m)
    {
        return mul_tw_diff(u, v, t, bound, m);
    }
    // Classic Barrett (HAC 14.42, base 2) for T = a*b with a, b < p < 2^k:
    // q = ((T >> (k-1)) * mu) >> (k+1) underestimates floor(T/p) by at most 2, so
    // r = T - q*p lies in [0, 3p); two conditional subtractions give the canonical residue.
    // This is main-barrett.cc:145-150 with the 128-bit product replaced by 32-bit words.
    static NTT_HD uint32_t mul_raw(uint32_t a, uint32_t b, const ModParams &m)
    {
        uint64_t T = (uint64_t)a * b;
        uint32_t q1 = (uint32_t)(T >> (m.kbits - 1));
        uint32_t q3 = (uint32_t)(((uint64_t)q1 * m.mu) >> (m.kbits + 1));
        uint64_t r = T - (uint64_t)q3 * m.p;
        if (r >= m.p2) r -= m.p2;
        if (r >= m.p) r -= m.p;
        return (uint32_t)r;                            // canonical [0, p)
    }
    static inline tw_t make_tw(uint32_t w, const ModParams &m)
    {
        return tw_t{w, (uint32_t)(((uint64_t)w << 32) / m.p)};
    }
};

}  // namespace ntt
