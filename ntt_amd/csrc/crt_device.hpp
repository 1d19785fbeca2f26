// crt_device.hpp -- device functions of the CRT path shared by the stand-alone kernels (kernels_misc.hip) and the
// fused inverse-column-pass + merge kernel (kernels_crt.hip).  Replace CRT_parallel's implicit input reduction
// (main-openmp.cc:426-427) and CRT_combine_parallel (main-openmp.cc:468-485): Garner's mixed-radix form with 32-bit
// Shoup multiplications, then a Horner evaluation mod p64 with 64-bit Montgomery multiplications.
#pragma once
#include "device_ntt.hpp"

namespace ntt {

__device__ __forceinline__ uint32_t red32(uint32_t x, const CrtPrime &c)    // x mod m, any u32 x
{
    return csub(Barrett32::mul_tw(x, c.one, c.mp), c.mp.p);
}

// (acc - v) * inv mod m_j, acc in [0, m_j), v any u32
__device__ __forceinline__ uint32_t garner_step(uint32_t acc, uint32_t v, Barrett32::tw_t inv,
                                                const CrtPrime &c)
{
    const uint32_t d = acc - red32(v, c) + c.mp.p;                  // (0, 2 m_j)
    return csub(Barrett32::mul_tw(d, inv, c.mp), c.mp.p);
}

__device__ __forceinline__ uint64_t mont64_mul(uint64_t a, uint64_t b, const CrtConsts &cc)
{
    // REDC(a*b) with R = 2^64; a*b < p64 * 2^64, p64 < 2^63
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    const uint64_t q = lo * cc.nprime64;
    uint64_t t = hi + __umul64hi(q, cc.p64) + (lo != 0 ? 1u : 0u);   // low words cancel to 0 or 2^64
    return t >= cc.p64 ? t - cc.p64 : t;
}

// (acc - v) * inv mod m_j without reducing v first: `off` is a multiple of m_j that is >= v (CrtConsts::off), and the
// Shoup multiplication takes any u32
__device__ __forceinline__ uint32_t garner_step_off(uint32_t acc, uint32_t v, uint32_t off, Barrett32::tw_t inv, const CrtPrime &c)
{
    return csub(Barrett32::mul_tw(acc - v + off, inv, c.mp), c.mp.p);
}

// One Montgomery step by 2^32 on a value of up to 97 bits held as (w0, w1, hi): returns (value + q * p64) / 2^32 as
// (low word, rest); exact division, q = w0 * (-p64^-1) mod 2^32
__device__ __forceinline__ void redc32_step(uint32_t w0, uint32_t w1, uint64_t hi, const CrtConsts &cc, uint32_t &o0, uint64_t &ohi)
{
    const uint32_t q = w0 * (uint32_t)cc.nprime64;
    const uint64_t t0 = (uint64_t)q * (uint32_t)cc.p64 + w0;                  // low word cancels
    const uint64_t t1 = (uint64_t)q * (uint32_t)(cc.p64 >> 32) + w1 + (t0 >> 32);
    o0 = (uint32_t)t1;
    ohi = hi + (t1 >> 32);
}

// The same value from the mixed-radix digits as ONE 96-bit sum: x = sum_k v_k * (m_0 ... m_(k-1)) with the constants
// premultiplied by 2^64 (c64), then two Montgomery steps by 2^32 and one conditional subtraction -- 12 multiply
// instructions where Horner with three 64-bit Montgomery multiplications takes about 60.  Needs cc.fast.
__device__ __forceinline__ uint64_t crt_merge_fast(uint32_t r0v, uint32_t r1v, uint32_t r2v, uint32_t r3v, const CrtConsts &cc)
{
    const uint32_t v0 = r0v;
    const uint32_t v1 = garner_step_off(r1v, v0, cc.off[0][1], cc.inv[0][1], cc.prime[1]);
    uint32_t v2 = garner_step_off(r2v, v0, cc.off[0][2], cc.inv[0][2], cc.prime[2]);
    v2 = garner_step_off(v2, v1, cc.off[1][2], cc.inv[1][2], cc.prime[2]);
    uint32_t v3 = garner_step_off(r3v, v0, cc.off[0][3], cc.inv[0][3], cc.prime[3]);
    v3 = garner_step_off(v3, v1, cc.off[1][3], cc.inv[1][3], cc.prime[3]);
    v3 = garner_step_off(v3, v2, cc.off[2][3], cc.inv[2][3], cc.prime[3]);
    // S = sum v_k * c64[k] < 4 * 2^31 * 2^63 = 2^96: low halves (v * c_lo < 2^63) in two u64 sums, high halves
    // (v * c_hi < 2^62) in one
    const uint32_t v[4] = {v0, v1, v2, v3};
    uint64_t a01 = 0, a23 = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t lo = (uint64_t)v[k] * (uint32_t)cc.c64[k];
        b += (uint64_t)v[k] * (uint32_t)(cc.c64[k] >> 32);
        if (k < 2) a01 += lo; else a23 += lo;
    }
    const uint64_t lo64 = a01 + a23;
    const uint64_t c1 = lo64 < a01 ? 1u : 0u;                                   // carry into bit 64
    const uint32_t w0 = (uint32_t)lo64;
    const uint64_t mid = (lo64 >> 32) + (uint32_t)b;                            // bits 32.. (33 bits)
    const uint32_t w1 = (uint32_t)mid;
    const uint64_t w2 = (b >> 32) + c1 + (mid >> 32);                           // bits 64..95
    uint32_t t0, u0;
    uint64_t thi, uhi;
    redc32_step(w0, w1, w2, cc, t0, thi);                                       // < 2^64 + 2^63: (t0, thi) with thi of 33 bits
    redc32_step(t0, (uint32_t)thi, thi >> 32, cc, u0, uhi);                     // < p64 + 1.5 * 2^32 < 2 p64
    const uint64_t x = ((uint64_t)uhi << 32) | u0;
    return x >= cc.p64 ? x - cc.p64 : x;
}

__device__ __forceinline__ uint64_t crt_merge_slow(uint32_t r0v, uint32_t r1v, uint32_t r2v, uint32_t r3v, const CrtConsts &cc);

// CRT(r0, r1, r2, r3) mod p64 for canonical residues r_k in [0, m_k): the value of main-openmp.cc:473-484
__device__ __forceinline__ uint64_t crt_merge_value(uint32_t r0v, uint32_t r1v, uint32_t r2v, uint32_t r3v, const CrtConsts &cc)
{
    if (cc.fast) return crt_merge_fast(r0v, r1v, r2v, r3v, cc);                 // wave-uniform
    return crt_merge_slow(r0v, r1v, r2v, r3v, cc);
}

// Horner over 64-bit Montgomery products: any odd p64 < 2^63 (the route of moduli below 2^33)
__device__ __forceinline__ uint64_t crt_merge_slow(uint32_t r0v, uint32_t r1v, uint32_t r2v, uint32_t r3v, const CrtConsts &cc)
{
    // Garner: x = v0 + m0 * (v1 + m1 * (v2 + m2 * v3)), v_j in [0, m_j)
    const uint32_t v0 = r0v;
    const uint32_t v1 = garner_step(r1v, v0, cc.inv[0][1], cc.prime[1]);
    uint32_t v2 = garner_step(r2v, v0, cc.inv[0][2], cc.prime[2]);
    v2 = garner_step(v2, v1, cc.inv[1][2], cc.prime[2]);
    uint32_t v3 = garner_step(r3v, v0, cc.inv[0][3], cc.prime[3]);
    v3 = garner_step(v3, v1, cc.inv[1][3], cc.prime[3]);
    v3 = garner_step(v3, v2, cc.inv[2][3], cc.prime[3]);
    // Horner mod p64; plain value times Montgomery-form constant = plain value
    const uint64_t p = cc.p64;
    auto addmod = [p](uint64_t x, uint32_t v) {
        uint64_t vv = v;
        if (vv >= p) vv %= p;                                       // only when p64 < 2^31
        const uint64_t s = x + vv;
        return s >= p ? s - p : s;
    };
    uint64_t acc = addmod(0, v3);
    acc = addmod(mont64_mul(acc, cc.mR[2], cc), v2);
    acc = addmod(mont64_mul(acc, cc.mR[1], cc), v1);
    acc = addmod(mont64_mul(acc, cc.mR[0], cc), v0);
    return acc;
}

}  // namespace ntt
