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

// CRT(r0, r1, r2, r3) mod p64 for canonical residues r_k in [0, m_k): the value of main-openmp.cc:473-484
__device__ __forceinline__ uint64_t crt_merge_value(uint32_t r0v, uint32_t r1v, uint32_t r2v, uint32_t r3v, const CrtConsts &cc)
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
