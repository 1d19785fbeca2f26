// row_body.hpp -- device bodies of the 4096-point row transforms (see kernels_row.hip), shared by
// the stand-alone kernels and the fused pipeline kernel (kernels_pipe.hip).
#pragma once
#include "device_ntt.hpp"

namespace ntt {

constexpr int kLdsWords = kRowLen + kRowLen / 16;      // 4352 dwords = 17 KiB

__device__ __forceinline__ uint32_t P1(uint32_t i) { return i + ((i >> 8) << 4); }
__device__ __forceinline__ uint32_t P2(uint32_t i) { return i + (i >> 4); }

// x[j] = row[t + 256*j] on entry; on exit x[k] = spectrum at position 16*t + k.
template <class MM, bool LAZY>
__device__ __forceinline__ void row_forward(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    reg_ntt<MM, LAZY, 4, false, true>(x, a.roots, m);            // loose: 15 of 16 outputs are multiplied next
    x[0] = O::tighten(x[0], m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], a.twA[k * 256 + t], m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[P1(k * 256 + t)] = x[k];
    __syncthreads();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = lds[P1(b * 256 + j * 16 + i)];
    __syncthreads();
    reg_ntt<MM, LAZY, 4, false, true>(x, a.roots, m);
    x[0] = O::tighten(x[0], m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], a.twB[k * 16 + i], m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[P2(b * 256 + k * 16 + i)] = x[k];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = lds[P2(t * 16 + j)];
    __syncthreads();
    reg_ntt<MM, LAZY, 4, false>(x, a.roots, m);
}

// x[k] = spectrum at position 16*t + k on entry; on exit x[j] = row[t + 256*j] (unscaled).
template <class MM, bool LAZY>
__device__ __forceinline__ void row_inverse(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    reg_ntt<MM, LAZY, 4, false, true>(x, a.iroots, m);           // loose values travel through LDS
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[P2(t * 16 + j)] = x[j];
    __syncthreads();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[P2(b * 256 + k * 16 + i)];
    __syncthreads();
    x[0] = O::tighten(x[0], m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], a.itwB[k * 16 + i], m);
    reg_ntt<MM, LAZY, 4, false, true>(x, a.iroots, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[P1(b * 256 + j * 16 + i)] = x[j];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = lds[P1(k * 256 + t)];
    __syncthreads();
    x[0] = O::tighten(x[0], m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], a.itwA[k * 256 + t], m);
    // loose output: the innermost inverse column level multiplies every row (rows >= 1 by the
    // 4-step twiddle, row 0 by the scale constant), so no reduction is needed here
    reg_ntt<MM, LAZY, 4, false, true>(x, a.iroots, m);
}

// Fused multiply of one 4096-point row (forward a, forward b, pointwise, inverse).
template <class MM, bool LAZY>
__device__ __forceinline__ void row_mul_body(const RowArgs<MM> &a, const ModParams &m, uint32_t *lds,
                                             const uint32_t row_index)
{
    using O = Ops<MM, LAZY>;
    const uint32_t t = threadIdx.x;
    const size_t row = (size_t)row_index * kRowLen;
    const uint32_t *__restrict__ pa = a.a + row;
    const uint32_t *__restrict__ pb = a.b + row;
    uint32_t xa[16], xb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = pa[t + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j) xb[j] = pb[t + 256 * j];
    row_forward<MM, LAZY>(xa, lds, t, a, m);
    row_forward<MM, LAZY>(xb, lds, t, a, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY>(xa, lds, t, a, m);
    uint32_t *__restrict__ po = a.out + row;
#pragma unroll
    for (int j = 0; j < 16; ++j) po[t + 256 * j] = xa[j];
}

}  // namespace ntt
