// row_body.hpp -- device bodies of the 4096-point row transforms (see kernels_row.hip), shared by
// the stand-alone kernels and the one-kernel products (kernels_fused.hip).
#pragma once
#include "device_ntt.hpp"

// Ablation switches for timing experiments (tools/ablate_row.sh); never defined in the shipped build:
//   NTT_ABL_NODATA  row kernels take synthetic inputs instead of global loads and skip the store
//   NTT_ABL_NOTW    4-step/row twiddles come from the SGPR root table instead of global loads
//   NTT_ABL_NOLDS   the LDS exchanges (and their barriers) are skipped
//   NTT_ABL_VEC4    16-byte global accesses from the wrong positions (what a permuted hand-off layout would allow):
//                   +0.5 % end to end (round 2), so the dword accesses are not what the row kernel waits for
#if defined(NTT_ABL_NOTW)
#define NTT_TW(tab, idx) a.roots.w[(idx) & 31]
#else
#define NTT_TW(tab, idx) (tab)[idx]
#endif

#if defined(NTT_ABL_NOLDS)
#define NTT_LDS_ST(addr, v) asm volatile("" ::"v"(v))
#define NTT_LDS_LD(dst, addr) asm volatile("" : "+v"(dst))
#define NTT_BARRIER()
#else
#define NTT_LDS_ST(addr, v) lds[addr] = (v)
#define NTT_LDS_LD(dst, addr) (dst) = lds[addr]
#define NTT_BARRIER() __syncthreads()
#endif

namespace ntt {

constexpr int kLdsWords = kRowLen + kRowLen / 16;      // 4352 dwords = 17 KiB

__device__ __forceinline__ uint32_t P1(uint32_t i) { return i + ((i >> 8) << 4); }
__device__ __forceinline__ uint32_t P2(uint32_t i) { return i + (i >> 4); }

// WIDE (p < 2^29, device_ntt.hpp): the loose 16-point transforms keep sums unreduced up to 4p and hand out values below
// 8p; what is not multiplied next is brought back below 2p (tighten_wide).  Inputs are below 2p as always.
template <bool WIDE>
__device__ __forceinline__ uint32_t row_tighten(uint32_t x, const ModParams &m, uint32_t bound)
{
    return WIDE ? tighten_wide(x, m) : csub(x, bound);
}

// x[j] = row[t + 256*j] on entry; on exit x[k] = spectrum at position 16*t + k.
template <class MM, bool LAZY, bool WIDE = false>
__device__ __forceinline__ void row_forward(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.roots, m);      // loose: 15 of 16 outputs are multiplied next
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], NTT_TW(a.twA, k * 256 + t), m);
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_ST(P1(k * 256 + t), x[k]);
    NTT_BARRIER();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], P1(b * 256 + j * 16 + i));
    NTT_BARRIER();
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.roots, m);
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], NTT_TW(a.twB, k * 16 + i), m);
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_ST(P2(b * 256 + k * 16 + i), x[k]);
    NTT_BARRIER();
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], P2(t * 16 + j));
    NTT_BARRIER();
    reg_ntt<MM, LAZY, 4, false>(x, a.roots, m);
}

// x[k] = spectrum at position 16*t + k on entry; on exit x[j] = row[t + 256*j] (unscaled).
template <class MM, bool LAZY, bool WIDE = false>
__device__ __forceinline__ void row_inverse(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);     // loose values travel through LDS
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_ST(P2(t * 16 + j), x[j]);
    NTT_BARRIER();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], P2(b * 256 + k * 16 + i));
    NTT_BARRIER();
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], NTT_TW(a.itwB, k * 16 + i), m);
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_ST(P1(b * 256 + j * 16 + i), x[j]);
    NTT_BARRIER();
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], P1(k * 256 + t));
    NTT_BARRIER();
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], NTT_TW(a.itwA, k * 256 + t), m);
    // loose output: the innermost inverse column level multiplies every row (rows >= 1 by the
    // 4-step twiddle, row 0 by the scale constant), so no reduction is needed here (WIDE: below 8p, any u32 is fine there)
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);
}

// Fused multiply of one 4096-point row (forward a, forward b, pointwise, inverse).
template <class MM, bool LAZY, bool WIDE = false>
__device__ __forceinline__ void row_mul_body(const RowArgs<MM> &a, const ModParams &m, uint32_t *lds,
                                             const uint32_t row_index)
{
    using O = Ops<MM, LAZY>;
    const uint32_t t = threadIdx.x;
    const size_t row = (size_t)row_index * kRowLen;
    const uint32_t *__restrict__ pa = a.a + row;
    const uint32_t *__restrict__ pb = a.b + row;
    uint32_t xa[16], xb[16];
#if defined(NTT_ABL_NODATA)
#pragma unroll
    for (int j = 0; j < 16; ++j) { xa[j] = t + j; xb[j] = t * 3 + j; }
    (void)pa; (void)pb;
#elif defined(NTT_ABL_VEC4)
    // timing experiment only (results are wrong): 16-byte accesses, as a permuted hand-off layout would allow
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 va = reinterpret_cast<const uint4 *>(pa + 1024 * q)[t], vb = reinterpret_cast<const uint4 *>(pb + 1024 * q)[t];
        xa[4 * q] = va.x; xa[4 * q + 1] = va.y; xa[4 * q + 2] = va.z; xa[4 * q + 3] = va.w;
        xb[4 * q] = vb.x; xb[4 * q + 1] = vb.y; xb[4 * q + 2] = vb.z; xb[4 * q + 3] = vb.w;
    }
#else
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = pa[t + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j) xb[j] = pb[t + 256 * j];
#endif
    row_forward<MM, LAZY, WIDE>(xa, lds, t, a, m);
    row_forward<MM, LAZY, WIDE>(xb, lds, t, a, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY, WIDE>(xa, lds, t, a, m);
    uint32_t *__restrict__ po = a.out + row;
#if defined(NTT_ABL_NODATA)
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= xa[j];
    if (acc == 0x9e3779b9u) po[t] = acc;            // keeps the computation live, practically never stores
#elif defined(NTT_ABL_VEC4)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        reinterpret_cast<uint4 *>(po + 1024 * q)[t] = make_uint4(xa[4 * q], xa[4 * q + 1], xa[4 * q + 2], xa[4 * q + 3]);
#else
#pragma unroll
    for (int j = 0; j < 16; ++j) po[t + 256 * j] = xa[j];
#endif
}

}  // namespace ntt
