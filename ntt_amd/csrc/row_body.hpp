// row_body.hpp -- device bodies of the 4096-point row transforms (see kernels_row.hip), shared by
// the stand-alone kernels and the one-kernel products (kernels_fused.hip).
#pragma once
#include <cstring>
#include "device_ntt.hpp"

// Ablation switches for timing experiments (tools/ablate_row.sh); never defined in the shipped build:
//   NTT_ABL_NODATA  row kernels take synthetic inputs instead of global loads and skip the store
//   NTT_ABL_NOTW    row twiddles are not loaded (RowTable::at returns a constant)
//   NTT_ABL_NOLDS   the LDS exchanges (and their barriers) are skipped
//   NTT_ABL_VEC4    16-byte global accesses from the wrong positions (what a permuted hand-off layout would allow):
//                   +0.5 % end to end (round 2), so the dword accesses are not what the row kernel waits for
#if defined(NTT_ABL_NOLDS)
#define NTT_LDS_ST(addr, v) asm volatile("" ::"v"(v))
#define NTT_LDS_LD(dst, addr) asm volatile("" : "+v"(dst))
#define NTT_BARRIER()
#else
#define NTT_LDS_ST(addr, v) lds[addr] = (v)
#define NTT_LDS_LD(dst, addr) (dst) = lds[addr]
#define NTT_BARRIER() __syncthreads()
#endif

#ifndef NTT_ROW_SHARE_TW
#define NTT_ROW_SHARE_TW 3                // bit 0: the fused kernels, bit 1: the others (Barrett, wide range)
#endif
#ifndef NTT_ROW_FUSE_TW
#define NTT_ROW_FUSE_TW 15               // bits 0 / 1: first / second forward pass, 2: the inverse's first, 3: its second (A/B builds)
#endif

namespace ntt {

constexpr int kLdsWords = kRowLen + kRowLen / 16;      // 4352 dwords = 17 KiB

__device__ __forceinline__ uint32_t P1(uint32_t i) { return i + ((i >> 8) << 4); }
__device__ __forceinline__ uint32_t P2(uint32_t i) { return i + (i >> 4); }

// Table reads of the row passes go through buffer descriptors (device_ntt.hpp): entry (k, lane) of a [16][cols] table is
// reached as lane offset + scalar offset k * cols * sizeof(entry), so no table or row access costs a vector address
// computation (the 64-bit pointer forms needed ~35 of them per wave).
template <class TW>
struct RowTable {
    bufrsrc_t r;
    uint32_t lane;                                    // byte offset of this lane's column
    template <int COLS>                               // entry k of this lane: COLS entries further per k
    __device__ __forceinline__ TW at(int k) const
    {
#if defined(NTT_ABL_NOTW)
        TW w{};
        std::memcpy(&w, &lane, sizeof(lane));
        return w;
#else
        return buf_load_tw<TW>(r, lane, (uint32_t)k * COLS * (uint32_t)sizeof(TW));
#endif
    }
};
template <class TW>
__device__ __forceinline__ RowTable<TW> row_table(const TW *base, uint32_t rows, uint32_t cols, uint32_t col)
{
    return RowTable<TW>{make_buf(base, rows * cols * (uint32_t)sizeof(TW)), col * (uint32_t)sizeof(TW)};
}
// the pair tables of the fused passes ([8][cols] uint2)
struct RowTable2 {
    bufrsrc_t r;
    uint32_t lane;
    template <int COLS>
    __device__ __forceinline__ uint2 at(int q) const
    {
#if defined(NTT_ABL_NOTW)
        return make_uint2(lane, 3u * lane);
#else
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane, (int)((uint32_t)q * COLS * 8u), 0);
        return make_uint2(v[0], v[1]);
#endif
    }
};
__device__ __forceinline__ RowTable2 row_table2(const uint2 *base, uint32_t cols, uint32_t col, uint32_t rows = 8)
{
    return RowTable2{make_buf(base, rows * cols * 8u), col * 8u};
}

// WIDE (p < 2^29, device_ntt.hpp): the loose 16-point transforms keep sums unreduced up to 4p and hand out values below
// 8p; what is not multiplied next is brought back below 2p (tighten_wide).  Inputs are below 2p as always.
template <bool WIDE>
__device__ __forceinline__ uint32_t row_tighten(uint32_t x, const ModParams &m, uint32_t bound)
{
    return WIDE ? tighten_wide(x, m) : csub(x, bound);
}

// which row transforms run the fused passes (Montgomery plans outside the wide range; RowArgs then carries the pair tables)
template <class MM, bool WIDE>
constexpr bool row_fuse() { return NTT_ROW_FUSE_TW && acc4_enabled<MM, WIDE>(); }
template <class MM, bool WIDE>
constexpr bool row_fuse_inv() { return (NTT_ROW_FUSE_TW & 4) && acc4_enabled<MM, WIDE>(); }

// One fused pass: the table entries are requested first, as one batch in front of the arithmetic (the scheduling barrier
// keeps them there: left alone, the compiler issued every table load of the kernel directly in front of its use, each
// with a full wait -- 530 k instead of 570 k products/s).
// COLS / COLS2: entries between consecutive outputs in the plain and in the pair table; TW0: output 0 has a twiddle too
template <class MM, bool LAZY, int COLS, int COLS2 = COLS, bool TW0 = false>
__device__ __forceinline__ void fused_pass(uint32_t (&x)[16], const Roots64<MM> &roots, const ModParams &m,
                                           const RowTable<typename MM::tw_t> &tab, const RowTable2 &tab2)
{
    typename MM::tw_t tv[16];
    uint2 t2[8];
#pragma unroll
    for (int k = TW0 ? 0 : 1; k < 16; ++k)
        if ((k & 4) == 0) tv[k] = tab.template at<COLS>(k);
#pragma unroll
    for (int q = 0; q < 8; ++q) t2[q] = tab2.at<COLS2>(q);
    __builtin_amdgcn_sched_barrier(0);
    reg_ntt16_tw<MM, LAZY, TW0>(x, roots, m, [&](int k) { return tv[k]; }, [&](int q) { return t2[q]; });
}

// x[k] *= table entry k (k >= 1; x[0] is only brought into range), the entries requested as one batch
template <class MM, bool LAZY, bool WIDE, int COLS>
__device__ __forceinline__ void twiddle_pass(uint32_t (&x)[16], const ModParams &m, const RowTable<typename MM::tw_t> &tab)
{
    using O = Ops<MM, LAZY>;
    typename MM::tw_t tv[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tv[k] = tab.template at<COLS>(k);
    __builtin_amdgcn_sched_barrier(0);
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], tv[k], m);
}

// a forward pass that is not fused: transform (loose outputs), then the twiddles, their entries requested in front of it
template <class MM, bool LAZY, bool WIDE, int COLS>
__device__ __forceinline__ void plain_pass(uint32_t (&x)[16], const Roots64<MM> &roots, const ModParams &m,
                                           const RowTable<typename MM::tw_t> &tab)
{
    using O = Ops<MM, LAZY>;
    typename MM::tw_t tv[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tv[k] = tab.template at<COLS>(k);
    __builtin_amdgcn_sched_barrier(0);
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, roots, m);          // loose: 15 of 16 outputs are multiplied next
    x[0] = row_tighten<WIDE>(x[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = O::mul_tw(x[k], tv[k], m);
}

// x[j] = row[t + 256*j] on entry; on exit x[k] = spectrum at position 16*t + k.
// FUSE (row_mul_kernel of the Montgomery plans outside the wide range): the first two passes run their last two stages
// together with the twiddles that follow (reg_ntt16_tw, a.twA2 / a.twB2).
template <class MM, bool LAZY, bool WIDE = false, bool FUSE = false>
__device__ __forceinline__ void row_forward(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    static_assert(!FUSE || (MM::kMontgomery && !WIDE), "the fused passes are Montgomery passes outside the wide range");
    using tw_t = typename MM::tw_t;
    const RowTable<tw_t> tA = row_table(a.twA, 16, 256, t);
    if constexpr (FUSE && (NTT_ROW_FUSE_TW & 1)) {
        const RowTable2 tA2 = row_table2(a.twA2, 256, t);
        fused_pass<MM, LAZY, 256>(x, a.roots, m, tA, tA2);
    } else {
        plain_pass<MM, LAZY, WIDE, 256>(x, a.roots, m, tA);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_ST(t + 272 * k, x[k]);                  // P1(k * 256 + t)
    NTT_BARRIER();
    const uint32_t b = t >> 4, i = t & 15;
    const uint32_t e1 = 272 * b + i;                                            // both padded layouts, spelled out: t < 256
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], e1 + 16 * j);                 // P1(b * 256 + j * 16 + i)
    NTT_BARRIER();
    const RowTable<tw_t> tB = row_table(a.twB, 16, 16, i);
    if constexpr (FUSE && (NTT_ROW_FUSE_TW & 2)) {
        const RowTable2 tB2 = row_table2(a.twB2, 16, i);
        fused_pass<MM, LAZY, 16>(x, a.roots, m, tB, tB2);
    } else {
        plain_pass<MM, LAZY, WIDE, 16>(x, a.roots, m, tB);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_ST(e1 + 17 * k, x[k]);                 // P2(b * 256 + k * 16 + i)
    NTT_BARRIER();
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], 17 * t + j);                  // P2(t * 16 + j)
    NTT_BARRIER();
    reg_ntt<MM, LAZY, 4, false>(x, a.roots, m);
}

// Both operands' forward transforms side by side (fused passes only): one read of every table entry serves both, so a
// workgroup reads 46 KiB less from its tables per row (a third of its table reads; each operand's exchange still has
// the tile to itself).
template <class MM, bool LAZY, int COLS>
__device__ __forceinline__ void fused_pass2(uint32_t (&xa)[16], uint32_t (&xb)[16], const Roots64<MM> &roots, const ModParams &m,
                                            const RowTable<typename MM::tw_t> &tab, const RowTable2 &tab2)
{
    typename MM::tw_t tv[16];
    uint2 t2[8];
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if ((k & 4) == 0) tv[k] = tab.template at<COLS>(k);
#pragma unroll
    for (int q = 0; q < 8; ++q) t2[q] = tab2.at<COLS>(q);
    __builtin_amdgcn_sched_barrier(0);
    reg_ntt16_tw<MM, LAZY>(xa, roots, m, [&](int k) { return tv[k]; }, [&](int q) { return t2[q]; });
    reg_ntt16_tw<MM, LAZY>(xb, roots, m, [&](int k) { return tv[k]; }, [&](int q) { return t2[q]; });
}
// the same for the passes that are not fused (Barrett kernels, wide-range kernels)
template <class MM, bool LAZY, bool WIDE, int COLS>
__device__ __forceinline__ void plain_pass2(uint32_t (&xa)[16], uint32_t (&xb)[16], const Roots64<MM> &roots, const ModParams &m,
                                            const RowTable<typename MM::tw_t> &tab)
{
    using O = Ops<MM, LAZY>;
    typename MM::tw_t tv[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) tv[k] = tab.template at<COLS>(k);
    __builtin_amdgcn_sched_barrier(0);
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(xa, roots, m);
    xa[0] = row_tighten<WIDE>(xa[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) xa[k] = O::mul_tw(xa[k], tv[k], m);
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(xb, roots, m);
    xb[0] = row_tighten<WIDE>(xb[0], m, O::bound(m));
#pragma unroll
    for (int k = 1; k < 16; ++k) xb[k] = O::mul_tw(xb[k], tv[k], m);
}
template <class MM, bool LAZY, bool WIDE = false, bool FUSE = true>
__device__ __forceinline__ void row_forward2(uint32_t (&xa)[16], uint32_t (&xb)[16], uint32_t *lds, const uint32_t t,
                                             const RowArgs<MM> &a, const ModParams &m)
{
    using tw_t = typename MM::tw_t;
    const uint32_t b = t >> 4, i = t & 15;
    const uint32_t e1 = 272 * b + i;
    if constexpr (FUSE) fused_pass2<MM, LAZY, 256>(xa, xb, a.roots, m, row_table(a.twA, 16, 256, t), row_table2(a.twA2, 256, t));
    else plain_pass2<MM, LAZY, WIDE, 256>(xa, xb, a.roots, m, row_table(a.twA, 16, 256, t));
    auto exchange1 = [&](uint32_t (&x)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) NTT_LDS_ST(t + 272 * k, x[k]);
        NTT_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], e1 + 16 * j);
        NTT_BARRIER();
    };
    exchange1(xa);
    exchange1(xb);
    if constexpr (FUSE) fused_pass2<MM, LAZY, 16>(xa, xb, a.roots, m, row_table(a.twB, 16, 16, i), row_table2(a.twB2, 16, i));
    else plain_pass2<MM, LAZY, WIDE, 16>(xa, xb, a.roots, m, row_table(a.twB, 16, 16, i));
    auto exchange2 = [&](uint32_t (&x)[16]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) NTT_LDS_ST(e1 + 17 * k, x[k]);
        NTT_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; ++j) NTT_LDS_LD(x[j], 17 * t + j);
        NTT_BARRIER();
    };
    exchange2(xa);
    exchange2(xb);
    reg_ntt<MM, LAZY, 4, false>(xa, a.roots, m);
    reg_ntt<MM, LAZY, 4, false>(xb, a.roots, m);
}

// x[k] = spectrum at position 16*t + k on entry; on exit x[j] = row[t + 256*j] (unscaled).
// FUSE: the twiddles of the second and third pass are applied by the thread that PRODUCES the value, fused into the
// pass in front of the exchange (row k of a table becomes a per-thread row; the tables are symmetric, w^(k * i)).
template <class MM, bool LAZY, bool WIDE = false, bool FUSE = false>
__device__ __forceinline__ void row_inverse(uint32_t (&x)[16], uint32_t *lds, const uint32_t t,
                                            const RowArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    if constexpr (FUSE) {
        using tw_t = typename MM::tw_t;
        const uint32_t b = t >> 4, i = t & 15;
        const uint32_t e1 = 272 * b + i;
        // the reader (b', i') of exchange 2 multiplied its k-th value by itwB[k][i']; that value is output j = i' of thread
        // 16 b' + k: entry [t & 15][j] = [j][t & 15]
        fused_pass<MM, LAZY, 16>(x, a.iroots, m, row_table(a.itwB, 16, 16, i), row_table2(a.itwB2, 16, i));
#pragma unroll
        for (int j = 0; j < 16; ++j) NTT_LDS_ST(17 * t + j, x[j]);
        NTT_BARRIER();
#pragma unroll
        for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], e1 + 17 * k);
        NTT_BARRIER();
        // the reader t' of exchange 1 multiplied its k-th value by itwA[k][t']; that value is output j of thread (b, i) with
        // k = b, t' = 16 j + i: entry [b][16 j + i], and output 0 has a twiddle as well
        if constexpr ((NTT_ROW_FUSE_TW & 8) != 0) {
            fused_pass<MM, LAZY, 16, 16, true>(x, a.iroots, m, row_table(a.itwA, 16, 256, b * 256 + i),
                                               row_table2(a.itwA2, 16, b * 128 + i, 128));
        } else {                                                // the same twiddles, not fused (9 table words per lane less)
            const RowTable<tw_t> tA = row_table(a.itwA, 16, 256, b * 256 + i);
            tw_t tv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) tv[k] = tA.template at<16>(k);
            __builtin_amdgcn_sched_barrier(0);
            reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = O::mul_tw(x[k], tv[k], m);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) NTT_LDS_ST(e1 + 16 * j, x[j]);
        NTT_BARRIER();
#pragma unroll
        for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], t + 272 * k);
        NTT_BARRIER();
        reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);
        return;
    }
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);     // loose values travel through LDS
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_ST(17 * t + j, x[j]);                  // P2(t * 16 + j)
    NTT_BARRIER();
    const uint32_t b = t >> 4, i = t & 15;
    const uint32_t e1 = 272 * b + i;                                            // both padded layouts, spelled out: t < 256
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], e1 + 17 * k);                 // P2(b * 256 + k * 16 + i)
    NTT_BARRIER();
    using tw_t = typename MM::tw_t;
    const RowTable<tw_t> tB = row_table(a.itwB, 16, 16, i);
    twiddle_pass<MM, LAZY, WIDE, 16>(x, m, tB);
    reg_ntt<MM, LAZY, 4, false, true, WIDE>(x, a.iroots, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) NTT_LDS_ST(e1 + 16 * j, x[j]);                 // P1(b * 256 + j * 16 + i)
    NTT_BARRIER();
#pragma unroll
    for (int k = 0; k < 16; ++k) NTT_LDS_LD(x[k], t + 272 * k);                  // P1(k * 256 + t)
    NTT_BARRIER();
    const RowTable<tw_t> tA = row_table(a.itwA, 16, 256, t);
    twiddle_pass<MM, LAZY, WIDE, 256>(x, m, tA);
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
    const bufrsrc_t ra = make_buf(pa, kRowLen * 4u), rb = make_buf(pb, kRowLen * 4u);
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = buf_load(ra, t * 4u, 1024u * j);
#pragma unroll
    for (int j = 0; j < 16; ++j) xb[j] = buf_load(rb, t * 4u, 1024u * j);
#endif
    constexpr bool FUSE = row_fuse<MM, WIDE>();
    if constexpr (FUSE && NTT_ROW_SHARE_TW && (NTT_ROW_FUSE_TW & 3) == 3) {
        row_forward2<MM, LAZY>(xa, xb, lds, t, a, m);
    } else if constexpr (!FUSE && (NTT_ROW_SHARE_TW & 2)) {
        row_forward2<MM, LAZY, WIDE, false>(xa, xb, lds, t, a, m);
    } else {
        row_forward<MM, LAZY, WIDE, FUSE>(xa, lds, t, a, m);
        row_forward<MM, LAZY, WIDE, FUSE>(xb, lds, t, a, m);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY, WIDE, row_fuse_inv<MM, WIDE>()>(xa, lds, t, a, m);
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
    const bufrsrc_t ro = make_buf(po, kRowLen * 4u);
#pragma unroll
    for (int j = 0; j < 16; ++j) buf_store(xa[j], ro, t * 4u, 1024u * j);
#endif
}

}  // namespace ntt
