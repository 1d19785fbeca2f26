// kernels_fused.hip -- whole products of transform length L = 8192 or 16384 in ONE kernel, gfx950.
//
// For these lengths the 4-step decomposition is a column level of N1 = 2 or 4 points around rows of
// 4096: as three kernels (col_kernel, row_mul_kernel, col_kernel) the tiny column level costs two full
// passes over memory and two more launches.  Here one workgroup owns one product and takes its N1 rows one
// after the other on one LDS tile, with the registers of ONE row live at a time.  For row k1 = 0 .. N1-1 a
// thread recomputes its 16 points of that row of the forward column level from the caller's operands -- a
// dot product of the NIN nonzero inputs of the column with one table (4-step twiddle, N1-point root and
// negacyclic twist folded into (psi w_L^k1)^i), so nothing of the other rows is kept --, runs row_forward x2 /
// pointwise / row_inverse (row_body.hpp, the body of row_mul_kernel), multiplies by the inverse level's twiddle
// and parks the 16 results; after the last row the N1-point inverse, scale, inverse twist, canonical store and
// the 2n-1 cut finish in registers.  N1 - 1 parked rows + two live ones: 94 - 123 VGPRs, four workgroups per CU
// (at N1 = 4 the compiler parks the rows in scratch memory, 208 B per thread, which measures faster than holding
// them in registers: see seq_row); the row loop stays rolled (one copy of the row body in the instruction cache).
// The operands are read N1 times (from L1 / L2 after the first) and the product written once.  Same row tables and
// inverse twiddles as the three-kernel path, same results.
//
// Serves (profiles/time_medium_r02.txt, us per product, three kernels -> this kernel): linear products of
// 2048 < n <= 4096 (L = 8192, upper half zero: 0.077 -> 0.044) and 4096 < n <= 8192 (L = 16384: 0.137 -> 0.113),
// and the full-length L = 8192 of a negacyclic product (0.083 -> 0.054).  The full-length L = 16384 (four
// inputs per column, sixteen dot-product terms) only ties with the three kernels and stays there.
//
// History: round 1 shipped a form for the linear L = 8192 only that held both rows of both operands in
// registers (fused_product_kernel, 168 VGPRs, three workgroups per CU, 0.043 - 0.047 us); its N1 = 4 and
// full-length variants needed 128 data registers or spilled and lost to the three kernels.  What made the
// row-at-a-time form fit: buffer-descriptor loads (no per-load bounds branch), a rolled row loop, and an opaque
// copy of the thread index per iteration so that the addresses and table loads of one row are not hoisted or
// kept for the next.
//
// Replaces, like the kernels it fuses, the stages of NTT_iterative (main-montgomery.cc:255-295) and the
// pointwise loop (:341-344) for n in (2048, 8192].
#include "row_body.hpp"

namespace ntt {

// one row: forward a, forward b, pointwise, inverse (loose, unscaled) -- the body of row_mul_kernel on registers
template <class MM, bool LAZY>
__device__ __forceinline__ void fused_row(uint32_t (&xa)[16], uint32_t (&xb)[16], uint32_t *lds, const uint32_t t,
                                          const RowArgs<MM> &row, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    row_forward<MM, LAZY, false, row_fuse<MM, false>()>(xa, lds, t, row, m);
    row_forward<MM, LAZY, false, row_fuse<MM, false>()>(xb, lds, t, row, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY, false, row_fuse_inv<MM, false>()>(xa, lds, t, row, m);
}

// one row of seq_product_kernel: recompute the forward column level, transform, multiply, inverse, inverse twiddle, park
template <class MM, bool LAZY, int LOGN1, bool HALF>
__device__ __forceinline__ void seq_row(const int k1, uint32_t (&xa)[16], uint32_t (&xb)[16], uint32_t (&park)[(1 << LOGN1) - 1][16],
                                        uint32_t *lds, const uint32_t t, const bufrsrc_t pa, const bufrsrc_t pb,
                                        const bufrsrc_t tab, const SeqArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    using tw_t = typename MM::tw_t;
    constexpr int N1 = 1 << LOGN1, NIN = HALF ? N1 / 2 : N1;
    // the thread index as the loop body sees it: opaque, so that the addresses derived from it (16 columns, LDS
    // slots, table entries) are computed where they are used instead of hoisted out of the loop and spilled
    uint32_t tt = t;
    asm volatile("" : "+v"(tt));
    tt &= 255u;                                           // (and known to be small: 32-bit offsets on scalar bases)
    // G columns at a time, fenced for the instruction scheduler: left alone it issues the loads of all 16 columns
    // first and spills what it parked
    constexpr int G = 8 / NIN;                                // (4, 8 or 16 loads in flight per operand measure the same)
    if (k1 == 0 && a.plain_row0) {                        // row 0 of a cyclic transform: the column's sum
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += G) {
#pragma unroll
            for (int j = j0; j < j0 + G; ++j) {
                const uint32_t i2 = tt + 256u * j;
                uint32_t sa = 0, sb = 0;
#pragma unroll
                for (int i1 = 0; i1 < NIN; ++i1) {
                    const uint32_t idx = (uint32_t)i1 * kRowLen + i2;
                    const uint32_t ca = buf_load(pa, idx * 4u, 0), cb = buf_load(pb, idx * 4u, 0);   // 0 beyond in_len
                    sa = i1 ? O::add(sa, ca, m) : ca;
                    sb = i1 ? O::add(sb, cb, m) : cb;
                }
                xa[j] = sa;
                xb[j] = sb;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const uint32_t trow = (uint32_t)k1 * NIN * kRowLen * (uint32_t)sizeof(tw_t);   // row k1 of the table
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += G) {
#pragma unroll
            for (int j = j0; j < j0 + G; ++j) {
                const uint32_t i2 = tt + 256u * j;
                uint32_t sa = 0, sb = 0;
#pragma unroll
                for (int i1 = 0; i1 < NIN; ++i1) {
                    const uint32_t idx = (uint32_t)i1 * kRowLen + i2;
                    const uint32_t ca = buf_load(pa, idx * 4u, 0), cb = buf_load(pb, idx * 4u, 0);   // 0 beyond in_len
                    const tw_t w = buf_load_tw<tw_t>(tab, idx << tw_shift<tw_t>(), trow);
                    const uint32_t va = O::mul_tw(ca, w, m), vb = O::mul_tw(cb, w, m);
                    sa = i1 ? O::add(sa, va, m) : va;
                    sb = i1 ? O::add(sb, vb, m) : vb;
                }
                xa[j] = sa;
                xb[j] = sb;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    fused_row<MM, LAZY>(xa, xb, lds, tt, a.r.row, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = O::mul_tw(xa[j], a.tw_inv[(uint32_t)k1 * kRowLen + tt + 256u * j], m);
    // (N1 = 4: the compiler turns these wave-uniform copies into one dynamically addressed store, i.e. the rows are
    // parked in scratch memory; forcing them into registers with selects -- 128 VGPRs -- is 17 - 30 % slower:
    // 0.135 vs 0.116 us at L = 16384, 0.072 vs 0.056 us at the full-length L = 8192 with its one parked row)
#pragma unroll
    for (int q = 0; q < N1 - 1; ++q)
        if (k1 == q) {                                    // wave-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) park[q][j] = xa[j];
        }
}

template <class MM, bool LAZY, int LOGN1, bool HALF>
__global__ void __launch_bounds__(256, 4) seq_product_kernel(const SeqArgs<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    using tw_t = typename MM::tw_t;
    constexpr int N1 = 1 << LOGN1, NIN = HALF ? N1 / 2 : N1;
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t t = threadIdx.x;
    // the range check of the descriptors is the zero padding (loads at or beyond in_len return 0) and the output cut
    const bufrsrc_t pa = make_buf(a.r.row.a + (size_t)blockIdx.x * a.r.in_stride, a.r.in_len * 4u);
    const bufrsrc_t pb = make_buf(a.r.row.b + (size_t)blockIdx.x * a.r.in_stride, a.r.in_len * 4u);
    const bufrsrc_t tab = make_buf(a.tw_in, (uint32_t)(N1 * NIN * kRowLen * sizeof(tw_t)));
    uint32_t park[N1 - 1][16];
    uint32_t xa[16], xb[16];
#pragma nounroll
    for (int k1 = 0; k1 < N1; ++k1) seq_row<MM, LAZY, LOGN1, HALF>(k1, xa, xb, park, lds, t, pa, pb, tab, a, m);
    const bufrsrc_t po = make_buf(a.r.row.out + (size_t)blockIdx.x * a.r.out_stride, a.r.out_len * 4u);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i2 = t + 256u * j;
        uint32_t c[N1];
#pragma unroll
        for (int q = 0; q < N1 - 1; ++q) c[q] = park[q][j];
        c[N1 - 1] = xa[j];
        reg_ntt<MM, LAZY, LOGN1, false, true>(c, a.r.row.iroots, m);              // loose outputs
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) {
            const uint32_t idx = (uint32_t)i1 * kRowLen + i2;
            uint32_t v;
            if (a.r.itwist) v = O::canon(O::mul_tw(c[i1], a.r.itwist[idx], m), m);   // negacyclic: result * psi^-i
            else v = O::canon(O::tighten(c[i1], m), m);
            buf_store(v, po, idx * 4u, 0);                               // dropped at or beyond out_len
        }
    }
}

template <class MM, bool LAZY>
static void launch_seq_t(int logN1, bool half, const SeqArgs<MM> &a, const ModParams &mp, unsigned batch, hipStream_t s)
{
    if (logN1 == 1 && half) hipLaunchKernelGGL((seq_product_kernel<MM, LAZY, 1, true>), dim3(batch), dim3(256), 0, s, a, mp);
    else if (logN1 == 1) hipLaunchKernelGGL((seq_product_kernel<MM, LAZY, 1, false>), dim3(batch), dim3(256), 0, s, a, mp);
    else if (logN1 == 2 && half) hipLaunchKernelGGL((seq_product_kernel<MM, LAZY, 2, true>), dim3(batch), dim3(256), 0, s, a, mp);
    else if (logN1 == 2) hipLaunchKernelGGL((seq_product_kernel<MM, LAZY, 2, false>), dim3(batch), dim3(256), 0, s, a, mp);
    else note_unsupported_launch("seq_product_kernel: column level of 2^logN1 points, logN1 =", logN1);
}

void launch_seq_product(int kind, bool lazy, int logN1, bool half, const void *args, const ModParams &mp,
                        unsigned batch, hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const SeqArgs<Mont32> *>(args);
        if (lazy) launch_seq_t<Mont32, true>(logN1, half, a, mp, batch, s);
        else launch_seq_t<Mont32, false>(logN1, half, a, mp, batch, s);
    } else {
        const auto &a = *static_cast<const SeqArgs<Barrett32> *>(args);
        if (lazy) launch_seq_t<Barrett32, true>(logN1, half, a, mp, batch, s);
        else launch_seq_t<Barrett32, false>(logN1, half, a, mp, batch, s);
    }
}

}  // namespace ntt
