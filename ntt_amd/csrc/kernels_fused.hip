// kernels_fused.hip -- whole products of transform length L = 8192 or 16384 in ONE kernel, gfx950.
//
// For these lengths the 4-step decomposition is a column level of N1 = 2 or 4 points around rows of
// 4096: as three kernels (col_kernel, row_mul_kernel, col_kernel) the tiny column level costs two full
// passes over memory and two more launches.  Here one workgroup owns one product: every thread loads its
// columns (stride 4096), runs the N1-point column transform and the 4-step twiddle in registers -- the
// forward half of col_kernel --, which leaves it holding 16 points of each of the N1 rows exactly as
// row_forward wants them (row[t + 256 j]); the rows go through row_forward / pointwise / row_inverse one
// after the other on one LDS tile, and the inverse column step (twiddle, N1-point inverse, scale, twist,
// canonical store with the 2n-1 cut) finishes in registers again.  Operands are read once, the product
// written once; nothing else touches memory.  Same tables as the three-kernel path (level-0 column
// twiddles, row tables), same arithmetic, same results.
//
// Shipped for N1 = 2 with a zero upper half only, i.e. linear products of 2048 < n <= 4096 (0.077 -> 0.043 us
// per product, tools/sweep_small.py).  Tried and left on the three-kernel path, each parity-green but slower
// than it: the full-length input of a negacyclic product (twice the loads and the twist: 246 VGPRs, or 168
// with spills; also with operand a's spectra parked in LDS, and with buffer-descriptor addressing), N1 = 4
// (128 data registers), and a form with 256 * N1 threads and one row per four-wave thread group (registers
// fit, but with one or two workgroups per CU nothing runs while a group waits at its barriers).
//
// Replaces, like the kernels it fuses, the stages of NTT_iterative (main-montgomery.cc:255-295) and the
// pointwise loop (:341-344) for n in (2048, 4096].
#include "row_body.hpp"

namespace ntt {

template <class MM, bool LAZY, int LOGN1, bool HALF>
__device__ __forceinline__ void fused_load(uint32_t (&x)[1 << LOGN1][16], const uint32_t *__restrict__ p, const uint32_t t,
                                           const FusedArgs<MM> &a, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    constexpr int N1 = 1 << LOGN1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i2 = t + 256u * j;
        uint32_t c[N1];
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) {
            const uint32_t idx = (uint32_t)i1 * kRowLen + i2;
            if (HALF && i1 >= N1 / 2) c[i1] = 0;
            else c[i1] = idx < a.r.in_len ? p[idx] : 0u;
        }
        if (a.r.twist) {                                      // negacyclic: x[i] * psi^i (uniform branch)
#pragma unroll
            for (int i1 = 0; i1 < N1; ++i1)
                if (!(HALF && i1 >= N1 / 2)) c[i1] = O::mul_tw(c[i1], a.r.twist[(uint32_t)i1 * kRowLen + i2], m);
        }
        reg_ntt<MM, LAZY, LOGN1, HALF, true>(c, a.r.row.roots, m);
        c[0] = O::tighten(c[0], m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) c[k1] = O::mul_tw(c[k1], a.tw_fwd[(uint32_t)k1 * kRowLen + i2], m);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) x[k1][j] = c[k1];
    }
}

// one row: forward a, forward b, pointwise, inverse (loose, unscaled) -- the body of row_mul_kernel on registers
template <class MM, bool LAZY>
__device__ __forceinline__ void fused_row(uint32_t (&xa)[16], uint32_t (&xb)[16], uint32_t *lds, const uint32_t t,
                                          const RowArgs<MM> &row, const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    row_forward<MM, LAZY>(xa, lds, t, row, m);
    row_forward<MM, LAZY>(xb, lds, t, row, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY>(xa, lds, t, row, m);
}

template <class MM, bool LAZY, int LOGN1, bool HALF>
#ifndef NTT_FUSED_WAVES
#define NTT_FUSED_WAVES 3
#endif
__global__ void __launch_bounds__(256, NTT_FUSED_WAVES) fused_product_kernel(const FusedArgs<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    constexpr int N1 = 1 << LOGN1;
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t t = threadIdx.x;
    const uint32_t *__restrict__ pa = a.r.row.a + (size_t)blockIdx.x * a.r.in_stride;
    const uint32_t *__restrict__ pb = a.r.row.b + (size_t)blockIdx.x * a.r.in_stride;
    uint32_t xa[N1][16], xb[N1][16];
    fused_load<MM, LAZY, LOGN1, HALF>(xa, pa, t, a, m);
    fused_load<MM, LAZY, LOGN1, HALF>(xb, pb, t, a, m);
    // rows one after the other on the one LDS tile (spelled out: the optimizer does not unroll a loop with
    // this body, and a rolled loop would index the register arrays dynamically)
    fused_row<MM, LAZY>(xa[0], xb[0], lds, t, a.r.row, m);
    fused_row<MM, LAZY>(xa[1], xb[1], lds, t, a.r.row, m);
    static_assert(N1 == 2, "two rows per thread: see the header for what was tried beyond");
    uint32_t *__restrict__ po = a.r.row.out + (size_t)blockIdx.x * a.r.out_stride;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i2 = t + 256u * j;
        uint32_t c[N1];
        c[0] = O::mul_tw(xa[0][j], a.r.scale, m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) c[k1] = O::mul_tw(xa[k1][j], a.tw_inv[(uint32_t)k1 * kRowLen + i2], m);
        reg_ntt<MM, LAZY, LOGN1, false, true>(c, a.r.row.iroots, m);          // loose outputs
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) {
            const uint32_t idx = (uint32_t)i1 * kRowLen + i2;
            uint32_t v;
            if (a.r.itwist) v = O::canon(O::mul_tw(c[i1], a.r.itwist[idx], m), m);   // negacyclic: result * psi^-i
            else v = O::canon(O::tighten(c[i1], m), m);
            if (idx < a.r.out_len) po[idx] = v;
        }
    }
}

template <class MM, bool LAZY>
static void launch_fused_t(const FusedArgs<MM> &a, const ModParams &mp, unsigned batch, hipStream_t s)
{
    hipLaunchKernelGGL((fused_product_kernel<MM, LAZY, 1, true>), dim3(batch), dim3(256), 0, s, a, mp);
}

void launch_fused_product(int kind, bool lazy, const void *args, const ModParams &mp,
                          unsigned batch, hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const FusedArgs<Mont32> *>(args);
        if (lazy) launch_fused_t<Mont32, true>(a, mp, batch, s);
        else launch_fused_t<Mont32, false>(a, mp, batch, s);
    } else {
        const auto &a = *static_cast<const FusedArgs<Barrett32> *>(args);
        if (lazy) launch_fused_t<Barrett32, true>(a, mp, batch, s);
        else launch_fused_t<Barrett32, false>(a, mp, batch, s);
    }
}

}  // namespace ntt
