// kernels_dist.hip -- the column pass of ONE transform split over G devices, with the two exchanges of the
// distributed 4-step inside the kernel, gfx950.
//
// Slot h of a multi-device context owns the columns i2 in [h * cols, (h + 1) * cols), cols = S / G, of the G x S
// matrix x[i1][i2] whose row i1 lives on slot i1.  One thread owns one column: it LOADS its G points through a table
// of per-row base pointers -- rows of peer devices are read over xGMI in place (or from a local staging array filled
// by peer copies) --, runs the G-point transform in registers, applies the 4-step twiddle w_L^(i2 * k1) and STORES row
// k1 through a second pointer table straight into slot k1's array (or into a local send array).  With both tables
// pointing at the peers the whole "all-to-all, column pass, all-to-all" of ntt_amd/shard.py::dist_transform is this
// one kernel per device: 256 contiguous bytes per wave and peer in either direction.
//
// The reference has nothing like it (its transforms live in static arrays of 300000 on one host,
// main-montgomery.cc:297; its only exchange is the gather of main-mpi.cc:772-791); BASELINE.json configs[4].
#include "device_ntt.hpp"

namespace ntt {

template <class MM, bool LAZY, int LOGG, bool INVERSE, bool HALF = false>
__global__ void __launch_bounds__(256) dist_col_kernel(const DistColArgs<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    constexpr int G = 1 << LOGG;
    static_assert(!(HALF && INVERSE), "zero padding is an input of the forward transform");
    const uint32_t j = a.col0 + blockIdx.x * 256u + threadIdx.x;       // column within this slot's piece
    if (j >= a.col0 + a.ncols) return;                                  // (the host launches whole workgroups; belt and braces)
    uint32_t x[G];
#pragma unroll
    for (int i1 = 0; i1 < G; ++i1) x[i1] = (HALF && i1 >= G / 2) ? 0u : a.in[i1][j];   // HALF: the upper blocks are not read at all
    if constexpr (INVERSE) {
        // canonical inputs (the local inverse transform stores canonical residues); row 0 carries G^-1
        x[0] = a.use_scale0 ? O::mul_tw(x[0], a.scale0, m) : x[0];
#pragma unroll
        for (int k1 = 1; k1 < G; ++k1) x[k1] = O::mul_tw(x[k1], a.tw[(size_t)k1 * a.cols + j], m);
    }
    reg_ntt<MM, LAZY, LOGG, HALF, true>(x, a.roots, m);                 // loose outputs
    if constexpr (!INVERSE) {
        x[0] = O::tighten(x[0], m);
#pragma unroll
        for (int k1 = 1; k1 < G; ++k1) x[k1] = O::mul_tw(x[k1], a.tw[(size_t)k1 * a.cols + j], m);
    }
#pragma unroll
    for (int k1 = 0; k1 < G; ++k1) a.out[k1][j] = O::canon(INVERSE ? O::tighten(x[k1], m) : x[k1], m);
}

template <class MM, bool LAZY>
static bool launch_dist_l(int logG, bool inverse, const DistColArgs<MM> &a, const ModParams &mp, hipStream_t s)
{
    const dim3 grid((a.ncols + 255u) / 256u), block(256);
#define NTT_DIST_CASE(LG)                                                                                         \
    case LG:                                                                                                      \
        if (inverse) hipLaunchKernelGGL((dist_col_kernel<MM, LAZY, LG, true>), grid, block, 0, s, a, mp);         \
        else if (a.half) hipLaunchKernelGGL((dist_col_kernel<MM, LAZY, LG, false, true>), grid, block, 0, s, a, mp); \
        else hipLaunchKernelGGL((dist_col_kernel<MM, LAZY, LG, false>), grid, block, 0, s, a, mp);                \
        return true;
    switch (logG) {
        NTT_DIST_CASE(1)
        NTT_DIST_CASE(2)
        NTT_DIST_CASE(3)
        NTT_DIST_CASE(4)
    default:
        note_unsupported_launch("dist_col_kernel: no instantiation for log2(world) =", logG);
        return false;
    }
#undef NTT_DIST_CASE
}

// Pointwise product of two spectra (main-montgomery.cc:341-344).  Montgomery: REDC(a b) carries R^-1; one more
// multiplication by R (table form R^2 mod p) removes it, so that the inverse transform's plain L^-1 scale applies.
template <class MM, bool LAZY>
__global__ void __launch_bounds__(256) dist_pointwise_kernel(uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t count,
                                                             const ModParams m)
{
    using O = Ops<MM, LAZY>;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;                                             // count is a multiple of 4 (a power of two >= 256)
    const uint4 va = *reinterpret_cast<const uint4 *>(a + i), vb = *reinterpret_cast<const uint4 *>(b + i);
    uint32_t x[4] = {va.x, va.y, va.z, va.w};
    const uint32_t y[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t t = O::mul(x[k], y[k], m);
        if constexpr (MM::kMontgomery) t = O::mul_tw(t, typename MM::tw_t(m.r2), m);
        x[k] = t;
    }
    *reinterpret_cast<uint4 *>(a + i) = make_uint4(x[0], x[1], x[2], x[3]);
}

void launch_dist_pointwise(int kind, bool lazy, uint32_t *a, const uint32_t *b, size_t count, const ModParams &mp, hipStream_t s)
{
    const dim3 grid((unsigned)((count / 4 + 255) / 256)), block(256);
    if (kind == 0) {
        if (lazy) hipLaunchKernelGGL((dist_pointwise_kernel<Mont32, true>), grid, block, 0, s, a, b, count, mp);
        else hipLaunchKernelGGL((dist_pointwise_kernel<Mont32, false>), grid, block, 0, s, a, b, count, mp);
    } else {
        if (lazy) hipLaunchKernelGGL((dist_pointwise_kernel<Barrett32, true>), grid, block, 0, s, a, b, count, mp);
        else hipLaunchKernelGGL((dist_pointwise_kernel<Barrett32, false>), grid, block, 0, s, a, b, count, mp);
    }
}

bool launch_dist_col(int kind, bool lazy, int logG, bool inverse, const void *args, const ModParams &mp, hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const DistColArgs<Mont32> *>(args);
        return lazy ? launch_dist_l<Mont32, true>(logG, inverse, a, mp, s) : launch_dist_l<Mont32, false>(logG, inverse, a, mp, s);
    }
    const auto &a = *static_cast<const DistColArgs<Barrett32> *>(args);
    return lazy ? launch_dist_l<Barrett32, true>(logG, inverse, a, mp, s) : launch_dist_l<Barrett32, false>(logG, inverse, a, mp, s);
}

}  // namespace ntt
