// kernels_col16.hip -- LDS-staged column pass for N1 = 32 and 64 (N1 = 16 * R, R = 2 or 4), gfx950.
//
// The register-only column kernel (kernels_col.hip) needs ~230 VGPRs for 64 points per thread
// (2 waves per SIMD) and rocprofv3 shows it ~55 % VALU-busy.  Here a workgroup of 256 threads
// owns a tile of C = 256/R columns x N1 rows and every thread keeps 16 points:
//   pass 1: thread (c, r) loads rows 4a + r (resp. 2a + r), a < 16, runs a 16-point register NTT
//           over a and multiplies by the wave-uniform inner twiddle w_N1^(r*ka)   (SGPR operands)
//   LDS   : one exchange of the 16 KiB tile, laid out [r][ka][c]: reads and writes are
//           lane-consecutive, bank-conflict free, one barrier
//   pass 2: thread (c, q) takes the ka in [q*16/R, (q+1)*16/R), runs the radix-R NTT over r, and
//           owns output rows k = ka + 16*kr, which it multiplies by the 4-step twiddle and stores
// Same arithmetic as col_kernel (X[ka + 16 kr] = sum_r w_R^(r kr) w_N1^(r ka) sum_a x[R a + r] w_16^(a ka)),
// ~45 VGPRs, 8 waves per SIMD.  Across the 64 lanes of a wave every global access is 256 contiguous
// bytes, as before.
#include "col16_body.hpp"

namespace ntt {

template <class MM, bool LAZY, int LOGR, bool INVERSE, bool HALF>
__global__ void __launch_bounds__(LOGR <= 2 ? 256 : 64 << LOGR) col16_kernel(const ColArgs<MM> a, const ModParams m)
{
    __shared__ uint32_t lds[LOGR <= 2 ? 4096 : 1024 << LOGR];
    col16_body<MM, LAZY, LOGR, INVERSE, HALF>(a, m, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <class MM, bool LAZY, int LOGR>
static void launch_t(bool inverse, bool half_in, const ColArgs<MM> &a, const ModParams &mp, dim3 grid,
                     hipStream_t s)
{
    const dim3 block(LOGR <= 2 ? 256 : 64 << LOGR);
    if (inverse)
        hipLaunchKernelGGL((col16_kernel<MM, LAZY, LOGR, true, false>), grid, block, 0, s, a, mp);
    else if (half_in)
        hipLaunchKernelGGL((col16_kernel<MM, LAZY, LOGR, false, true>), grid, block, 0, s, a, mp);
    else
        hipLaunchKernelGGL((col16_kernel<MM, LAZY, LOGR, false, false>), grid, block, 0, s, a, mp);
}

template <class MM, bool LAZY>
static void launch_l(int logN1, bool inverse, bool half_in, const ColArgs<MM> &a, const ModParams &mp,
                     unsigned cols_per_row, unsigned batch, unsigned nops, hipStream_t s)
{
    if (logN1 == 8) launch_t<MM, LAZY, 4>(inverse, half_in, a, mp, dim3(cols_per_row / 64, batch, nops), s);
    else if (logN1 == 7) launch_t<MM, LAZY, 3>(inverse, half_in, a, mp, dim3(cols_per_row / 64, batch, nops), s);
    else if (logN1 == 6) launch_t<MM, LAZY, 2>(inverse, half_in, a, mp, dim3(cols_per_row / 64, batch, nops), s);
    else if (logN1 == 5) launch_t<MM, LAZY, 1>(inverse, half_in, a, mp, dim3(cols_per_row / 128, batch, nops), s);
    else note_unsupported_launch("col16_kernel: no instantiation for log2(N1) =", logN1);
}

void launch_col16(int kind, bool lazy, int logN1, bool inverse, bool half_in, const void *args,
                  const ModParams &mp, unsigned cols_per_row, unsigned batch, unsigned nops,
                  hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const ColArgs<Mont32> *>(args);
        if (lazy) launch_l<Mont32, true>(logN1, inverse, half_in, a, mp, cols_per_row, batch, nops, s);
        else launch_l<Mont32, false>(logN1, inverse, half_in, a, mp, cols_per_row, batch, nops, s);
    } else {
        const auto &a = *static_cast<const ColArgs<Barrett32> *>(args);
        if (lazy) launch_l<Barrett32, true>(logN1, inverse, half_in, a, mp, cols_per_row, batch, nops, s);
        else launch_l<Barrett32, false>(logN1, inverse, half_in, a, mp, cols_per_row, batch, nops, s);
    }
}

}  // namespace ntt
