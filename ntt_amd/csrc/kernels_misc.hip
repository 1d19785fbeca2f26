// kernels_misc.hip -- small-length transforms, order permutation and the CRT kernels, gfx950.
#include "crt_device.hpp"
#include "device_ntt.hpp"

namespace ntt {

// ------------------------------------------------------------------------------------------------
// small_kernel: L <= 4096.  One workgroup per polynomial pair; the whole product (pad, two forward
// transforms, pointwise product, inverse transform, scaling) runs out of one 32 KiB LDS tile with
// radix-2 stages.  This is the plumbing path for the reference's tiny case (nttdata/0.in, n = 4)
// and for unit-test sizes; the throughput path is kernels_col/kernels_row.
// Replaces main-montgomery.cc:326-351 for len <= 4096.
// ------------------------------------------------------------------------------------------------
template <class MM, bool LAZY>
__device__ __forceinline__ void lds_dif(uint32_t *x, int logL, const typename MM::tw_t *__restrict__ w,
                                        const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    const uint32_t L = 1u << logL;
    for (int s = 0; s < logL; ++s) {
        const uint32_t half = L >> (s + 1);
        for (uint32_t idx = threadIdx.x; idx < L / 2; idx += 256) {
            const uint32_t j = idx & (half - 1);
            const uint32_t lo = ((idx - j) << 1) + j;
            const uint32_t u = x[lo], v = x[lo + half];
            x[lo] = O::add(u, v, m);
            x[lo + half] = O::mul_tw(O::sub_raw(u, v, m), w[j << s], m);
        }
        __syncthreads();
    }
}

template <class MM, bool LAZY>
__device__ __forceinline__ void lds_dit(uint32_t *x, int logL, const typename MM::tw_t *__restrict__ w,
                                        const ModParams &m)
{
    using O = Ops<MM, LAZY>;
    const uint32_t L = 1u << logL;
    for (int s = logL - 1; s >= 0; --s) {
        const uint32_t half = L >> (s + 1);
        for (uint32_t idx = threadIdx.x; idx < L / 2; idx += 256) {
            const uint32_t j = idx & (half - 1);
            const uint32_t lo = ((idx - j) << 1) + j;
            const uint32_t u = x[lo], v = O::mul_tw(x[lo + half], w[j << s], m);
            x[lo] = O::add(u, v, m);
            x[lo + half] = O::sub(u, v, m);
        }
        __syncthreads();
    }
}

template <class MM, bool LAZY>
__global__ void __launch_bounds__(256) small_kernel(const SmallArgs<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    __shared__ uint32_t A[kRowLen];
    __shared__ uint32_t B[kRowLen];
    const uint32_t L = 1u << a.logL;
    const uint32_t *__restrict__ pa = a.a + (size_t)blockIdx.x * a.in_stride;
    uint32_t *__restrict__ po = a.out + (size_t)blockIdx.x * a.out_stride;
    if (a.mode == 0) {
        const uint32_t *__restrict__ pb = a.b + (size_t)blockIdx.x * a.in_stride;
        for (uint32_t i = threadIdx.x; i < L; i += 256) {
            A[i] = i < a.in_len ? pa[i] : 0u;
            B[i] = i < a.in_len ? pb[i] : 0u;
            if (a.twist) {                                  // negacyclic: operands times psi^i
                A[i] = O::mul_tw(A[i], a.twist[i], m);
                B[i] = O::mul_tw(B[i], a.twist[i], m);
            }
        }
        __syncthreads();
        lds_dif<MM, LAZY>(A, a.logL, a.w, m);
        lds_dif<MM, LAZY>(B, a.logL, a.w, m);
        for (uint32_t i = threadIdx.x; i < L; i += 256) A[i] = O::mul(A[i], B[i], m);
        __syncthreads();
        lds_dit<MM, LAZY>(A, a.logL, a.iw, m);
        for (uint32_t i = threadIdx.x; i < L && i < a.out_len; i += 256)
            po[i] = csub(MM::mul_tw(A[i], a.itwist ? a.itwist[i] : a.scale, m), m.p);
    } else if (a.mode == 1) {
        for (uint32_t i = threadIdx.x; i < L; i += 256) A[i] = pa[i];
        __syncthreads();
        lds_dif<MM, LAZY>(A, a.logL, a.w, m);
        for (uint32_t i = threadIdx.x; i < L; i += 256)
            po[__brev(i) >> (32 - a.logL)] = O::canon(A[i], m);
    } else {
        for (uint32_t i = threadIdx.x; i < L; i += 256) A[__brev(i) >> (32 - a.logL)] = pa[i];
        __syncthreads();
        lds_dit<MM, LAZY>(A, a.logL, a.iw, m);
        for (uint32_t i = threadIdx.x; i < L; i += 256) po[i] = csub(MM::mul_tw(A[i], a.scale, m), m.p);
    }
}

void launch_small(int kind, bool lazy, const void *args, const ModParams &mp, unsigned batch, hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const SmallArgs<Mont32> *>(args);
        if (lazy) hipLaunchKernelGGL((small_kernel<Mont32, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else hipLaunchKernelGGL((small_kernel<Mont32, false>), dim3(batch), dim3(256), 0, s, a, mp);
    } else {
        const auto &a = *static_cast<const SmallArgs<Barrett32> *>(args);
        if (lazy) hipLaunchKernelGGL((small_kernel<Barrett32, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else hipLaunchKernelGGL((small_kernel<Barrett32, false>), dim3(batch), dim3(256), 0, s, a, mp);
    }
}

// ------------------------------------------------------------------------------------------------
// permute_kernel: converts between natural order and the engine's digit-reversed spectrum order
// for the public ntt_forward / ntt_inverse (NTT_iterative is natural-in / natural-out,
// main-montgomery.cc:255-295).  The multiply path never runs it.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) permute_kernel(const PermArgs pa)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    const size_t row = (size_t)blockIdx.y << pa.logL;
    int shift = pa.logL, acc = 0;
    uint32_t k = 0;
    for (int d = 0; d < pa.nlevels; ++d) {
        const int w = pa.level_log[d];
        shift -= w;
        k |= ((pos >> shift) & ((1u << w) - 1)) << acc;
        acc += w;
    }
    if (pa.to_natural) {
        uint32_t v = pa.src[row + pos];
        if (pa.canon_p) v = csub(v, pa.canon_p);
        pa.dst[row + k] = v;
    } else {
        pa.dst[row + pos] = pa.src[row + k];
    }
}

void launch_permute(const PermArgs &pa, unsigned batch, hipStream_t s)
{
    hipLaunchKernelGGL(permute_kernel, dim3((1u << pa.logL) / 256, batch), dim3(256), 0, s, pa);
}

// Tiled digit reversal (see PermTileArgs): both the loads and the stores of a workgroup are contiguous
// runs (2^b and 2^a elements), the reshuffle happens in a padded LDS tile.  The plain kernel above
// stores 4-byte words 2^(logL - w1) elements apart: 1.47 ms per 2^26-point array against 0.22 ms here.
template <int N>
__device__ __forceinline__ uint32_t apply_runs(uint32_t x, const BitRun (&r)[N], int n)
{
    uint32_t y = 0;
    for (int j = 0; j < n; ++j) y |= ((x >> r[j].from) & ((1u << r[j].width) - 1u)) << r[j].to;
    return y;
}

template <int LOGT>                                                       // tile of 2^LOGT elements
__global__ void __launch_bounds__(256) permute_tiled_kernel(const PermTileArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    constexpr int PER = (1 << LOGT) / 256;                                 // elements per thread
    const size_t row = (size_t)blockIdx.y << pa.logL;
    const uint32_t base = apply_runs(blockIdx.x, pa.mid, pa.n_mid);        // bits of i outside the tile
    const uint32_t nb = 1u << pa.b, pitch = nb + 1;
    uint32_t v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {                                        // contiguous input runs, all loads in flight
        const uint32_t e = threadIdx.x + 256u * j;
        const uint32_t l = e & (nb - 1), t = e >> pa.b;
        v[j] = pa.src[row + (base | apply_runs(t, pa.top, pa.n_top) | l)];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t e = threadIdx.x + 256u * j;
        tile[(e >> pa.b) * pitch + (e & (nb - 1))] = v[j];
    }
    __syncthreads();
    const uint32_t na = 1u << pa.a;
#pragma unroll
    for (int j = 0; j < PER; ++j) {                                        // contiguous output runs
        const uint32_t e = threadIdx.x + 256u * j;
        const uint32_t t = e & (na - 1), l = e >> pa.a;
        const uint32_t i = base | apply_runs(t, pa.top, pa.n_top) | l;
        uint32_t x = tile[t * pitch + l];
        if (pa.canon_p) x = csub(x, pa.canon_p);
        pa.dst[row + apply_runs(i, pa.rev, pa.n_rev)] = x;
    }
}

void launch_permute_tiled(const PermTileArgs &pa, unsigned batch, hipStream_t s)
{
    const size_t lds = (size_t)((1u << pa.b) + 1) * (1u << pa.a) * sizeof(uint32_t);
    const dim3 grid(1u << (pa.logL - pa.a - pa.b), batch);
    switch (pa.a + pa.b) {
    case 10: hipLaunchKernelGGL(permute_tiled_kernel<10>, grid, dim3(256), lds, s, pa); break;
    case 11: hipLaunchKernelGGL(permute_tiled_kernel<11>, grid, dim3(256), lds, s, pa); break;
    case 12: hipLaunchKernelGGL(permute_tiled_kernel<12>, grid, dim3(256), lds, s, pa); break;
    case 13: hipLaunchKernelGGL(permute_tiled_kernel<13>, grid, dim3(256), lds, s, pa); break;
    default: note_unsupported_launch("permute_tiled_kernel: no instantiation for tile bits =", pa.a + pa.b); break;   // make_perm_tile only builds these
    }
}

// ------------------------------------------------------------------------------------------------
// CRT kernels.  Replace CRT_parallel's implicit input reduction (main-openmp.cc:426-427) and
// CRT_combine_parallel (main-openmp.cc:468-485).  The reference sums r_k * inv_k * (M/m_k) mod M
// with __uint128_t `%` (software division); on the GPU the same canonical value is obtained by
// Garner's mixed-radix form with 32-bit Shoup multiplications followed by a Horner evaluation
// mod p64 with 64-bit Montgomery multiplications -- no 128-bit division, no MFMA.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) crt_reduce_kernel(const uint64_t *__restrict__ x,
                                                         uint32_t *__restrict__ out, size_t n,
                                                         const CrtPrime c)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = x[i];
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    // v = hi * 2^32 + lo
    const uint32_t h = csub(Barrett32::mul_tw(hi, c.two32, c.mp), c.mp.p);
    const uint32_t l = red32(lo, c);
    out[i] = csub(h + l, c.mp.p);
}

// Both operands modulo all four primes in one launch (ntt_crt_multiply: a single product is bound by
// the number of launches): blockIdx.y = operand, out[k] = residues mod m_k at out + k * stride.
__global__ void __launch_bounds__(256) crt_reduce_all_kernel(const uint64_t *__restrict__ a,
                                                             const uint64_t *__restrict__ b,
                                                             uint32_t *__restrict__ ra, uint32_t *__restrict__ rb,
                                                             size_t n, size_t stride, const CrtConsts cc)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = blockIdx.y ? b[i] : a[i];
    uint32_t *__restrict__ out = blockIdx.y ? rb : ra;
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const CrtPrime &c = cc.prime[k];
        const uint32_t h = csub(Barrett32::mul_tw(hi, c.two32, c.mp), c.mp.p);
        out[k * stride + i] = csub(h + red32(lo, c), c.mp.p);
    }
}

void launch_crt_reduce_all(const uint64_t *a, const uint64_t *b, uint32_t *ra, uint32_t *rb, size_t n,
                           size_t stride, const CrtConsts &cc, hipStream_t s)
{
    hipLaunchKernelGGL(crt_reduce_all_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, s, a, b, ra, rb,
                       n, stride, cc);
}

void launch_crt_reduce(const uint64_t *x, uint32_t *out, size_t n, const CrtPrime &c, hipStream_t s)
{
    hipLaunchKernelGGL(crt_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n, c);
}

__global__ void __launch_bounds__(256) crt_merge_kernel(const uint32_t *__restrict__ r0,
                                                        const uint32_t *__restrict__ r1,
                                                        const uint32_t *__restrict__ r2,
                                                        const uint32_t *__restrict__ r3,
                                                        uint64_t *__restrict__ out, size_t n,
                                                        const CrtConsts cc)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // the public ntt_crt_merge takes residues from the caller: one reduction each makes any u32 acceptable (the fast
    // Garner steps of crt_device.hpp assume r_k < m_k; the fused kernel's residues are canonical by construction)
    out[i] = crt_merge_value(red32(r0[i], cc.prime[0]), red32(r1[i], cc.prime[1]), red32(r2[i], cc.prime[2]),
                             red32(r3[i], cc.prime[3]), cc);
}

void launch_crt_merge(const uint32_t *const res[4], uint64_t *out, size_t n, const CrtConsts &cc,
                      hipStream_t s)
{
    hipLaunchKernelGGL(crt_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, res[0],
                       res[1], res[2], res[3], out, n, cc);
}

}  // namespace ntt
