// kernels_row.hip -- LDS-staged 4096-point row transforms and the fused multiply kernel, gfx950.
//
// One workgroup (256 threads = 4 waves, 16 points per thread) owns one contiguous 4096-point
// row.  A row transform is three radix-16 register passes (4096 = 16*16*16) with two exchanges
// through a 17 KiB LDS tile; both exchange layouts are padded so that every ds_read_b32 /
// ds_write_b32 is bank-conflict free:
//     exchange 1: P1(i) = i + 16*(i >> 8)   written  k*256 + t,        read  b*256 + j*16 + i
//     exchange 2: P2(i) = i + (i >> 4)      written  b*256 + k*16 + i, read  t*16 + j
// The forward transform leaves the spectrum in digit-reversed order (position k1*256+k2*16+k3
// holds natural index k1 + 16*k2 + 256*k3); the inverse consumes exactly that order, so no
// bit-reversal pass (main-montgomery.cc:238-253) exists anywhere.
//
// row_mul_kernel fuses: forward(a row), forward(b row), pointwise product
// (main-montgomery.cc:341-344) and the inverse row transform -- the 12 small-stride stages of
// three NTT_iterative calls plus the pointwise loop in one read of a, b and one write.
#include <hip/hip_ext.h>

#include "row_body.hpp"

namespace ntt {

#if defined(NTT_ROW_WAVES)                         // experiment builds: cap the registers for this many waves per SIMD
#define NTT_ROW_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NTT_ROW_WAVES, NTT_ROW_WAVES)))
#else
#define NTT_ROW_WAVES_ATTR
#endif
template <class MM, bool LAZY, bool WIDE = false>
__global__ void __launch_bounds__(256) NTT_ROW_WAVES_ATTR row_mul_kernel(const RowArgs<MM> a, const ModParams m)
{
    __shared__ uint32_t lds[kLdsWords];
    row_mul_body<MM, LAZY, WIDE>(a, m, lds, blockIdx.x);
}

// Whole product of one pair when L = 4096: the row kernel with the padding, twist, scale and output cut
// that the column passes do for longer transforms (3-4x faster than radix-2 stages in LDS, small_kernel).
template <class MM, bool LAZY, bool WIDE = false>
__global__ void __launch_bounds__(256) row_product_kernel(const Row4096Args<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t t = threadIdx.x;
    const uint32_t *__restrict__ pa = a.row.a + (size_t)blockIdx.x * a.in_stride;
    const uint32_t *__restrict__ pb = a.row.b + (size_t)blockIdx.x * a.in_stride;
    uint32_t xa[16], xb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = t + 256 * j;
        xa[j] = i < a.in_len ? pa[i] : 0u;
        xb[j] = i < a.in_len ? pb[i] : 0u;
    }
    if (a.twist) {                                            // negacyclic: operands times psi^i (uniform)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const typename MM::tw_t w = a.twist[t + 256 * j];
            xa[j] = O::mul_tw(xa[j], w, m);
            xb[j] = O::mul_tw(xb[j], w, m);
        }
    }
    row_forward<MM, LAZY, WIDE, row_fuse<MM, WIDE>()>(xa, lds, t, a.row, m);
    row_forward<MM, LAZY, WIDE, row_fuse<MM, WIDE>()>(xb, lds, t, a.row, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = O::mul(xa[k], xb[k], m);
    row_inverse<MM, LAZY, WIDE, row_fuse_inv<MM, WIDE>()>(xa, lds, t, a.row, m);        // loose outputs, unscaled
    uint32_t *__restrict__ po = a.row.out + (size_t)blockIdx.x * a.out_stride;
    if (a.fold) {                                             // x^2048 + 1: position i + 2048 sits in the same thread (uniform)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = t + 256 * j;
            const uint32_t lo = csub(MM::mul_tw(xa[j], a.scale, m), m.p), hi = csub(MM::mul_tw(xa[j + 8], a.scale, m), m.p);
            const uint32_t d = lo - hi;
            if (i < a.out_len) po[i] = lo < hi ? d + m.p : d;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = t + 256 * j;
        const typename MM::tw_t w = a.itwist ? a.itwist[i] : a.scale;
        if (i < a.out_len) po[i] = csub(MM::mul_tw(xa[j], w, m), m.p);
    }
}

void launch_row_product(int kind, int lazy, const void *args, const ModParams &mp, unsigned batch, hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const Row4096Args<Mont32> *>(args);
        if (lazy == 2) hipLaunchKernelGGL((row_product_kernel<Mont32, true, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else if (lazy) hipLaunchKernelGGL((row_product_kernel<Mont32, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else hipLaunchKernelGGL((row_product_kernel<Mont32, false>), dim3(batch), dim3(256), 0, s, a, mp);
    } else {
        const auto &a = *static_cast<const Row4096Args<Barrett32> *>(args);
        if (lazy == 2) hipLaunchKernelGGL((row_product_kernel<Barrett32, true, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else if (lazy) hipLaunchKernelGGL((row_product_kernel<Barrett32, true>), dim3(batch), dim3(256), 0, s, a, mp);
        else hipLaunchKernelGGL((row_product_kernel<Barrett32, false>), dim3(batch), dim3(256), 0, s, a, mp);
    }
}

// Forward only: natural row in, digit-reversed spectrum out (position 16*t + k).
template <class MM, bool LAZY, bool WIDE = false>
__global__ void __launch_bounds__(256) row_fwd_kernel(const RowArgs<MM> a, const ModParams m)
{
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t t = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kRowLen;
    const uint32_t *__restrict__ pa = a.a + row;
    uint32_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = pa[t + 256 * j];
    row_forward<MM, LAZY, WIDE, row_fuse<MM, WIDE>()>(x, lds, t, a, m);
    uint4 *__restrict__ po = reinterpret_cast<uint4 *>(a.out + row + 16 * t);
#pragma unroll
    for (int q = 0; q < 4; ++q) po[q] = make_uint4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// Inverse only: digit-reversed spectrum in, natural row out (unscaled; the column pass or the
// caller's scale twiddle finishes it).
template <class MM, bool LAZY, bool WIDE = false>
__global__ void __launch_bounds__(256) row_inv_kernel(const RowArgs<MM> a, const ModParams m)
{
    __shared__ uint32_t lds[kLdsWords];
    const uint32_t t = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kRowLen;
    const uint4 *__restrict__ pa = reinterpret_cast<const uint4 *>(a.a + row + 16 * t);
    uint32_t x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = pa[q];
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    row_inverse<MM, LAZY, WIDE, row_fuse_inv<MM, WIDE>()>(x, lds, t, a, m);
    uint32_t *__restrict__ po = a.out + row;
#pragma unroll
    for (int j = 0; j < 16; ++j) po[t + 256 * j] = x[j];
}

template <int WHICH, class MM, bool LAZY, bool WIDE = false>
static void launch_row_t(const RowArgs<MM> &a, const ModParams &mp, unsigned rows, hipStream_t s)
{
    if (WHICH == 0) hipLaunchKernelGGL((row_mul_kernel<MM, LAZY, WIDE>), dim3(rows), dim3(256), 0, s, a, mp);
    if (WHICH == 1) hipLaunchKernelGGL((row_fwd_kernel<MM, LAZY, WIDE>), dim3(rows), dim3(256), 0, s, a, mp);
    if (WHICH == 2) hipLaunchKernelGGL((row_inv_kernel<MM, LAZY, WIDE>), dim3(rows), dim3(256), 0, s, a, mp);
}

// lazy: 0 strict [0, p), 1 lazy [0, 2p), 2 lazy with the wide-range transforms (p < 2^29)
template <int WHICH>
static void launch_row_any(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows,
                           hipStream_t s)
{
    if (kind == 0) {
        const auto &a = *static_cast<const RowArgs<Mont32> *>(args);
        if (lazy == 2) launch_row_t<WHICH, Mont32, true, true>(a, mp, rows, s);
        else if (lazy) launch_row_t<WHICH, Mont32, true>(a, mp, rows, s);
        else launch_row_t<WHICH, Mont32, false>(a, mp, rows, s);
    } else {
        const auto &a = *static_cast<const RowArgs<Barrett32> *>(args);
        if (lazy == 2) launch_row_t<WHICH, Barrett32, true, true>(a, mp, rows, s);
        else if (lazy) launch_row_t<WHICH, Barrett32, true>(a, mp, rows, s);
        else launch_row_t<WHICH, Barrett32, false>(a, mp, rows, s);
    }
}

template <class MM>
static void launch_row_mul_ext(int lazy, const RowArgs<MM> &a, const ModParams &mp, unsigned rows, hipStream_t s,
                               hipEvent_t e0, hipEvent_t e1)
{
    if (lazy == 2) hipExtLaunchKernelGGL((row_mul_kernel<MM, true, true>), dim3(rows), dim3(256), 0, s, e0, e1, 0, a, mp);
    else if (lazy) hipExtLaunchKernelGGL((row_mul_kernel<MM, true>), dim3(rows), dim3(256), 0, s, e0, e1, 0, a, mp);
    else hipExtLaunchKernelGGL((row_mul_kernel<MM, false>), dim3(rows), dim3(256), 0, s, e0, e1, 0, a, mp);
}

void launch_row_mul_timed(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s,
                          hipEvent_t start, hipEvent_t stop)
{
    if (kind == 0) launch_row_mul_ext<Mont32>(lazy, *static_cast<const RowArgs<Mont32> *>(args), mp, rows, s, start, stop);
    else launch_row_mul_ext<Barrett32>(lazy, *static_cast<const RowArgs<Barrett32> *>(args), mp, rows, s, start, stop);
}

void launch_row_mul(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s)
{
    launch_row_any<0>(kind, lazy, args, mp, rows, s);
}
void launch_row_fwd(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s)
{
    launch_row_any<1>(kind, lazy, args, mp, rows, s);
}
void launch_row_inv(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s)
{
    launch_row_any<2>(kind, lazy, args, mp, rows, s);
}

}  // namespace ntt
