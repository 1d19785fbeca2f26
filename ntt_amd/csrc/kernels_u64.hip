// kernels_u64.hip -- direct path for ONE NTT-friendly prime p < 2^62: 64-bit Montgomery lanes, gfx950.
//
// The reference serves a modulus above 32 bits in two ways: the 4-prime CRT (main-openmp.cc:693-711,
// here crt.cpp) and, for p_ < 2^50, the single-modulus branch main-openmp.cc:678-692 with 64-bit
// lanes and `montgomery m(R = 2^60, p_)` (:650-651; its `power` limits it to 32-bit moduli in
// practice).  These kernels are that single-modulus branch for any NTT-friendly prime below 2^62:
// R = 2^64, one u64 per residue, lazy range [0, 2p).
//
//   REDC(T) for T = x * wR < 2^64 * p:  q = T_lo * (-p^-1) mod 2^64,  (T + q*p) / 2^64 in [0, 2p)
//   = T_hi + mulhi(q, p) + (T_lo != 0)        (the low words cancel to exactly 0 or 2^64)
//
// Same decomposition as the 32-bit path: column levels of N1 <= 16 points in
// registers with the 4-step twiddle from an HBM table, rows of 4096 points as three radix-16
// register passes around a 34 KiB LDS tile (row64_mul_kernel).  Transforms of L <= 4096 run in
// lds64_kernel: radix-2 in LDS, decimation in frequency forward (natural in, bit-reversed out),
// decimation in time inverse (bit-reversed in, natural out), pointwise product in between.  Neither
// needs a reordering pass.
//
// Two arithmetic policies run the same kernels (template parameter A):
//   M64  the Montgomery lanes above, any p < 2^62;
//   F64  for p < 2^51: residues as exact integers in double-precision lanes, |x| <= p.  x * w mod p is
//        h = x*w (rounded), l = fma(x, w, -h) (its exact error), q = rint(h / p), r = fma(-q, p, h) + l.
//        With |x| <= 2p, |w| <= p: |x*w/p| < 2^52, the three roundings behind q (1/p, h, h * 1/p)
//        move it by < 1.5, rint by 0.5, so |x*w - q*p| < 2p < 2^52: the fma result and the sum are exact
//        integers; one fold (r - copysign(p, r)) brings |r| back to <= p.  Sums and differences of two
//        values stay below 2^52, exact as well.  Eight full-rate FP64 instructions per multiplication
//        against eleven 32x32 multiply-adds plus carries, three per modular add against seven.
//        Tables hold plain residues as doubles (no Montgomery factor).  Operands must be < p.
#include "kernels.hpp"

namespace ntt {

struct M64 {
    using T = uint64_t;
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ T ext_in(uint64_t v, const ModParams64 &) { return v; }     // caller's residue
    static __device__ __forceinline__ T raw(uint64_t bits) { return bits; }                       // table entry / intermediate
    static __device__ __forceinline__ uint64_t bits(T v) { return v; }
    static __device__ __forceinline__ uint64_t canon_bits(T v, const ModParams64 &m) { return csub(v, m.p); }
    static __device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m)
    {
        const uint64_t d = x - m;
        return d < x ? d : x;               // x >= m  <=>  no wrap
    }
    // x * w for any u64 x, wR = w * 2^64 mod p (< p): result in [0, 2p)
    static __device__ __forceinline__ uint64_t mulw(uint64_t x, uint64_t wR, const ModParams64 &m)
    {
        // 128-bit product from four 32x32->64 multiply-adds (x * wR and __umul64hi(x, wR) written
        // separately cost seven), q from three, the high half of q * p from four
        const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), wl = (uint32_t)wR, wh = (uint32_t)(wR >> 32);
        const uint64_t t0 = (uint64_t)xl * wl;
        const uint64_t t1 = (uint64_t)xl * wh + (t0 >> 32);
        const uint64_t t2 = (uint64_t)xh * wl + (uint32_t)t1;
        const uint64_t hi = (uint64_t)xh * wh + (t1 >> 32) + (t2 >> 32);
        const uint32_t lo0 = (uint32_t)t0, lo1 = (uint32_t)t2;
        const uint32_t nl = (uint32_t)m.nprime, nh = (uint32_t)(m.nprime >> 32);
        const uint64_t q0 = (uint64_t)lo0 * nl;
        const uint32_t ql = (uint32_t)q0, qh = (uint32_t)(q0 >> 32) + lo0 * nh + lo1 * nl;
        const uint32_t pl = (uint32_t)m.p, ph = (uint32_t)(m.p >> 32);
        const uint64_t u0 = (uint64_t)ql * pl;
        const uint64_t u1 = (uint64_t)ql * ph + (u0 >> 32);
        const uint64_t u2 = (uint64_t)qh * pl + (uint32_t)u1;
        const uint64_t h2 = (uint64_t)qh * ph + (u1 >> 32) + (u2 >> 32);
        return hi + h2 + ((lo0 | lo1) != 0 ? 1u : 0u);
    }
    static __device__ __forceinline__ uint64_t add(uint64_t u, uint64_t v, const ModParams64 &m)
    {
        return csub(u + v, m.p2);
    }
    static __device__ __forceinline__ uint64_t sub(uint64_t u, uint64_t v, const ModParams64 &m)
    {
        return csub(u - v + m.p2, m.p2);
    }
    static __device__ __forceinline__ uint64_t sub_raw(uint64_t u, uint64_t v, const ModParams64 &m)
    {
        return u - v + m.p2;                // (0, 4p): valid mulw operand
    }
};

struct F64 {
    using T = double;
    static __device__ __forceinline__ T zero() { return 0.0; }
    // u64 < 2^52 <-> double through the 2^52 offset (exponent bits 0x433)
    static __device__ __forceinline__ T ext_in(uint64_t v, const ModParams64 &)
    {
        return __builtin_bit_cast(double, v | 0x4330000000000000ull) - 0x1p52;
    }
    static __device__ __forceinline__ T raw(uint64_t bits) { return __builtin_bit_cast(double, bits); }
    static __device__ __forceinline__ uint64_t bits(T v) { return __builtin_bit_cast(uint64_t, v); }
    static __device__ __forceinline__ uint64_t canon_bits(T v, const ModParams64 &m)
    {
        double t = v < 0.0 ? v + m.pd : v;          // [0, p]
        t = t >= m.pd ? t - m.pd : t;               // [0, p)
        return __builtin_bit_cast(uint64_t, t + 0x1p52) & 0x000fffffffffffffull;
    }
    static __device__ __forceinline__ T fold(T s, const ModParams64 &m)      // |s| <= 2p  ->  |s| <= p
    {
        return s - __builtin_copysign(m.pd, s);
    }
    static __device__ __forceinline__ T mulw(T x, T w, const ModParams64 &m)  // |x| <= 2p, |w| <= p
    {
        const double h = x * w;
        const double l = __builtin_fma(x, w, -h);
        const double q = __builtin_rint(h * m.pinv);
        const double r = __builtin_fma(-q, m.pd, h) + l;
        return fold(r, m);
    }
    static __device__ __forceinline__ T add(T u, T v, const ModParams64 &m) { return fold(u + v, m); }
    static __device__ __forceinline__ T sub(T u, T v, const ModParams64 &m) { return fold(u - v, m); }
    static __device__ __forceinline__ T sub_raw(T u, T v, const ModParams64 &) { return u - v; }
};

// F64 for p < 2^50: one bit of headroom more lets residues live in |x| <= 2p and the multiplication go
// without its closing fold: for |x| <= 4p, |w| <= p (twiddles) or |x|, |w| <= 2p (pointwise) the quotient is
// off by < 4p * p / p * 3 * 2^-53 + 0.5 <= 2, so |r| <= 2p already, and |h - q*p| < 2^53 keeps the fma exact.
// Six instructions per multiplication instead of eight, ten per butterfly instead of twelve.
struct F64L : F64 {
    static __device__ __forceinline__ uint64_t canon_bits(T v, const ModParams64 &m)
    {
        double t = v < 0.0 ? v + m.pd2 : v;         // [0, 2p]
        t = t >= m.pd ? t - m.pd : t;               // [0, p]
        t = t >= m.pd ? t - m.pd : t;               // [0, p)
        return __builtin_bit_cast(uint64_t, t + 0x1p52) & 0x000fffffffffffffull;
    }
    static __device__ __forceinline__ T fold(T s, const ModParams64 &m)      // |s| <= 4p  ->  |s| <= 2p
    {
        return s - __builtin_copysign(m.pd2, s);
    }
    static __device__ __forceinline__ T mulw(T x, T w, const ModParams64 &m)
    {
        const double h = x * w;
        const double l = __builtin_fma(x, w, -h);
        const double q = __builtin_rint(h * m.pinv);
        return __builtin_fma(-q, m.pd, h) + l;
    }
    static __device__ __forceinline__ T add(T u, T v, const ModParams64 &m) { return fold(u + v, m); }
    static __device__ __forceinline__ T sub(T u, T v, const ModParams64 &m) { return fold(u - v, m); }
};

// the arithmetic policy of a plan: 0 Montgomery lanes, 1 F64 (p < 2^51), 2 F64L (p < 2^50)
#define NTT_U64_DISPATCH(mp, CALL_M64, CALL_F64, CALL_F64L) \
    do {                                                     \
        if ((mp).fp == 2) { CALL_F64L; }                     \
        else if ((mp).fp == 1) { CALL_F64; }                 \
        else { CALL_M64; }                                   \
    } while (0)

__host__ __device__ constexpr int bitrev64_c(int x, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// N-point transform in registers, N = 2^LOGN <= 16, natural in / natural out, values in [0, 2p).
// roots.w[j] = w_16^j (forward or inverse table), so w_(2*half)^j = w[j * 8 / half].
template <class A, int LOGN, bool HALF>
__device__ __forceinline__ void reg_ntt64(typename A::T (&x)[1 << LOGN], const Roots16x64 &roots, const ModParams64 &m)
{
    using T = typename A::T;
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int s = 0; s < LOGN; ++s) {
        const int half = N >> (s + 1);
        const int step = 8 / half;
#pragma unroll
        for (int blk = 0; blk < N; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const T u = x[blk + j], v = x[blk + j + half];
                if (HALF && s == 0) {
                    x[blk + j + half] = j == 0 ? u : A::mulw(u, A::raw(roots.w[j * step]), m);   // v == 0
                } else {
                    x[blk + j] = A::add(u, v, m);
                    x[blk + j + half] = j == 0 ? A::sub(u, v, m) : A::mulw(A::sub_raw(u, v, m), A::raw(roots.w[j * step]), m);
                }
            }
        }
    }
    T y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[bitrev64_c(i, LOGN)] = x[i];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = y[i];
}

// ---- column pass ---------------------------------------------------------------------------------------
// One thread per column of a block viewed as N1 x S (index maps as kernels_col.hip).
template <class A, int LOGN1, bool INVERSE, bool HALF>
__global__ void __launch_bounds__(256) col64_kernel(const Col64Args a, const ModParams64 m)
{
    using T = typename A::T;
    constexpr int N1 = 1 << LOGN1;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    const uint32_t S = 1u << a.logS;
    const uint32_t i2 = c & (S - 1);
    const uint32_t base = ((c >> a.logS) << a.logM) + i2;
    const int op = blockIdx.z;
    const uint64_t *__restrict__ in = a.in[op] + (size_t)blockIdx.y * a.in_stride;
    uint64_t *__restrict__ out = a.out[op] + (size_t)blockIdx.y * a.out_stride;
    const uint64_t *__restrict__ tw = a.tw + i2;

    T x[N1];
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1) {
        if (HALF && i1 >= N1 / 2) {
            x[i1] = A::zero();
        } else {
            const uint32_t idx = base + ((uint32_t)i1 << a.logS);
            const uint64_t v = idx < a.in_len ? in[idx] : 0ull;
            x[i1] = a.ext_in ? A::ext_in(v, m) : A::raw(v);
        }
    }
    if (!INVERSE && a.twist) {                                // negacyclic: x[i] * psi^i (uniform branch)
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
            if (!(HALF && i1 >= N1 / 2)) x[i1] = A::mulw(x[i1], A::raw(a.twist[base + ((uint32_t)i1 << a.logS)]), m);
    }
    if constexpr (INVERSE) {
        if (a.use_scale0) x[0] = A::mulw(x[0], A::raw(a.scale0), m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) x[k1] = A::mulw(x[k1], A::raw(tw[(size_t)k1 << a.logS]), m);
    }
    reg_ntt64<A, LOGN1, HALF>(x, a.roots, m);
    if constexpr (!INVERSE) {
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) x[k1] = A::mulw(x[k1], A::raw(tw[(size_t)k1 << a.logS]), m);
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        const uint32_t idx = base + ((uint32_t)k1 << a.logS);
        T v = x[k1];
        if (INVERSE && a.canon && a.twist) v = A::mulw(v, A::raw(a.twist[idx]), m);      // negacyclic: result * psi^-i
        if (idx < a.out_len) {
            if (a.canon) __builtin_nontemporal_store(A::canon_bits(v, m), out + idx);          // the caller's: not read back
            else out[idx] = A::bits(v);
        }
    }
}

template <class A, int LOGN1>
static void launch_col64_t(bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp, dim3 grid, hipStream_t s)
{
    if (inverse) hipLaunchKernelGGL((col64_kernel<A, LOGN1, true, false>), grid, dim3(256), 0, s, a, mp);
    else if (half_in) hipLaunchKernelGGL((col64_kernel<A, LOGN1, false, true>), grid, dim3(256), 0, s, a, mp);
    else hipLaunchKernelGGL((col64_kernel<A, LOGN1, false, false>), grid, dim3(256), 0, s, a, mp);
}

template <class A>
static void launch_col64_a(int logN1, bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp, dim3 grid,
                           hipStream_t s)
{
    switch (logN1) {
    case 1: launch_col64_t<A, 1>(inverse, half_in, a, mp, grid, s); break;
    case 2: launch_col64_t<A, 2>(inverse, half_in, a, mp, grid, s); break;
    case 3: launch_col64_t<A, 3>(inverse, half_in, a, mp, grid, s); break;
    case 4: launch_col64_t<A, 4>(inverse, half_in, a, mp, grid, s); break;
    default: note_unsupported_launch("col64_kernel: no instantiation for log2(N1) =", logN1); break;
    }
}

void launch_col64(int logN1, bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp,
                  unsigned cols_per_row, unsigned batch, unsigned nops, hipStream_t s)
{
    const dim3 grid(cols_per_row / 256, batch, nops);
    NTT_U64_DISPATCH(mp, launch_col64_a<M64>(logN1, inverse, half_in, a, mp, grid, s),
                     launch_col64_a<F64>(logN1, inverse, half_in, a, mp, grid, s),
                     launch_col64_a<F64L>(logN1, inverse, half_in, a, mp, grid, s));
}

// ---- LDS-staged column level: N1 = 16 * R points, R = 2 or 4 -------------------------------------------------
// Same scheme as the 32-bit col16 kernel (col16_body.hpp): a workgroup owns C = 256/R columns x N1 rows;
// thread (c, r) transforms rows R*j + r (16 points, registers), multiplies by the inner twiddle
// w_N1^(r*ka), the 32 KiB tile goes through LDS once, then thread (c, q) runs the radix-R transforms for
// its ka and applies the 4-step twiddle.  One pass over memory for 5 or 6 bits of the transform length
// where the register kernel above needs two (64 points of 64 bits do not fit a thread: 404-512 VGPRs).
template <class A, int LOGR, bool INVERSE, bool HALF>
__global__ void __launch_bounds__(LOGR <= 2 ? 256 : 64 << LOGR) col64x_kernel(const Col64Args a, const ModParams64 m)
{
    using T = typename A::T;
    constexpr int R = 1 << LOGR;
    constexpr int LOGT = LOGR <= 2 ? 8 : LOGR + 6;        // 256 threads; 512 for N1 = 128 (64 columns keep r wave-uniform)
    constexpr int LOGC = LOGT - LOGR;
    constexpr int C = 1 << LOGC;
    constexpr int KA = 16 / R;
    __shared__ uint64_t lds[LOGR <= 2 ? 4096 : 1024 << LOGR];
    const uint32_t tid = threadIdx.x;
    const uint32_t cl = tid & (C - 1);
    const int r = __builtin_amdgcn_readfirstlane(tid >> LOGC);        // wave-uniform
    const uint32_t c = blockIdx.x * C + cl;
    const uint32_t S = 1u << a.logS;
    const uint32_t i2 = c & (S - 1);
    const uint32_t base = ((c >> a.logS) << a.logM) + i2;
    const int op = blockIdx.z;
    const uint64_t *__restrict__ in = a.in[op] + (size_t)blockIdx.y * a.in_stride;
    uint64_t *__restrict__ out = a.out[op] + (size_t)blockIdx.y * a.out_stride;
    const uint64_t *__restrict__ tw = a.tw + i2;

    // forward: the 4-step twiddles of this thread's 16 outputs (rows ka + 16 kr, ka = r*KA + j) are fetched
    // up front, in flight together with the data loads, not after the exchange
    uint64_t twv[INVERSE ? 1 : 16];
    if constexpr (!INVERSE) {
#pragma unroll
        for (int j = 0; j < KA; ++j)
#pragma unroll
            for (int kr = 0; kr < R; ++kr)
                twv[j * R + kr] = tw[(size_t)(r * KA + j + 16 * kr) << a.logS];
    }
    T x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (HALF && j >= 8) {
            x[j] = A::zero();
        } else {
            const uint32_t row = (uint32_t)(R * j + r);
            const uint32_t idx = base + (row << a.logS);
            const uint64_t vb = idx < a.in_len ? in[idx] : 0ull;
            T v = a.ext_in ? A::ext_in(vb, m) : A::raw(vb);
            if (!INVERSE && a.twist) v = A::mulw(v, A::raw(a.twist[idx]), m);           // negacyclic: x[i] * psi^i
            if constexpr (INVERSE) {
                if (row == 0) { if (a.use_scale0) v = A::mulw(v, A::raw(a.scale0), m); }
                else v = A::mulw(v, A::raw(tw[(size_t)row << a.logS]), m);
            }
            x[j] = v;
        }
    }
    reg_ntt64<A, 4, HALF>(x, a.roots, m);
    if (r != 0) {
#pragma unroll
        for (int ka = 1; ka < 16; ++ka) x[ka] = A::mulw(x[ka], A::raw(a.tw_in[r * 16 + ka]), m);
    }
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) lds[(r * 16 + ka) * C + cl] = A::bits(x[ka]);
    __syncthreads();
    const int q = r;
#pragma unroll
    for (int j = 0; j < KA; ++j) {
        const int ka = q * KA + j;
        T y[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) y[rr] = A::raw(lds[(rr * 16 + ka) * C + cl]);
        reg_ntt64<A, LOGR, false>(y, a.roots, m);
#pragma unroll
        for (int kr = 0; kr < R; ++kr) {
            const uint32_t row = (uint32_t)(ka + 16 * kr);
            const uint32_t idx = base + (row << a.logS);
            T v = y[kr];
            if constexpr (!INVERSE) {
                if (row != 0) v = A::mulw(v, A::raw(twv[j * R + kr]), m);
            } else {
                if (a.canon && a.twist) v = A::mulw(v, A::raw(a.twist[idx]), m);        // negacyclic: result * psi^-i
            }
            if (idx < a.out_len) {
                if (a.canon) __builtin_nontemporal_store(A::canon_bits(v, m), out + idx);      // the caller's: not read back
                else out[idx] = A::bits(v);
            }
        }
    }
}

template <class A, int LOGR>
static void launch_col64x_t(bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp, dim3 grid, hipStream_t s)
{
    const dim3 block(LOGR <= 2 ? 256 : 64 << LOGR);
    if (inverse) hipLaunchKernelGGL((col64x_kernel<A, LOGR, true, false>), grid, block, 0, s, a, mp);
    else if (half_in) hipLaunchKernelGGL((col64x_kernel<A, LOGR, false, true>), grid, block, 0, s, a, mp);
    else hipLaunchKernelGGL((col64x_kernel<A, LOGR, false, false>), grid, block, 0, s, a, mp);
}

void launch_col64x(int logN1, bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp,
                   unsigned cols_per_row, unsigned batch, unsigned nops, hipStream_t s)
{
    const dim3 grid(cols_per_row / (logN1 == 5 ? 128 : 64), batch, nops);
    if (logN1 == 7) {
        NTT_U64_DISPATCH(mp, (launch_col64x_t<M64, 3>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64, 3>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64L, 3>(inverse, half_in, a, mp, grid, s)));
    } else if (logN1 == 6) {
        NTT_U64_DISPATCH(mp, (launch_col64x_t<M64, 2>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64, 2>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64L, 2>(inverse, half_in, a, mp, grid, s)));
    } else if (logN1 != 5) {
        note_unsupported_launch("col64x_kernel: no instantiation for log2(N1) =", logN1);
    } else {
        NTT_U64_DISPATCH(mp, (launch_col64x_t<M64, 1>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64, 1>(inverse, half_in, a, mp, grid, s)),
                         (launch_col64x_t<F64L, 1>(inverse, half_in, a, mp, grid, s)));
    }
}

// ---- LDS transform of one block of len = 2^logLen <= 4096 points ---------------------------------------
// Forward: DIF, natural in -> bit-reversed out.  w[j] = w_len^j, j < len/2.
template <class A>
__device__ __forceinline__ void lds_forward64(uint64_t *lds, int logLen, const uint64_t *__restrict__ w,
                                              const ModParams64 &m)
{
    const uint32_t nb = 1u << (logLen - 1);             // butterflies per stage
    for (int s = 0; s < logLen; ++s) {
        const int lh = logLen - 1 - s;                  // log2(half)
        for (uint32_t idx = threadIdx.x; idx < nb; idx += 256) {
            const uint32_t j = idx & ((1u << lh) - 1);
            const uint32_t i0 = ((idx >> lh) << (lh + 1)) + j, i1 = i0 + (1u << lh);
            const typename A::T u = A::raw(lds[i0]), v = A::raw(lds[i1]);
            lds[i0] = A::bits(A::add(u, v, m));
            lds[i1] = A::bits(A::mulw(A::sub_raw(u, v, m), A::raw(w[j << s]), m));
        }
        __syncthreads();
    }
}

// Inverse: DIT, bit-reversed in -> natural out, unscaled.  iw[j] = w_len^-j.
template <class A>
__device__ __forceinline__ void lds_inverse64(uint64_t *lds, int logLen, const uint64_t *__restrict__ iw,
                                              const ModParams64 &m)
{
    const uint32_t nb = 1u << (logLen - 1);
    for (int s = logLen - 1; s >= 0; --s) {
        const int lh = logLen - 1 - s;
        for (uint32_t idx = threadIdx.x; idx < nb; idx += 256) {
            const uint32_t j = idx & ((1u << lh) - 1);
            const uint32_t i0 = ((idx >> lh) << (lh + 1)) + j, i1 = i0 + (1u << lh);
            const typename A::T u = A::raw(lds[i0]), v = A::mulw(A::raw(lds[i1]), A::raw(iw[j << s]), m);
            lds[i0] = A::bits(A::add(u, v, m));
            lds[i1] = A::bits(A::sub(u, v, m));
        }
        __syncthreads();
    }
}

// mode 0: multiply (forward a, forward b, pointwise, inverse); 1: forward only; 2: inverse only
// (bit-reversed order on the spectrum side in every mode).  One workgroup per block of len points;
// blocks_per_row blocks make one batch row (1 for the whole-transform case L <= 4096, L/4096 rows
// otherwise).  `scale` (Montgomery form) multiplies the output when use_scale is set.
template <class A>
__global__ void __launch_bounds__(256) lds64_kernel(const Lds64Args a, const ModParams64 m)
{
    using T = typename A::T;
    __shared__ uint64_t lds[4096];
    const uint32_t len = 1u << a.logLen;
    const uint32_t blk = blockIdx.x % a.blocks_per_row, row = blockIdx.x / a.blocks_per_row;
    const size_t off = (size_t)blk << a.logLen;
    const uint64_t *__restrict__ pa = a.a + (size_t)row * a.in_stride + off;
    uint64_t *__restrict__ po = a.out + (size_t)row * a.out_stride + off;
    const uint32_t in_len = a.in_len, out_len = a.out_len;        // per row, relative to `off` = 0 when blocks_per_row == 1

    const bool neg = a.mode == 0 && a.twist != nullptr;       // negacyclic multiply (whole transform in this block)
    for (uint32_t i = threadIdx.x; i < len; i += 256) {
        const uint64_t vb = (off + i) < in_len ? pa[i] : 0ull;
        T v = a.ext_in ? A::ext_in(vb, m) : A::raw(vb);
        if (neg) v = A::mulw(v, A::raw(a.twist[i]), m);
        lds[i] = A::bits(v);
    }
    __syncthreads();
    if (a.mode == 2) {
        lds_inverse64<A>(lds, a.logLen, a.iw, m);
    } else {
        lds_forward64<A>(lds, a.logLen, a.w, m);
        if (a.mode == 0) {
            const uint64_t *__restrict__ pb = a.b + (size_t)row * a.in_stride + off;
            uint64_t ra[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t i = threadIdx.x + 256u * k;
                ra[k] = i < len ? lds[i] : 0ull;
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < len; i += 256) {
                const uint64_t vb = (off + i) < in_len ? pb[i] : 0ull;
                T v = a.ext_in ? A::ext_in(vb, m) : A::raw(vb);
                if (neg) v = A::mulw(v, A::raw(a.twist[i]), m);
                lds[i] = A::bits(v);
            }
            __syncthreads();
            lds_forward64<A>(lds, a.logLen, a.w, m);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t i = threadIdx.x + 256u * k;
                // M64: REDC(a*b), a, b < 2p, a*b < 2^64 * p; the stray R^-1 is folded into the scale
                if (i < len) lds[i] = A::bits(A::mulw(A::raw(ra[k]), A::raw(lds[i]), m));
            }
            __syncthreads();
            lds_inverse64<A>(lds, a.logLen, a.iw, m);
        }
    }
    for (uint32_t i = threadIdx.x; i < len; i += 256) {
        T v = A::raw(lds[i]);
        if (neg) v = A::mulw(v, A::raw(a.itwist[i]), m);            // psi^-i with the scale folded in
        else if (a.use_scale) v = A::mulw(v, A::raw(a.scale), m);
        if ((off + i) < out_len) po[i] = a.canon ? A::canon_bits(v, m) : A::bits(v);
    }
}

// ---- 4096-point rows in registers: 4096 = 16 * 16 * 16 -----------------------------------------------------
// Same three-pass structure and index maps as the 32-bit row kernel (row_body.hpp): thread t of 256 holds
// 16 points, runs a 16-point register transform, multiplies by the pass twiddle and exchanges through a
// padded LDS tile (P1, P2 as there, in units of 8-byte words).  On exit of row64_forward x[k] is the
// spectrum at position 16 t + k of a fixed digit-reversed order, which row64_inverse consumes.
constexpr int kRow64Words = kRowLen + kRowLen / 16;          // 4352 x 8 B = 34 KiB

__device__ __forceinline__ uint32_t Q1(uint32_t i) { return i + ((i >> 8) << 4); }
__device__ __forceinline__ uint32_t Q2(uint32_t i) { return i + (i >> 4); }

template <class A>
__device__ __forceinline__ void row64_forward(typename A::T (&x)[16], uint64_t *lds, const uint32_t t, const Row64Args &a,
                                              const ModParams64 &m)
{
    reg_ntt64<A, 4, false>(x, a.roots, m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = A::mulw(x[k], A::raw(a.twA[k * 256 + t]), m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[Q1(k * 256 + t)] = A::bits(x[k]);
    __syncthreads();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = A::raw(lds[Q1(b * 256 + j * 16 + i)]);
    __syncthreads();
    reg_ntt64<A, 4, false>(x, a.roots, m);
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = A::mulw(x[k], A::raw(a.twB[k * 16 + i]), m);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[Q2(b * 256 + k * 16 + i)] = A::bits(x[k]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = A::raw(lds[Q2(t * 16 + j)]);
    __syncthreads();
    reg_ntt64<A, 4, false>(x, a.roots, m);
}

template <class A>
__device__ __forceinline__ void row64_inverse(typename A::T (&x)[16], uint64_t *lds, const uint32_t t, const Row64Args &a,
                                              const ModParams64 &m)
{
    reg_ntt64<A, 4, false>(x, a.iroots, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[Q2(t * 16 + j)] = A::bits(x[j]);
    __syncthreads();
    const uint32_t b = t >> 4, i = t & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = A::raw(lds[Q2(b * 256 + k * 16 + i)]);
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = A::mulw(x[k], A::raw(a.itwB[k * 16 + i]), m);
    reg_ntt64<A, 4, false>(x, a.iroots, m);
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[Q1(b * 256 + j * 16 + i)] = A::bits(x[j]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = A::raw(lds[Q1(k * 256 + t)]);
    __syncthreads();
#pragma unroll
    for (int k = 1; k < 16; ++k) x[k] = A::mulw(x[k], A::raw(a.itwA[k * 256 + t]), m);
    reg_ntt64<A, 4, false>(x, a.iroots, m);
}

// Fused multiply of one 4096-point row: forward a, forward b, pointwise, inverse (in place on a's row).
template <class A>
__global__ void __launch_bounds__(256) row64_mul_kernel(const Row64Args a, const ModParams64 m)
{
    using T = typename A::T;
    __shared__ uint64_t lds[kRow64Words];
    const uint32_t t = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kRowLen;
    const uint64_t *__restrict__ pa = a.a + row;
    const uint64_t *__restrict__ pb = a.b + row;
    T xa[16], xb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xa[j] = A::raw(pa[t + 256 * j]);
#pragma unroll
    for (int j = 0; j < 16; ++j) xb[j] = A::raw(pb[t + 256 * j]);
    row64_forward<A>(xa, lds, t, a, m);
    row64_forward<A>(xb, lds, t, a, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = A::mulw(xa[k], xb[k], m);      // M64: REDC(a*b); R^-1 folded into the scale
    row64_inverse<A>(xa, lds, t, a, m);
    uint64_t *__restrict__ po = a.out + row;
#pragma unroll
    for (int j = 0; j < 16; ++j) po[t + 256 * j] = A::bits(xa[j]);
}

// Whole product of one pair when L = 4096: the row kernel with the padding, twist, scale and output cut that
// the column passes do for longer transforms (the radix-2 LDS kernel below takes 12 barrier-separated stages
// per transform for the same work).
template <class A>
__global__ void __launch_bounds__(256) row64_product_kernel(const Row64ProdArgs a, const ModParams64 m)
{
    using T = typename A::T;
    __shared__ uint64_t lds[kRow64Words];
    const uint32_t t = threadIdx.x;
    const uint64_t *__restrict__ pa = a.row.a + (size_t)blockIdx.x * a.in_stride;
    const uint64_t *__restrict__ pb = a.row.b + (size_t)blockIdx.x * a.in_stride;
    T xa[16], xb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = t + 256u * j;
        xa[j] = i < a.in_len ? A::ext_in(pa[i], m) : A::zero();
        xb[j] = i < a.in_len ? A::ext_in(pb[i], m) : A::zero();
    }
    if (a.twist) {                                            // negacyclic: operands times psi^i (uniform)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const T w = A::raw(a.twist[t + 256u * j]);
            xa[j] = A::mulw(xa[j], w, m);
            xb[j] = A::mulw(xb[j], w, m);
        }
    }
    row64_forward<A>(xa, lds, t, a.row, m);
    row64_forward<A>(xb, lds, t, a.row, m);
#pragma unroll
    for (int k = 0; k < 16; ++k) xa[k] = A::mulw(xa[k], xb[k], m);
    row64_inverse<A>(xa, lds, t, a.row, m);
    uint64_t *__restrict__ po = a.row.out + (size_t)blockIdx.x * a.out_stride;
    if (a.fold) {                                             // x^2048 + 1: position i + 2048 sits in the same thread (uniform)
        const T w = A::raw(a.scale);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = t + 256u * j;
            const T d = A::sub(A::mulw(xa[j], w, m), A::mulw(xa[j + 8], w, m), m);
            if (i < a.out_len) po[i] = A::canon_bits(d, m);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = t + 256u * j;
        const T w = A::raw(a.itwist ? a.itwist[i] : a.scale);
        if (i < a.out_len) po[i] = A::canon_bits(A::mulw(xa[j], w, m), m);
    }
}

void launch_row64_product(const Row64ProdArgs &a, const ModParams64 &mp, unsigned batch, hipStream_t s)
{
    NTT_U64_DISPATCH(mp, hipLaunchKernelGGL(row64_product_kernel<M64>, dim3(batch), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(row64_product_kernel<F64>, dim3(batch), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(row64_product_kernel<F64L>, dim3(batch), dim3(256), 0, s, a, mp));
}

void launch_row64_mul(const Row64Args &a, const ModParams64 &mp, unsigned rows, hipStream_t s)
{
    NTT_U64_DISPATCH(mp, hipLaunchKernelGGL(row64_mul_kernel<M64>, dim3(rows), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(row64_mul_kernel<F64>, dim3(rows), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(row64_mul_kernel<F64L>, dim3(rows), dim3(256), 0, s, a, mp));
}

void launch_lds64(const Lds64Args &a, const ModParams64 &mp, unsigned blocks, hipStream_t s)
{
    NTT_U64_DISPATCH(mp, hipLaunchKernelGGL(lds64_kernel<M64>, dim3(blocks), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(lds64_kernel<F64>, dim3(blocks), dim3(256), 0, s, a, mp),
                     hipLaunchKernelGGL(lds64_kernel<F64L>, dim3(blocks), dim3(256), 0, s, a, mp));
}

}  // namespace ntt
