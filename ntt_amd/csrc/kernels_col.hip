// kernels_col.hip -- column passes of the 4-step (Bailey) decomposition, gfx950.
//
// One thread owns one column of a block viewed as an N1 x S matrix (N1 <= 64): it loads the N1
// points (stride S; across the 64 lanes of a wave every load/store instruction touches 256
// contiguous bytes), transforms them entirely in VGPRs with SGPR twiddles, applies the 4-step
// twiddle w_M^(i2*k1) from the HBM table and stores.  No LDS, no barrier.
//
// Replaces the large-stride stages (len > 4096) of NTT_iterative, main-montgomery.cc:260-283,
// which sweep the whole array once per stage; here log2(N1) stages cost one read and one write.
//
// Forward: x -> NTT_N1 over i1 -> times w_M^(i2*k1) -> row k1.            (zero padding: loads
//          beyond in_len return 0; HALF skips the upper half entirely)
// Inverse: row k1 times w_M^(-i2*k1) [row 0 times scale0] -> inverse NTT_N1 -> row i1; the
//          outermost inverse pass stores canonical residues and drops indices >= out_len.
#include <cstdlib>
#include "device_ntt.hpp"

namespace ntt {

// one thread, one column of batch row `by`
template <class MM, bool LAZY, int LOGN1, bool INVERSE, bool HALF, bool U64IN, bool WIDE = false>
__device__ __forceinline__ void col_body(const ColArgs<MM> &a, const ModParams &m, const uint32_t by)
{
    static_assert(!WIDE || !INVERSE, "the wide-range transform serves the forward passes");
    using O = Ops<MM, LAZY>;
    using tw_t = typename MM::tw_t;
    constexpr int N1 = 1 << LOGN1;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;          // column id within the row
    const uint32_t S = 1u << a.logS;
    const uint32_t i2 = c & (S - 1);
    const uint32_t base = ((c >> a.logS) << a.logM) + i2;        // blk * M + i2
    const int op = blockIdx.z;
    // descriptors of this batch row: the range check gives zero padding (in_len) and the output cut
    const bufrsrc_t rin = make_buf(a.in[op] + (size_t)by * a.in_stride, a.in_len * 4u);
    const bufrsrc_t rout = make_buf(a.out[op] + (size_t)by * a.out_stride, a.out_len * 4u);
    const bufrsrc_t rtw = make_buf(a.tw, (uint32_t)sizeof(tw_t) << a.logM);
    const uint32_t voff = base * 4u;
    const uint32_t twoff = i2 * (uint32_t)sizeof(tw_t);
    const int rowsh = a.logS + 2, twsh = a.logS + tw_shift<tw_t>();

    uint32_t x[N1];
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1) {
        if (HALF && i1 >= N1 / 2) x[i1] = 0;
        else if constexpr (U64IN) {
            // the caller's u64 coefficient (beyond in_len: 0), reduced modulo p on the way in: hi * 2^32 + lo
            const bufrsrc_t rin64 = make_buf(reinterpret_cast<const uint64_t *>(a.in[op]) + (size_t)by * a.in_stride, a.in_len * 8u);
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rin64, (int)(2u * (voff + ((uint32_t)i1 << rowsh))), 0, NTT_NT_AUX);
            // hi * 2^32 + lo = REDC(hi * R^2 + lo * R): two multiply-adds into one accumulator and one reduction (6
            // instructions; two Shoup products, their reductions and a modular add took 13).  The sum is below 2^33 p and
            // with the reduction's q p below 3 * 2^32 p < 2^64 for p < 2^32 / 3 (launch_col_u64in checks it: every CRT
            // prime is), the result below 3p
            const uint32_t r = Mont32::redc_lazy((uint64_t)v[1] * m.r2 + (uint64_t)v[0] * m.r1, m);
            x[i1] = LAZY ? csub(r, m.p2) : csub(csub(r, m.p2), m.p);
        } else x[i1] = (!INVERSE && HALF) ? buf_load_nt(rin, voff + ((uint32_t)i1 << rowsh), 0)     // the caller's operand
                                          : buf_load(rin, voff + ((uint32_t)i1 << rowsh), 0);
    }
    if (!INVERSE && a.twist) {                                    // negacyclic: x[i] * psi^i  (uniform branch)
        const bufrsrc_t rtwist = make_buf(a.twist, (uint32_t)sizeof(tw_t) << a.logM);
        const uint32_t toff = base * (uint32_t)sizeof(tw_t);
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1)
            if (!(HALF && i1 >= N1 / 2)) x[i1] = O::mul_tw(x[i1], buf_load_tw<tw_t>(rtwist, toff, (uint32_t)i1 << twsh), m);
    }
    if constexpr (INVERSE) {
        // inputs may be loose: every row is multiplied, except row 0 of an outer level
        x[0] = a.use_scale0 ? O::mul_tw(x[0], a.scale0, m) : O::tighten(x[0], m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1)
            x[k1] = O::mul_tw(x[k1], buf_load_tw<tw_t>(rtw, twoff, (uint32_t)k1 << twsh), m);
    }
    reg_ntt<MM, LAZY, LOGN1, HALF, true, WIDE>(x, a.roots, m);    // loose outputs (WIDE: below 8p)
    if constexpr (!INVERSE) {
        x[0] = WIDE ? tighten_wide(x[0], m) : O::tighten(x[0], m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1)
            x[k1] = O::mul_tw(x[k1], buf_load_tw<tw_t>(rtw, twoff, (uint32_t)k1 << twsh), m);
    }
    // inverse: non-final levels hand loose values to the next level; the final one canonicalises.
    // Stores beyond out_len are dropped by the range check.  (a.canon is uniform: a branch, not a
    // select per value.)
    if (INVERSE && a.canon && a.twist) {                          // negacyclic: result * psi^-i, canonical
        const bufrsrc_t rtwist = make_buf(a.twist, (uint32_t)sizeof(tw_t) << a.logM);
        const uint32_t toff = base * (uint32_t)sizeof(tw_t);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1)
            buf_store_nt(O::canon(O::mul_tw(x[k1], buf_load_tw<tw_t>(rtwist, toff, (uint32_t)k1 << twsh), m), m), rout,
                         voff + ((uint32_t)k1 << rowsh), 0);
    } else if (a.canon) {
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1)
            buf_store_nt(O::canon(INVERSE ? O::tighten(x[k1], m) : x[k1], m), rout, voff + ((uint32_t)k1 << rowsh), 0);
    } else {
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) buf_store(x[k1], rout, voff + ((uint32_t)k1 << rowsh), 0);
    }
}

// A launch covers the batch in grid.y; a WALKING launch (a.loop_batch > 0, see launch_col) has fewer workgroups than
// batch rows and each takes rows blockIdx.y, blockIdx.y + gridDim.y, ...: what does not depend on the row -- the
// column's 4-step twiddles above all -- is then loaded once per workgroup.
template <class MM, bool LAZY, int LOGN1, bool INVERSE, bool HALF, bool U64IN = false, bool WIDE = false>
__global__ void __launch_bounds__(256) col_kernel(const ColArgs<MM> a, const ModParams m)
{
    uint32_t by = blockIdx.y;
    do {
        col_body<MM, LAZY, LOGN1, INVERSE, HALF, U64IN, WIDE>(a, m, by);
        by += gridDim.y;
    } while (by < a.loop_batch);
}

// The walking launch of the zero-padded first forward level with SOFTWARE PIPELINING across the walk: the N1 / 2 operand
// loads of the workgroup's NEXT product are issued before the current product's column is transformed, so that they
// are in flight under its butterflies and stores (two waves per SIMD cannot hide them otherwise: the kernel sat at
// 0.117 VALU instructions per SIMD-cycle).  The column's 4-step twiddles stay in registers for the walk.
// Plain cyclic transform only (no twist, intermediate output): the multiply path's first pass.
template <class MM, bool LAZY, int LOGN1, bool WIDE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) col_walk_prefetch_kernel(const ColArgs<MM> a, const ModParams m)
{
    using O = Ops<MM, LAZY>;
    using tw_t = typename MM::tw_t;
    constexpr int N1 = 1 << LOGN1, H = N1 / 2;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    const uint32_t S = 1u << a.logS;
    const uint32_t i2 = c & (S - 1);
    const uint32_t base = ((c >> a.logS) << a.logM) + i2;
    const int op = blockIdx.z;
    const bufrsrc_t rtw = make_buf(a.tw, (uint32_t)sizeof(tw_t) << a.logM);
    const uint32_t voff = base * 4u;
    const uint32_t twoff = i2 * (uint32_t)sizeof(tw_t);
    const int rowsh = a.logS + 2, twsh = a.logS + tw_shift<tw_t>();
    tw_t tw[N1];
#pragma unroll
    for (int k1 = 1; k1 < N1; ++k1) tw[k1] = buf_load_tw<tw_t>(rtw, twoff, (uint32_t)k1 << twsh);
    uint32_t by = blockIdx.y;
    uint32_t nx[H];
    {
        const bufrsrc_t rin = make_buf(a.in[op] + (size_t)by * a.in_stride, a.in_len * 4u);
#pragma unroll
        for (int i1 = 0; i1 < H; ++i1) nx[i1] = buf_load_nt(rin, voff + ((uint32_t)i1 << rowsh), 0);
    }
    for (;;) {
        uint32_t x[N1];
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) x[i1] = i1 < H ? nx[i1] : 0u;
        const uint32_t nby = by + gridDim.y;
        const bool more = nby < a.loop_batch;                         // uniform
        // an opaque copy of the column offset per iteration: otherwise the 64 row offsets are hoisted out of the loop
        // and held in registers for the whole walk (with them the kernel needs more than 256 registers: one wave per SIMD)
        uint32_t vo = voff;
        asm volatile("" : "+v"(vo));
        if (more) {
            const bufrsrc_t rin = make_buf(a.in[op] + (size_t)nby * a.in_stride, a.in_len * 4u);
#pragma unroll
            for (int i1 = 0; i1 < H; ++i1) nx[i1] = buf_load_nt(rin, vo + ((uint32_t)i1 << rowsh), 0);
        }
        __builtin_amdgcn_sched_barrier(0);                            // the loads stay in front of the arithmetic
        reg_ntt<MM, LAZY, LOGN1, true, true, WIDE>(x, a.roots, m);
        x[0] = WIDE ? tighten_wide(x[0], m) : O::tighten(x[0], m);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) x[k1] = O::mul_tw(x[k1], tw[k1], m);
        const bufrsrc_t rout = make_buf(a.out[op] + (size_t)by * a.out_stride, a.out_len * 4u);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) buf_store(x[k1], rout, vo + ((uint32_t)k1 << rowsh), 0);
        if (!more) break;
        by = nby;
    }
}

// NTT_AMD_COL_PREFETCH=0 keeps the plain walking kernel (read per launch, so that a test can compare both in one process):
// 0.775 -> 0.707 us per product for the pass alone, 537 -> 544 k products/s end to end (profiles/ab_prefetch_r03.txt)
static bool walk_prefetch()
{
    const char *e = std::getenv("NTT_AMD_COL_PREFETCH");
    return !(e && e[0] == '0');
}

template <class MM, bool LAZY, int LOGN1, bool WIDE = false>
static void launch_col_t(bool inverse, bool half_in, const ColArgs<MM> &a, const ModParams &mp,
                         dim3 grid, hipStream_t s)
{
    // Montgomery only: with Barrett's 8-byte twiddles the column's table alone takes 126 registers and the pipelined
    // kernel spills (148 VGPRs to scratch; 392 k -> 261 k products/s at the headline shape)
    if constexpr (LOGN1 >= 5 && MM::kMontgomery) {
        if (!inverse && half_in && a.loop_batch && !a.twist && !a.canon && walk_prefetch()) {
            hipLaunchKernelGGL((col_walk_prefetch_kernel<MM, LAZY, LOGN1, WIDE>), grid, dim3(256), 0, s, a, mp);
            return;
        }
    }
    if (inverse)
        hipLaunchKernelGGL((col_kernel<MM, LAZY, LOGN1, true, false>), grid, dim3(256), 0, s, a, mp);
    else if (half_in)
        hipLaunchKernelGGL((col_kernel<MM, LAZY, LOGN1, false, true, false, WIDE>), grid, dim3(256), 0, s, a, mp);
    else
        hipLaunchKernelGGL((col_kernel<MM, LAZY, LOGN1, false, false, false, WIDE>), grid, dim3(256), 0, s, a, mp);
}

template <class MM, bool LAZY, bool WIDE = false>
static void launch_col_l(int logN1, bool inverse, bool half_in, const ColArgs<MM> &a,
                         const ModParams &mp, dim3 grid, hipStream_t s)
{
    switch (logN1) {
    case 1: launch_col_t<MM, LAZY, 1>(inverse, half_in, a, mp, grid, s); break;
    case 2: launch_col_t<MM, LAZY, 2>(inverse, half_in, a, mp, grid, s); break;
    case 3: launch_col_t<MM, LAZY, 3>(inverse, half_in, a, mp, grid, s); break;
    case 4: launch_col_t<MM, LAZY, 4, WIDE>(inverse, half_in, a, mp, grid, s); break;      // the wide-range transform pays
    case 5: launch_col_t<MM, LAZY, 5, WIDE>(inverse, half_in, a, mp, grid, s); break;      // from 16 points up
    case 6: launch_col_t<MM, LAZY, 6, WIDE>(inverse, half_in, a, mp, grid, s); break;
    default: note_unsupported_launch("col_kernel: no instantiation for log2(N1) =", logN1); break;
    }
}

// products per column tile and operand in a walking forward launch (0: never walk)
static unsigned walk_cap()
{
    static const unsigned cap = [] { const char *e = std::getenv("NTT_AMD_COL_GRIDY"); return e ? (unsigned)std::atoi(e) : 16u; }();
    return cap;
}

bool launch_col_u64in(bool lazy, int logN1, const void *args, const ModParams &mp, unsigned cols_per_row, unsigned batch,
                      unsigned nops, hipStream_t s)
{
    if (mp.p >= 0x55555555u) return false;                // the input reduction of col_body needs 3 * 2^32 p < 2^64
    dim3 grid(cols_per_row / 256, batch, nops);
    auto a = *static_cast<const ColArgs<Mont32> *>(args);
    const unsigned cap = walk_cap();
    a.loop_batch = cap && batch > cap ? batch : 0;
    if (a.loop_batch) grid.y = cap;
    if (logN1 == 6 && lazy) hipLaunchKernelGGL((col_kernel<Mont32, true, 6, false, true, true>), grid, dim3(256), 0, s, a, mp);
    else if (logN1 == 6) hipLaunchKernelGGL((col_kernel<Mont32, false, 6, false, true, true>), grid, dim3(256), 0, s, a, mp);
    else if (logN1 == 5 && lazy) hipLaunchKernelGGL((col_kernel<Mont32, true, 5, false, true, true>), grid, dim3(256), 0, s, a, mp);
    else if (logN1 == 5) hipLaunchKernelGGL((col_kernel<Mont32, false, 5, false, true, true>), grid, dim3(256), 0, s, a, mp);
    else return false;
    return true;
}

void launch_col(int kind, int lazy, int logN1, bool inverse, bool half_in, const void *args,
                const ModParams &mp, unsigned cols_per_row, unsigned batch, unsigned nops,
                hipStream_t s)
{
    dim3 grid(cols_per_row / 256, batch, nops);
    // Forward passes of 32 / 64 points walk the batch: 16 workgroups per column tile and operand, each taking every
    // 16th product with its columns' 4-step twiddles loaded once (the compiler hoists the 31 / 63 table loads out
    // of the loop) -- 537 -> 547 k products/s at the headline shape, profiles/ab_gridy_r02.txt; fewer or more
    // workgroups, the 16-point levels (RNS shape) and the LDS-staged kernel do not gain.  NTT_AMD_COL_GRIDY=0: off.
    const unsigned cap = walk_cap();
    const bool walk = cap && batch > cap && !inverse && logN1 >= 5;
    if (walk) grid.y = cap;
    if (kind == 0) {
        auto a = *static_cast<const ColArgs<Mont32> *>(args);
        a.loop_batch = walk ? batch : 0;
        if (lazy == 2 && !inverse) launch_col_l<Mont32, true, true>(logN1, inverse, half_in, a, mp, grid, s);
        else if (lazy) launch_col_l<Mont32, true>(logN1, inverse, half_in, a, mp, grid, s);
        else launch_col_l<Mont32, false>(logN1, inverse, half_in, a, mp, grid, s);
    } else {
        auto a = *static_cast<const ColArgs<Barrett32> *>(args);
        a.loop_batch = walk ? batch : 0;
        if (lazy == 2 && !inverse) launch_col_l<Barrett32, true, true>(logN1, inverse, half_in, a, mp, grid, s);
        else if (lazy) launch_col_l<Barrett32, true>(logN1, inverse, half_in, a, mp, grid, s);
        else launch_col_l<Barrett32, false>(logN1, inverse, half_in, a, mp, grid, s);
    }
}

}  // namespace ntt
