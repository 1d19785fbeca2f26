// col16_body.hpp -- device body of the LDS-staged column pass (see kernels_col16.hip).
#pragma once
#include <type_traits>

#include "device_ntt.hpp"

namespace ntt {

// Body of the LDS-staged column pass for one tile: `bx` = tile index within the row (blockIdx.x
// of the stand-alone kernel), `by` = batch row, `op` = operand.  `lds` holds N1 * C words (4096; 8192 / 16384 for N1 = 128 / 256).
// `Args` is ColArgs<MM> or any struct with the same field names (the CRT kernel passes a slimmer one);
// `sink(idx, row, v)` receives output `idx` = j * R + kr of this thread (a compile-time constant after
// unrolling), which is row `row` of the thread's column: the stand-alone kernel stores it, the fused CRT kernel
// keeps it in registers.  `rout` / the element offset of (row, column) are the sink's business.
template <class MM, bool LAZY, int LOGR, bool INVERSE, bool HALF, class Args, class Sink>
__device__ __forceinline__ void col16_body_sink(const Args &a, const ModParams &m, uint32_t *lds,
                                                const uint32_t bx, const uint32_t by, const int op, const bool canon,
                                                Sink &&sink)
{
    using O = Ops<MM, LAZY>;
    using tw_t = typename MM::tw_t;
    constexpr int R = 1 << LOGR;             // 2, 4, 8 or 16: N1 = 32, 64, 128, 256 points
    constexpr int LOGT = LOGR <= 2 ? 8 : LOGR + 6;   // 256 threads; 512 / 1024 for N1 = 128 / 256 (64 columns keep r wave-uniform)
    constexpr int LOGC = LOGT - LOGR;        // columns per workgroup
    constexpr int C = 1 << LOGC;
    constexpr int KA = 16 / R;               // ka values per thread in pass 2
    const uint32_t tid = threadIdx.x;
    const uint32_t cl = tid & (C - 1);
    const int r = __builtin_amdgcn_readfirstlane(tid >> LOGC);     // wave-uniform
    const uint32_t c = bx * C + cl;                        // column id within the row
    const uint32_t S = 1u << a.logS;
    const uint32_t i2 = c & (S - 1);
    const uint32_t base = ((c >> a.logS) << a.logM) + i2;          // blk * M + i2
    // descriptors of this batch row: the range check gives zero padding (in_len) and the output cut
    const bufrsrc_t rin = make_buf(a.in[op] + (size_t)by * a.in_stride, a.in_len * 4u);
    const bufrsrc_t rtw = make_buf(a.tw, (uint32_t)sizeof(tw_t) << a.logM);
    const uint32_t voff = base * 4u;
    const uint32_t twoff = i2 * (uint32_t)sizeof(tw_t);
    const int rowsh = a.logS + 2, twsh = a.logS + tw_shift<tw_t>();

    // ---- pass 1: rows R*a + r ----------------------------------------------------------------
    uint32_t x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (HALF && j >= 8) {
            x[j] = 0;
        } else {
            const uint32_t row = (uint32_t)(R * j + r);                // wave-uniform
            // HALF marks the first forward level of a linear product: the caller's operand, read once
            uint32_t v = HALF ? buf_load_nt(rin, voff + (row << rowsh), 0) : buf_load(rin, voff + (row << rowsh), 0);
            if (!INVERSE && a.twist)                                   // negacyclic: x[i] * psi^i (uniform)
                v = O::mul_tw(v, buf_load_tw<tw_t>(make_buf(a.twist, (uint32_t)sizeof(tw_t) << a.logM),
                                                   base * (uint32_t)sizeof(tw_t), row << twsh), m);
            if constexpr (INVERSE) {
                // inputs may be loose: every row is multiplied, except row 0 of an outer level
                if (row == 0) v = a.use_scale0 ? O::mul_tw(v, a.scale0, m) : O::tighten(v, m);
                else v = O::mul_tw(v, buf_load_tw<tw_t>(rtw, twoff, row << twsh), m);
            }
            x[j] = v;
        }
    }
    reg_ntt<MM, LAZY, 4, HALF, true>(x, a.roots, m);              // loose outputs
    x[0] = O::tighten(x[0], m);
    if (r != 0) {                                                  // wave-uniform branch
#pragma unroll
        for (int ka = 1; ka < 16; ++ka) x[ka] = O::mul_tw(x[ka], a.tw_in[r * 16 + ka], m);
    } else {
#pragma unroll
        for (int ka = 1; ka < 16; ++ka) x[ka] = O::tighten(x[ka], m);
    }
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) lds[(r * 16 + ka) * C + cl] = x[ka];
    __syncthreads();

    // ---- pass 2: radix-R over r for ka in [q*KA, (q+1)*KA) ---------------------------------------
    // non-final inverse levels hand loose values to the next level; a final pass canonicalises
    // (a.canon is uniform: a branch around the whole pass, not a select per value); stores beyond
    // out_len are dropped by the range check
    const int q = r;
    auto pass2 = [&](auto canon_tag) {
        constexpr bool CANON = decltype(canon_tag)::value;
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            const int ka = q * KA + j;
            uint32_t y[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) y[rr] = lds[(rr * 16 + ka) * C + cl];
            reg_ntt<MM, LAZY, LOGR, false, true>(y, a.roots, m);  // loose outputs
#pragma unroll
            for (int kr = 0; kr < R; ++kr) {
                const uint32_t row = (uint32_t)(ka + 16 * kr);         // wave-uniform
                uint32_t v = y[kr];
                if constexpr (!INVERSE) {
                    v = row != 0 ? O::mul_tw(v, buf_load_tw<tw_t>(rtw, twoff, row << twsh), m) : O::tighten(v, m);
                    if constexpr (CANON) v = O::canon(v, m);
                } else if constexpr (CANON) {
                    if (a.twist)                                       // negacyclic: result * psi^-i (uniform)
                        v = O::canon(O::mul_tw(v, buf_load_tw<tw_t>(make_buf(a.twist, (uint32_t)sizeof(tw_t) << a.logM),
                                                                    base * (uint32_t)sizeof(tw_t), row << twsh), m), m);
                    else
                        v = O::canon(O::tighten(v, m), m);
                }
                sink(std::integral_constant<bool, CANON>{}, j * R + kr, voff + (row << rowsh), v);
            }
        }
    };
    if (canon) pass2(std::true_type{});
    else pass2(std::false_type{});
}

// The stand-alone form: every output is stored (streaming stores for the canonical final inverse pass: the
// caller's result is never read back).
template <class MM, bool LAZY, int LOGR, bool INVERSE, bool HALF>
__device__ __forceinline__ void col16_body(const ColArgs<MM> &a, const ModParams &m, uint32_t *lds,
                                           const uint32_t bx, const uint32_t by, const int op)
{
    const bufrsrc_t rout = make_buf(a.out[op] + (size_t)by * a.out_stride, a.out_len * 4u);
    col16_body_sink<MM, LAZY, LOGR, INVERSE, HALF>(a, m, lds, bx, by, op, a.canon != 0,
        [&](auto canon_tag, int, uint32_t byte_off, uint32_t v) {
            if constexpr (decltype(canon_tag)::value && INVERSE) buf_store_nt(v, rout, byte_off, 0);
            else buf_store(v, rout, byte_off, 0);
        });
}


}  // namespace ntt
