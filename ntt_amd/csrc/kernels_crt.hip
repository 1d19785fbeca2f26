// kernels_crt.hip -- the last inverse column pass of the four residue products of a large-modulus product and the CRT
// merge in ONE kernel, gfx950.
//
// The reference merges after the four per-prime products are complete (CRT_combine_parallel, main-openmp.cc:468-485,
// called at :710 behind the four CRT_parallel calls of :699-708).  As separate kernels that costs, per coefficient, a
// 4-byte store and a 4-byte load for each prime between the inverse column pass that produces a residue and the merge
// that consumes it (8 MiB per n = 2^17 product).  Here a workgroup runs the LDS-staged inverse column pass
// (col16_body.hpp) of the same tile for the four primes one after the other, keeps its 16 canonical residues per prime
// in registers, and then merges: Garner + Horner (crt_device.hpp), one u64 store per coefficient.
#include "col16_body.hpp"
#include "crt_device.hpp"

namespace ntt {

template <int LOGR>
__global__ void __launch_bounds__(256) crt_colinv_merge_kernel(const CrtInvArgs a)
{
    static_assert(LOGR == 1 || LOGR == 2, "levels of 32 or 64 points");
    __shared__ uint32_t lds[4096];
    constexpr int R = 1 << LOGR, KA = 16 / R;
    uint32_t res[4][16];
    uint32_t off[16];                                      // element index of output idx within the product (the same for every prime)
    auto run = [&](auto lazy_tag, const int k) {
        constexpr bool LAZY = decltype(lazy_tag)::value;
        col16_body_sink<Mont32, LAZY, LOGR, true, false>(a.prime[k], a.mp[k], lds, blockIdx.x, blockIdx.y, 0, true,
            [&](auto, int idx, uint32_t byte_off, uint32_t v) {
                res[k][idx] = v;
                off[idx] = byte_off >> 2;
            });
        __syncthreads();                                   // the tile is rewritten by the next prime's first pass
    };
    // 1004535809, 469762049 and 998244353 are below 2^30 (values kept in [0, 2p)); 1224736769 is not: strict [0, p)
    run(std::true_type{}, 0);
    run(std::false_type{}, 1);
    run(std::true_type{}, 2);
    run(std::true_type{}, 3);
    const bufrsrc_t rout = make_buf(a.out + (size_t)blockIdx.y * a.out_stride, a.out_len * 8u);
    // the route is chosen OUTSIDE the unrolled loop: with both bodies inside, the loop is no longer unrolled and the
    // residue arrays are indexed dynamically, i.e. live in scratch memory
    auto merge_all = [&](auto fast_tag) {
#pragma unroll
        for (int idx = 0; idx < KA * R; ++idx) {
            const uint64_t v = decltype(fast_tag)::value ? crt_merge_fast(res[0][idx], res[1][idx], res[2][idx], res[3][idx], a.cc)
                                                         : crt_merge_slow(res[0][idx], res[1][idx], res[2][idx], res[3][idx], a.cc);
            using u32x2 = uint32_t __attribute__((ext_vector_type(2)));
            u32x2 w;
            w[0] = (uint32_t)v;
            w[1] = (uint32_t)(v >> 32);
            __builtin_amdgcn_raw_buffer_store_b64(w, rout, (int)(off[idx] * 8u), 0, NTT_NT_AUX);   // beyond out_len: dropped
        }
    };
    if (a.cc.fast) merge_all(std::true_type{});
    else merge_all(std::false_type{});
}

bool launch_crt_colinv_merge(int logN1, const CrtInvArgs &a, unsigned cols_per_row, unsigned batch, hipStream_t s)
{
    switch (logN1) {
    case 5: hipLaunchKernelGGL(crt_colinv_merge_kernel<1>, dim3(cols_per_row / 128, batch), dim3(256), 0, s, a); return true;
    case 6: hipLaunchKernelGGL(crt_colinv_merge_kernel<2>, dim3(cols_per_row / 64, batch), dim3(256), 0, s, a); return true;
    default: note_unsupported_launch("crt_colinv_merge_kernel: no instantiation for log2(N1) =", logN1); return false;
    }
}

}  // namespace ntt
