// kernels.hpp -- launch interface between the host engine (engine.cpp) and the gfx950 kernels
// (kernels_*.hip).  Plain structs and function declarations; no HIP types besides hipStream_t.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

#include "modarith.hpp"

namespace ntt {

constexpr int kRowLog = 12;              // LDS-resident row transform: 4096 points
constexpr int kRowLen = 1 << kRowLog;
constexpr int kMaxColLog = 8;            // column level: up to 64 points in registers, 128 / 256 in the LDS-staged kernel

// w_64^j, j < 32 (forward) or w_64^-j (inverse), passed by value so they live in SGPRs.
template <class MM>
struct Roots64 {
    typename MM::tw_t w[32];
};

// One column pass of the 4-step decomposition (see DESIGN.md "Index maps").
template <class MM>
struct ColArgs {
    const uint32_t *in[2];               // blockIdx.z selects the operand (a / b)
    uint32_t *out[2];
    size_t in_stride, out_stride;        // elements between consecutive batch rows
    uint32_t in_len, out_len;            // valid elements per row: loads beyond in_len read 0,
                                         // stores beyond out_len are dropped
    int logM, logS;                      // block size M = N1 * S
    const typename MM::tw_t *tw;         // [N1][S] 4-step twiddles w_M^(+-i2*k1) (row 0 unused unless scale0)
    typename MM::tw_t scale0;            // inverse: multiplier of row k1 = 0 (carries L^-1), if use_scale0
    int use_scale0;
    int canon;                           // 1: final pass, store canonical residues
    Roots64<MM> roots;
    typename MM::tw_t tw_in[256];        // col16 kernels: inner twiddles w_N1^(+-r*ka), index r*16 + ka, r < N1/16
    // negacyclic products (x^n + 1): [L] table psi^(+-i); a forward pass multiplies its inputs by it,
    // the final inverse pass its outputs.  nullptr for the ordinary (cyclic) transform.
    const typename MM::tw_t *twist;
    // first forward level of a residue product of the CRT path (col_kernel<..., U64IN>): `in` holds the caller's u64
    // coefficients, reduced modulo this plan's prime as they are loaded (hi * 2^32 + lo with these two Shoup constants:
    // 1 and 2^32 mod p) -- the implicit reduction of CRT_parallel's toMontgomery, main-openmp.cc:426-427
    Barrett32::tw_t in_one, in_two32;
    // col_kernel only: > 0 = batch rows in total, a workgroup takes rows blockIdx.y, blockIdx.y + gridDim.y, ...
    // (0: one row per workgroup); set by launch_col, see there
    uint32_t loop_batch;
};

template <class MM>
struct RowArgs {
    const uint32_t *a, *b;               // [rows][4096]; b unused by the single-operand kernels
    uint32_t *out;
    const typename MM::tw_t *twA, *twB;          // forward [16][256], [16][16]
    const typename MM::tw_t *itwA, *itwB;        // inverse
    Roots64<MM> roots, iroots;
    // Montgomery plans outside the wide range: the forward passes of row_mul_kernel run their last two stages together
    // with the twiddles (reg_ntt16_tw) and read the entries of outputs 4 ... 7, 12 ... 15 from these instead:
    // [8][256] and [8][16] pairs { centred entry, centred rotated entry (times i, resp. -i) } of the q-th such output
    // (engine.cpp: rotated_table).  nullptr otherwise.
    const uint2 *twA2, *twB2;
    // the same for the inverse passes, whose twiddles move in front of the exchange they used to follow: itwB2 [8][16]
    // as twB2; itwA2 [16][8][16], entry (b, q, i) = the pair of w^-(b * (16 k_q + i)) (engine.cpp: rotated_table_inv)
    const uint2 *itwA2, *itwB2;
};

// A whole product of transform length L = 4096 in the row kernel (no column level): zero padding
// (loads at or beyond in_len read 0), optional negacyclic twist, scale and output cut included.
template <class MM>
struct Row4096Args {
    RowArgs<MM> row;                     // tables and roots; row.a / row.b / row.out = batch bases
    size_t in_stride, out_stride;
    uint32_t in_len, out_len;
    typename MM::tw_t scale;             // L^-1 (times R for Montgomery's pointwise product)
    const typename MM::tw_t *twist, *itwist;     // negacyclic: psi^i, and psi^-i with `scale` folded in
    int fold;                            // 1: store c[i] = ab[i] - ab[i + 2048], i < 2048 (x^2048 + 1 from the linear product)
};

// A whole product of L = 8192 or 16384 in one kernel, rows one after the other (kernels_fused.hip, seq_product_kernel):
// Row4096Args (r.twist unused: the negacyclic twist is part of tw_in; r.itwist is the plain psi^-i table; r.scale unused:
// the scale sits in tw_inv) plus the two column-level tables.
template <class MM>
struct SeqArgs {
    Row4096Args<MM> r;
    // [N1][NIN * 4096], entry (k1, i) = (psi * w_L^k1)^i (psi = 1 for the cyclic transform): row k1 of the forward
    // column level, 4-step twiddle and negacyclic twist included, is the dot product of the NIN inputs
    // i = i1 * 4096 + i2 with these; NIN = N1 / 2 for zero-padded (linear) operands, N1 otherwise
    const typename MM::tw_t *tw_in;
    const typename MM::tw_t *tw_inv;     // [N1][4096]: level-0 inverse twiddles, scale folded in (row 0 = the scale)
    int plain_row0;                      // 1: row 0 of tw_in is all ones (cyclic transform): the inputs are added up
};

template <class MM>
struct SmallArgs {
    const uint32_t *a, *b;
    uint32_t *out;
    size_t in_stride, out_stride;
    uint32_t in_len, out_len;
    int logL;
    const typename MM::tw_t *w, *iw;     // w_L^j and w_L^-j, j < L/2
    typename MM::tw_t scale;             // L^-1 (times R for Montgomery's pointwise product)
    int mode;                            // 0 multiply, 1 forward (natural out), 2 inverse (natural in)
    // negacyclic multiply (mode 0 only): psi^i, and psi^-i with `scale` folded in; nullptr otherwise
    const typename MM::tw_t *twist, *itwist;
};

// One of the four CRT primes: Barrett constants plus the Shoup forms of 1 and 2^32 mod m.
struct CrtPrime {
    ModParams mp;
    Barrett32::tw_t one, two32;
};
struct CrtConsts {
    CrtPrime prime[4];
    Barrett32::tw_t inv[4][4];           // inv[i][j] = m_i^-1 mod m_j (i < j), Shoup form mod m_j
    uint64_t p64, nprime64;              // target modulus (odd, < 2^63), -p64^-1 mod 2^64
    uint64_t mR[4];                      // m_k * 2^64 mod p64
    // fast merge (crt_device.hpp): off[i][j] = the multiple of m_j just above m_i - 1 (a Garner step then needs no
    // reduction of the digit it subtracts); c64[k] = m_0 ... m_(k-1) * 2^64 mod p64 (c64[0] = 2^64 mod p64): the
    // mixed-radix digits are summed as one 96-bit integer and reduced by two 32-bit Montgomery steps.
    // fast = 1 when p64 >= 2^33 and every offset keeps its step below 2^32 (true for the reference's primes)
    uint32_t off[4][4];
    uint64_t c64[4];
    uint32_t fast;
};

// A launch_* dispatcher that has no kernel for the shape it is asked for launches nothing and records it
// here (sticky, per thread); the entry points turn it into NTT_ERR_INTERNAL beside hipGetLastError
// (engine.cpp: take_launch_error), so a skipped launch can never return NTT_OK over stale work buffers.
void note_unsupported_launch(const char *what, int value);

// kind: 0 Montgomery, 1 Barrett; lazy: 0 strict [0, p); 1 values kept in [0, 2p) (needs 4p <= 2^32); 2 (row kernels and
// the forward register column pass; everything else treats it as 1) = 1 with the wide-range transforms of
// device_ntt.hpp (needs 8p <= 2^32)
void launch_col(int kind, int lazy, int logN1, bool inverse, bool half_in, const void *args,
                const ModParams &mp, unsigned cols_per_row, unsigned batch, unsigned nops,
                hipStream_t s);
// the zero-padded first forward level reading u64 operands (Montgomery kernels, 32 or 64 points): false if no such kernel
bool launch_col_u64in(bool lazy, int logN1, const void *args, const ModParams &mp, unsigned cols_per_row, unsigned batch,
                      unsigned nops, hipStream_t s);
void launch_col16(int kind, bool lazy, int logN1, bool inverse, bool half_in, const void *args,
                  const ModParams &mp, unsigned cols_per_row, unsigned batch, unsigned nops,
                  hipStream_t s);
void launch_row_mul(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s);
// the same launch with the kernel's own start / stop events (hipExtLaunchKernelGGL): kernel timing without the
// extra queue packets of hipEventRecord around it
void launch_row_mul_timed(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s,
                          hipEvent_t start, hipEvent_t stop);
// L = 4096 exactly: the whole product of one pair in the row kernel (args: Row4096Args<MM>)
void launch_row_product(int kind, int lazy, const void *args, const ModParams &mp, unsigned batch, hipStream_t s);
void launch_row_fwd(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s);
void launch_row_inv(int kind, int lazy, const void *args, const ModParams &mp, unsigned rows, hipStream_t s);
void launch_small(int kind, bool lazy, const void *args, const ModParams &mp, unsigned batch, hipStream_t s);

// dst[row][perm(i)] = src[row][i] (to_natural) or dst[row][i] = src[row][perm(i)]; perm is the
// mixed-radix digit reversal defined by the column levels (logs, outermost first) + the row.
struct PermArgs {
    const uint32_t *src;
    uint32_t *dst;
    int logL;
    int nlevels;
    int level_log[8];
    int to_natural;
    uint32_t canon_p;                    // != 0: values are in [0, 2p), store canonical residues
};
void launch_permute(const PermArgs &pa, unsigned batch, hipStream_t s);

// Tiled form of the same permutation: dst[R(i)] = src[i], R = reversal of the ORDER of the bit fields
// of i (each field keeps its bits).  A workgroup moves a tile of 2^(a+b) elements through LDS: the low
// b bits of i (one contiguous input run per value of the rest) times the a bits of i that R sends to
// the low a bits of the output index (one contiguous output run per value of the rest).  Bit moves are
// given as runs {from, width, to}: x' |= ((x >> from) & (2^width - 1)) << to.
struct BitRun {
    int from, width, to;
};
struct PermTileArgs {
    const uint32_t *src;
    uint32_t *dst;
    int logL, a, b;
    int n_rev, n_top, n_mid;
    BitRun rev[8];                       // i -> R(i)
    BitRun top[8];                       // tile row index (= low a bits of R(i)) -> its bits in i
    BitRun mid[12];                      // block index -> the bits of i outside the tile
    uint32_t canon_p;                    // != 0: values are in [0, 2p), store canonical residues
};
void launch_permute_tiled(const PermTileArgs &pa, unsigned batch, hipStream_t s);

// The last inverse column pass of the four residue products + the merge in one kernel (kernels_crt.hip).
// One of these per prime: the fields col16_body reads from ColArgs, with only the 64 inner twiddles that a level of
// 32 or 64 points uses, so that four of them fit the kernel-argument segment (L = 2^17 and 2^18; other shapes
// keep the separate kernels)
struct CrtInvPrime {
    const uint32_t *in[1];               // this prime's row-pass output, [batch][L]
    size_t in_stride;
    uint32_t in_len;
    int logM, logS;
    const uint32_t *tw;                  // [N1][S] inverse 4-step twiddles (Montgomery form)
    uint32_t scale0;                     // multiplier of row 0 (carries L^-1 R when the plan has one column level)
    int use_scale0;
    Roots64<Mont32> roots;               // inverse roots
    uint32_t tw_in[64];
    const uint32_t *twist;               // always nullptr here (linear products)
};
struct CrtInvArgs {
    CrtInvPrime prime[4];
    ModParams mp[4];
    CrtConsts cc;
    uint64_t *out;                       // [batch][out_stride] u64
    size_t out_stride;
    uint32_t out_len;                    // 2n - 1
};
bool launch_crt_colinv_merge(int logN1, const CrtInvArgs &a, unsigned cols_per_row, unsigned batch, hipStream_t s);
void launch_crt_reduce(const uint64_t *x, uint32_t *out, size_t n, const CrtPrime &c, hipStream_t s);
void launch_crt_reduce_all(const uint64_t *a, const uint64_t *b, uint32_t *ra, uint32_t *rb, size_t n,
                           size_t stride, const CrtConsts &cc, hipStream_t s);
void launch_crt_merge(const uint32_t *const res[4], uint64_t *out, size_t n, const CrtConsts &cc,
                      hipStream_t s);

// ---- one transform split over the devices of a multi-device context (kernels_dist.hip) --------------------
constexpr int kMaxDistWorld = 16;        // G-point column transform in registers: G = 2, 4, 8, 16
template <class MM>
struct DistColArgs {
    const uint32_t *in[kMaxDistWorld];   // in[i1][j]: point i1 of column j of this slot's piece (a peer's array, or local staging)
    uint32_t *out[kMaxDistWorld];        // out[k1][j]: row k1 goes to slot k1 (its array, or a local send array)
    const typename MM::tw_t *tw;         // [G][cols]: w_L^(+-(h * cols + j) * k1); the inverse table carries G^-1
    typename MM::tw_t scale0;            // inverse: multiplier of row 0 (G^-1)
    int use_scale0;
    uint32_t cols;                       // columns of this slot's piece = S / G (row length of tw)
    uint32_t col0, ncols;                // this launch: columns [col0, col0 + ncols), ncols a multiple of 256
    int half;                            // forward only: rows >= G / 2 are zero padding (not read; first butterfly stage skipped)
    Roots64<MM> roots;                   // forward or inverse w_64 powers (the G-th root is a stride of them)
};
bool launch_dist_col(int kind, bool lazy, int logG, bool inverse, const void *args, const ModParams &mp, hipStream_t s);
// a[i] = a[i] * b[i] mod p for two spectra of one transform (scrambled order; values in the engine's lazy range):
// the pointwise product of main-montgomery.cc:341-344 between the halves of a split product
void launch_dist_pointwise(int kind, bool lazy, uint32_t *a, const uint32_t *b, size_t count, const ModParams &mp, hipStream_t s);

// ---- direct 64-bit path (kernels_u64.hip): one NTT-friendly prime p < 2^62, Montgomery R = 2^64 ---------
struct ModParams64 {
    uint64_t p, p2, nprime;              // nprime = -p^-1 mod 2^64
    uint64_t r1, r2;                     // 2^64 mod p, 2^128 mod p (host side: table construction)
    double pd, pd2, pinv;                // F64 policies: p, 2p and 1/p as doubles
    uint32_t fp;                         // 0: Montgomery lanes; 1: F64 (p < 2^51), 2: F64L (p < 2^50) -- tables hold doubles
};
struct Roots16x64 {
    uint64_t w[8];                       // w_16^j (forward) or w_16^-j (inverse), Montgomery form
};
struct Col64Args {
    const uint64_t *in[2];
    uint64_t *out[2];
    size_t in_stride, out_stride;
    uint32_t in_len, out_len;
    int logM, logS;
    const uint64_t *tw;                  // [N1][S] 4-step twiddles, Montgomery form
    uint64_t scale0;
    int use_scale0;
    int canon;                           // output is the caller's: canonical u64 residues
    int ext_in;                          // input is the caller's (u64 residues < p), not an intermediate
    Roots16x64 roots;
    const uint64_t *twist;               // negacyclic: [L] psi^(+-i), Montgomery form; nullptr otherwise
    uint64_t tw_in[128];                 // LDS-staged levels of 32 / 64 / 128 points: w_N1^(+-r*ka), index r*16 + ka
};
struct Lds64Args {
    const uint64_t *a, *b;
    uint64_t *out;
    size_t in_stride, out_stride;        // elements between batch rows
    uint32_t in_len, out_len;            // valid elements per batch row
    uint32_t blocks_per_row;
    int logLen;                          // block length 2^logLen <= 4096
    const uint64_t *w, *iw;              // w_len^j, w_len^-j, j < len/2, Montgomery form
    uint64_t scale;
    int use_scale, canon, mode;          // mode 0 multiply, 1 forward, 2 inverse
    int ext_in;                          // input is the caller's (u64 residues < p)
    const uint64_t *twist, *itwist;      // negacyclic multiply: psi^i, and psi^-i with the scale folded in
};
struct Row64Args {
    const uint64_t *a, *b;               // [rows][4096]
    uint64_t *out;
    const uint64_t *twA, *twB;           // forward [16][256], [16][16], Montgomery form
    const uint64_t *itwA, *itwB;         // inverse
    Roots16x64 roots, iroots;
};
// A whole 64-bit product of transform length L = 4096 in the row kernel: zero padding, negacyclic twist, scale
// and output cut included (the 64-bit counterpart of Row4096Args)
struct Row64ProdArgs {
    Row64Args row;                       // tables and roots; row.a / row.b / row.out = batch bases
    size_t in_stride, out_stride;
    uint32_t in_len, out_len;
    uint64_t scale;                      // L^-1 (times R on the Montgomery lanes), table form
    const uint64_t *twist, *itwist;      // negacyclic: psi^i, and psi^-i with the scale folded in
    int fold;                            // 1: store c[i] = ab[i] - ab[i + 2048], i < 2048
};
void launch_row64_product(const Row64ProdArgs &a, const ModParams64 &mp, unsigned batch, hipStream_t s);
// logN1 in {1, 2}; half: the operands' upper half is zero padding (linear product)
void launch_seq_product(int kind, bool lazy, int logN1, bool half, const void *args, const ModParams &mp,
                        unsigned batch, hipStream_t s);
void launch_row64_mul(const Row64Args &a, const ModParams64 &mp, unsigned rows, hipStream_t s);
void launch_col64(int logN1, bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp,
                  unsigned cols_per_row, unsigned batch, unsigned nops, hipStream_t s);
// LDS-staged column level of N1 = 32 or 64 points (logN1 = 5, 6): 16 points per thread, one exchange
void launch_col64x(int logN1, bool inverse, bool half_in, const Col64Args &a, const ModParams64 &mp,
                   unsigned cols_per_row, unsigned batch, unsigned nops, hipStream_t s);
void launch_lds64(const Lds64Args &a, const ModParams64 &mp, unsigned blocks, hipStream_t s);

}  // namespace ntt
