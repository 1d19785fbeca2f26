/*
 * ntt_oracle.h -- CPU restatement of the reference's NTT polynomial-multiply path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ntt_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline, never as the product path.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Parity pin: nttdata/{0,1,2,3}.in/.out (tests/golden), and oracle/_ref (the reference's own
 * main-baseline.cc compiled in place) on seeded synthetic inputs.
 * The 4-prime CRT merge is "parity unpinned" by the reference's own data (nttdata/4.* is not
 * shipped); it is pinned against Python big-int schoolbook products instead.
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* power(): main-montgomery.cc:82-95 (int), main-openmp.cc:163-176 (u64, mod < 2^32). */
uint64_t orc_power(uint64_t base, uint64_t exp, uint64_t mod);
/* 128-bit-safe modpow, for 51-bit moduli (main-montgomery-GPU.cu:71-82 uses __uint128_t products). */
uint64_t orc_power128(uint64_t base, uint64_t exp, uint64_t mod);

/* class montgomery: main-montgomery.cc:133-234 (R=2^32), main-openmp.cc:178-358 (R=2^60). */
typedef struct {
    uint64_t N, R, R2, N_inv_neg;
    int logR;
} orc_mont;
int      orc_mont_init(orc_mont *m, uint64_t R, uint64_t N);  /* ctor :136-142; returns -1 if gcd!=1 (:163-166) */
uint64_t orc_mont_redc(const orc_mont *m, __uint128_t T);     /* REDC :170-181 */
uint64_t orc_mont_to(const orc_mont *m, uint64_t x);          /* toMont :184 */
uint64_t orc_mont_from(const orc_mont *m, uint64_t x);        /* fromMont :190 */
uint64_t orc_mont_mul(const orc_mont *m, uint64_t a, uint64_t b); /* ModMul :212 */

/* class Barrett: main-barrett.cc:130-172. */
typedef struct {
    uint64_t mod, factor;
} orc_barrett;
void     orc_barrett_init(orc_barrett *b, uint64_t mod);                 /* :133-138 */
uint64_t orc_barrett_reduce(const orc_barrett *b, __uint128_t x);        /* :145-150 */
uint64_t orc_barrett_mul(const orc_barrett *b, uint64_t x, uint64_t y);  /* :152-155 */

/* bit_reverse: main-montgomery.cc:238-253. */
void orc_bit_reverse(uint64_t *a, int n);

/* NTT_iterative, Montgomery lanes: main-montgomery.cc:255-295.  a[] holds Montgomery-form
 * residues in [0,p); inv = +1 forward, -1 inverse (scaled by n^-1). g = 3. */
void orc_ntt_mont(uint64_t *a, int n, uint64_t p, int inv, const orc_mont *m);
/* Same with the OpenMP block-parallel stage loop of main-openmp.cc:381-422. */
void orc_ntt_mont_omp(uint64_t *a, int n, uint64_t p, int inv, const orc_mont *m, int threads);
/* NTT_iterative, Barrett lanes: main-barrett.cc:193-228.  Plain residues. */
void orc_ntt_barrett(uint64_t *a, int n, uint64_t p, int inv, const orc_barrett *b);
/* Plain-domain forward/inverse NTT (natural in, natural out, g=3), any p < 2^32: convenience
 * wrapper = toMont, orc_ntt_mont, fromMont. */
void orc_ntt_plain(uint32_t *a, int n, uint32_t p, int inv);

/* poly_multiply (schoolbook, accumulating): main-montgomery.cc:71-77. */
void orc_poly_multiply_naive(const int *a, const int *b, int *ab, int n, int p);

/* Harness multiply body, Montgomery: main-montgomery.cc:317-351 (len = next power of FOUR >= 2n,
 * R = 2^32).  Writes ab[0..2n-2]. */
void orc_poly_multiply_mont(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p);
/* Harness multiply body, Barrett: main-barrett.cc:249-282 (len = next power of TWO >= 2n). */
void orc_poly_multiply_barrett(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p);
/* Harness multiply body, OpenMP single-prime branch: main-openmp.cc:678-692 (R = 2^60,
 * len = next power of two >= 2n).  `threads` replaces the hard-coded num_threads(8).
 * If seconds != NULL it receives the time between the reference's Start/End stamps (:675,:712). */
void orc_poly_multiply_omp(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p,
                           int threads, double *seconds);

/* CRT path: mods = {1004535809, 1224736769, 469762049, 998244353}  (main-openmp.cc:654). */
extern const uint64_t orc_crt_mods[4];
typedef struct {
    __uint128_t Mi[4];
    uint64_t inv[4];
    __uint128_t M;
} orc_crt_pre;
void orc_crt_precompute(orc_crt_pre *pre, const uint64_t *mods, int k);   /* main-openmp.cc:449-466 */
/* CRT_parallel: per-prime product polynomial of unreduced (<2^51.. any <2^60/R2-safe) inputs,
 * main-openmp.cc:424-440.  a, b are clobbered; result holds plain residues mod p. */
void orc_crt_residue_multiply(uint64_t *a, uint64_t *b, uint64_t *result, int len, uint64_t p,
                              int threads);
/* CRT_combine_parallel: main-openmp.cc:468-485. */
void orc_crt_combine(uint64_t *result, int n, const uint64_t *const *res_mods, const uint64_t *mods,
                     int k, const orc_crt_pre *pre, uint64_t target_mod, int threads);
/* Large-modulus branch of main(): main-openmp.cc:693-711.  Writes ab[0..2n-2] (< p64).
 * If residues != NULL it must hold 4*len u64 and receives the per-prime residue polynomials. */
void orc_poly_multiply_crt(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64,
                           int threads, uint64_t *residues, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
