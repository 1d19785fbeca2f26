/*
 * ntt_oracle.c -- CPU restatement of the reference NTT polynomial-multiply path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see ntt_oracle.h).  The product (ntt_amd/) never links this.
 * Citations are file:line relative to /root/reference.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC ntt_oracle.c -o libntt_oracle.so   (reference flags:
 * test.sh unpacked = "g++ main.cc -o main -O2 -fopenmp -lpthread -std=c++11").
 */
#include "ntt_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef __uint128_t u128;

/* ---- power: main-openmp.cc:163-176 --------------------------------------------------- */
uint64_t orc_power(uint64_t base, uint64_t exp, uint64_t mod)
{
    /* The reference multiplies in 64 bits (1LL * res * base), so it is only valid for
     * mod < 2^32; the restatement keeps that arithmetic width on purpose. */
    uint64_t res = 1;
    for (; exp > 0; exp >>= 1) {
        if (exp & 1) res = res * base % mod;
        base = base * base % mod;
    }
    return res;
}

uint64_t orc_power128(uint64_t base, uint64_t exp, uint64_t mod)
{
    uint64_t res = 1 % mod;
    base %= mod;
    for (; exp > 0; exp >>= 1) {
        if (exp & 1) res = (uint64_t)((u128)res * base % mod);
        base = (uint64_t)((u128)base * base % mod);
    }
    return res;
}

/* ---- montgomery: main-montgomery.cc:133-234 ------------------------------------------- */
/* gcd_extended :144-157, recursive in the reference; same recurrence here. */
static uint64_t ext_gcd(uint64_t a, uint64_t b, int64_t *x, int64_t *y)
{
    if (a == 0) { *x = 0; *y = 1; return b; }
    int64_t x1, y1;
    uint64_t g = ext_gcd(b % a, a, &x1, &y1);
    *x = y1 - (int64_t)(b / a) * x1;
    *y = x1;
    return g;
}

int orc_mont_init(orc_mont *m, uint64_t R, uint64_t N)
{
    m->N = N;
    m->R = R;
    int lg = 0;
    while ((1ULL << lg) < R && lg < 63) lg++;       /* std::log2(R) for a power of two, :138 */
    m->logR = lg;
    int64_t x, y;
    if (ext_gcd(N, R, &x, &y) != 1) return -1;      /* modinv throws, :163-166 */
    uint64_t Ninv = (uint64_t)((x % (int64_t)R + (int64_t)R) % (int64_t)R);   /* :167 */
    m->N_inv_neg = R - Ninv;                        /* :140 */
    m->R2 = (uint64_t)(((u128)R * R) % N);          /* :141 */
    return 0;
}

uint64_t orc_mont_redc(const orc_mont *m, u128 T)
{
    uint64_t mask = (1ULL << m->logR) - 1;                          /* :172 */
    uint64_t q = (((uint64_t)T & mask) * m->N_inv_neg) & mask;      /* :173 */
    u128 t = (T + (u128)q * m->N) >> m->logR;                       /* :174-175 */
    return (uint64_t)(t >= m->N ? t - m->N : t);                    /* :176-180 */
}
uint64_t orc_mont_to(const orc_mont *m, uint64_t x)   { return orc_mont_redc(m, (u128)x * m->R2); }
uint64_t orc_mont_from(const orc_mont *m, uint64_t x) { return orc_mont_redc(m, (u128)x); }
uint64_t orc_mont_mul(const orc_mont *m, uint64_t a, uint64_t b) { return orc_mont_redc(m, (u128)a * b); }

/* ---- Barrett: main-barrett.cc:130-172 -------------------------------------------------- */
void orc_barrett_init(orc_barrett *b, uint64_t mod)
{
    b->mod = mod;
    b->factor = (uint64_t)((((u128)1) << 64) / mod);     /* :137 */
}
uint64_t orc_barrett_reduce(const orc_barrett *b, u128 x)
{
    /* :147 multiplies the 128-bit x by factor in 128-bit arithmetic (upper bits wrap);
     * valid while x < 2^64, which holds for products of residues < 2^32. */
    uint64_t q = (uint64_t)((x * b->factor) >> 64);
    uint64_t r = (uint64_t)x - q * b->mod;
    return r >= b->mod ? r - b->mod : r;
}
uint64_t orc_barrett_mul(const orc_barrett *b, uint64_t x, uint64_t y)
{
    return orc_barrett_reduce(b, (u128)x * y);
}

/* ---- bit_reverse: main-montgomery.cc:238-253 ------------------------------------------- */
void orc_bit_reverse(uint64_t *a, int n)
{
    int j = 0;
    for (int i = 1; i < n; ++i) {
        int bit = n >> 1;
        while (j >= bit) { j -= bit; bit >>= 1; }
        j += bit;
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* ---- NTT_iterative (Montgomery): main-montgomery.cc:255-295 ----------------------------- */
static void ntt_mont_impl(uint64_t *a, int n, uint64_t p, int inv, const orc_mont *m, int threads)
{
    orc_bit_reverse(a, n);                                           /* :258 */
    for (int len = 2; len <= n && len > 0; len <<= 1) {              /* :260 */
        uint64_t wn = orc_power(3, (p - 1) / (uint64_t)len, p);      /* :262 */
        if (inv == -1) wn = orc_power(wn, p - 2, p);                 /* :263-266 */
        uint64_t wnR = orc_mont_to(m, wn);                           /* :268 */
        int half = len / 2, blocks = n / len;
        /* main-openmp.cc:397: parallel over the n/len blocks of this stage, static schedule. */
        #pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
        for (int blk = 0; blk < blocks; ++blk) {
            uint64_t *lo = a + (size_t)blk * len, *hi = lo + half;
            uint64_t w = orc_mont_to(m, 1);                          /* :272 */
            for (int j = 0; j < half; ++j) {                         /* :273-281 */
                uint64_t u = lo[j];
                uint64_t v = orc_mont_mul(m, w, hi[j]);
                lo[j] = (u + v) % p;
                hi[j] = (u + p - v) % p;
                w = orc_mont_mul(m, w, wnR);
            }
        }
    }
    if (inv == -1) {                                                 /* :285-294, serial */
        uint64_t ninv = orc_power((uint64_t)n, p - 2, p);
        uint64_t ninvR = orc_mont_to(m, ninv);
        for (int i = 0; i < n; ++i) a[i] = orc_mont_mul(m, a[i], ninvR);
    }
}
void orc_ntt_mont(uint64_t *a, int n, uint64_t p, int inv, const orc_mont *m)
{
    ntt_mont_impl(a, n, p, inv, m, 1);
}
void orc_ntt_mont_omp(uint64_t *a, int n, uint64_t p, int inv, const orc_mont *m, int threads)
{
    ntt_mont_impl(a, n, p, inv, m, threads);
}

/* ---- NTT_iterative (Barrett): main-barrett.cc:193-228 ----------------------------------- */
void orc_ntt_barrett(uint64_t *a, int n, uint64_t p, int inv, const orc_barrett *bt)
{
    orc_bit_reverse(a, n);
    for (int len = 2; len <= n && len > 0; len <<= 1) {
        uint64_t wn = orc_power(3, (p - 1) / (uint64_t)len, p);
        if (inv == -1) wn = orc_power(wn, p - 2, p);
        int half = len / 2;
        for (int i = 0; i < n; i += len) {
            uint64_t w = 1;
            for (int j = 0; j < half; ++j) {
                uint64_t u = a[i + j];
                uint64_t v = orc_barrett_mul(bt, w, a[i + j + half]);
                uint64_t s = u + v;                                   /* add :157-161 */
                a[i + j] = s >= p ? s - p : s;
                a[i + j + half] = u >= v ? u - v : p - (v - u);       /* sub :163-166 */
                w = orc_barrett_mul(bt, w, wn);
            }
        }
    }
    if (inv == -1) {
        uint64_t ninv = orc_power((uint64_t)n, p - 2, p);
        for (int i = 0; i < n; ++i) a[i] = orc_barrett_mul(bt, a[i], ninv);
    }
}

void orc_ntt_plain(uint32_t *a, int n, uint32_t p, int inv)
{
    orc_mont m;
    orc_mont_init(&m, 1ULL << 32, p);
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    for (int i = 0; i < n; ++i) t[i] = orc_mont_to(&m, a[i]);
    orc_ntt_mont(t, n, p, inv, &m);
    for (int i = 0; i < n; ++i) a[i] = (uint32_t)orc_mont_from(&m, t[i]);
    free(t);
}

/* ---- poly_multiply (schoolbook): main-montgomery.cc:71-77 ------------------------------- */
void orc_poly_multiply_naive(const int *a, const int *b, int *ab, int n, int p)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            ab[i + j] = (int)(((long long)a[i] * b[j] % p + ab[i + j]) % p);
}

/* ---- harness multiply body, Montgomery: main-montgomery.cc:315-351 ---------------------- */
void orc_poly_multiply_mont(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p)
{
    orc_mont m;
    orc_mont_init(&m, 1ULL << 32, p);                                /* :315-316 */
    int len = 1;
    while (len < 2 * n) len <<= 2;                                   /* :317-321: power of FOUR */
    uint64_t *A = (uint64_t *)calloc((size_t)len, 8), *B = (uint64_t *)calloc((size_t)len, 8);
    uint64_t *C = (uint64_t *)calloc((size_t)len, 8);
    for (int i = 0; i < n; ++i) { A[i] = orc_mont_to(&m, a[i]); B[i] = orc_mont_to(&m, b[i]); }
    orc_ntt_mont(A, len, p, 1, &m);                                  /* :339 */
    orc_ntt_mont(B, len, p, 1, &m);                                  /* :340 */
    for (int i = 0; i < len; ++i) C[i] = orc_mont_mul(&m, A[i], B[i]);   /* :341-344 */
    orc_ntt_mont(C, len, p, -1, &m);                                 /* :345 */
    for (int i = 0; i < 2 * n - 1; ++i) ab[i] = (uint32_t)orc_mont_from(&m, C[i]);   /* :347-351 */
    free(A); free(B); free(C);
}

/* ---- harness multiply body, Barrett: main-barrett.cc:248-282 ---------------------------- */
void orc_poly_multiply_barrett(const uint32_t *a, const uint32_t *b, uint32_t *ab, int n, uint32_t p)
{
    orc_barrett bt;
    orc_barrett_init(&bt, p);
    int len = 1;
    while (len < 2 * n) len <<= 1;                                   /* :249-253: power of TWO */
    uint64_t *A = (uint64_t *)calloc((size_t)len, 8), *B = (uint64_t *)calloc((size_t)len, 8);
    uint64_t *C = (uint64_t *)calloc((size_t)len, 8);
    for (int i = 0; i < n; ++i) { A[i] = a[i]; B[i] = b[i]; }
    orc_ntt_barrett(A, len, p, 1, &bt);
    orc_ntt_barrett(B, len, p, 1, &bt);
    for (int i = 0; i < len; ++i) C[i] = orc_barrett_mul(&bt, A[i], B[i]);
    orc_ntt_barrett(C, len, p, -1, &bt);
    for (int i = 0; i < 2 * n - 1; ++i) ab[i] = (uint32_t)C[i];
    free(A); free(B); free(C);
}

/* ---- OpenMP single-prime branch: main-openmp.cc:650-692 --------------------------------- */
static void vec_to_mont(const orc_mont *m, uint64_t *v, int len, int threads)   /* :277-284 */
{
    #pragma omp parallel for num_threads(threads) if (threads > 1)
    for (int i = 0; i < len; ++i) v[i] = orc_mont_to(m, v[i]);
}
static void vec_from_mont(const orc_mont *m, uint64_t *v, int len, int threads) /* :286-293 */
{
    #pragma omp parallel for num_threads(threads) if (threads > 1)
    for (int i = 0; i < len; ++i) v[i] = orc_mont_from(m, v[i]);
}

void orc_poly_multiply_omp(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p,
                           int threads, double *seconds)
{
    orc_mont m;
    orc_mont_init(&m, 1ULL << 60, p);                                /* :650-651 */
    int len = 1;
    while (len < 2 * n) len <<= 1;                                   /* :662-666 */
    uint64_t *A = (uint64_t *)calloc((size_t)len, 8), *B = (uint64_t *)calloc((size_t)len, 8);
    uint64_t *C = (uint64_t *)calloc((size_t)len, 8);
    for (int i = 0; i < n; ++i) { A[i] = a[i]; B[i] = b[i]; }
    double t0 = omp_get_wtime();                                     /* Start :675 */
    vec_to_mont(&m, A, len, threads);                                /* :680-681 */
    vec_to_mont(&m, B, len, threads);
    ntt_mont_impl(A, len, p, 1, &m, threads);                        /* :682-683 */
    ntt_mont_impl(B, len, p, 1, &m, threads);
    #pragma omp parallel for num_threads(threads) if (threads > 1)   /* :685-689 */
    for (int i = 0; i < len; ++i) C[i] = orc_mont_mul(&m, A[i], B[i]);
    ntt_mont_impl(C, len, p, -1, &m, threads);                       /* :690 */
    vec_from_mont(&m, C, len, threads);                              /* :691 */
    double t1 = omp_get_wtime();                                     /* End :712 */
    if (seconds) *seconds = t1 - t0;
    for (int i = 0; i < 2 * n - 1; ++i) ab[i] = C[i];                /* :713-716 */
    free(A); free(B); free(C);
}

/* ---- CRT path: main-openmp.cc:424-485, 693-711 ------------------------------------------ */
const uint64_t orc_crt_mods[4] = {1004535809ULL, 1224736769ULL, 469762049ULL, 998244353ULL};

void orc_crt_precompute(orc_crt_pre *pre, const uint64_t *mods, int k)
{
    pre->M = 1;
    for (int i = 0; i < k; ++i) pre->M *= mods[i];                   /* :452-456 */
    for (int i = 0; i < k; ++i) {
        pre->Mi[i] = pre->M / mods[i];                               /* :462 */
        pre->inv[i] = orc_power((uint64_t)(pre->Mi[i] % mods[i]), mods[i] - 2, mods[i]);   /* :463 */
    }
}

void orc_crt_residue_multiply(uint64_t *a, uint64_t *b, uint64_t *result, int len, uint64_t p,
                              int threads)
{
    orc_mont m;
    orc_mont_init(&m, 1ULL << 60, p);                                /* :656-661 */
    /* toMontgomery also reduces the unreduced inputs mod p (x*R2 < R*N for x < 2^60). :426-427 */
    vec_to_mont(&m, a, len, threads);
    vec_to_mont(&m, b, len, threads);
    ntt_mont_impl(a, len, p, 1, &m, threads);                        /* :429-430 */
    ntt_mont_impl(b, len, p, 1, &m, threads);
    #pragma omp parallel for num_threads(threads) if (threads > 1)   /* :432-436 */
    for (int i = 0; i < len; ++i) result[i] = orc_mont_mul(&m, a[i], b[i]);
    ntt_mont_impl(result, len, p, -1, &m, threads);                  /* :438 */
    vec_from_mont(&m, result, len, threads);                         /* :439 */
}

void orc_crt_combine(uint64_t *result, int n, const uint64_t *const *res_mods, const uint64_t *mods,
                     int k, const orc_crt_pre *pre, uint64_t target_mod, int threads)
{
    #pragma omp parallel for num_threads(threads) if (threads > 1)   /* :473 */
    for (int i = 0; i < n; ++i) {
        u128 res = 0;
        for (int j = 0; j < k; ++j) {
            uint64_t t = res_mods[j][i] * pre->inv[j] % mods[j];     /* :479 (u64 product, < 2^62) */
            u128 term = (u128)t * pre->Mi[j] % pre->M;               /* :480 */
            res = (res + term) % pre->M;                             /* :481 */
        }
        result[i] = (uint64_t)(res % target_mod);                    /* :483 */
    }
}

void orc_poly_multiply_crt(const uint64_t *a, const uint64_t *b, uint64_t *ab, int n, uint64_t p64,
                           int threads, uint64_t *residues, double *seconds)
{
    int len = 1;
    while (len < 2 * n) len <<= 1;
    orc_crt_pre pre;
    orc_crt_precompute(&pre, orc_crt_mods, 4);                       /* :655 */
    uint64_t *A = (uint64_t *)malloc((size_t)len * 8), *B = (uint64_t *)malloc((size_t)len * 8);
    uint64_t *own = residues ? NULL : (uint64_t *)malloc((size_t)len * 8 * 4);
    uint64_t *R = residues ? residues : own;
    uint64_t *C = (uint64_t *)malloc((size_t)len * 8);
    double t0 = omp_get_wtime();
    /* The reference runs the four primes in a 4-thread region with nested (inactive) inner
     * regions (:699); the arithmetic is identical run one prime after another. */
    for (int k = 0; k < 4; ++k) {
        memset(A, 0, (size_t)len * 8);
        memset(B, 0, (size_t)len * 8);
        for (int j = 0; j < n; ++j) { A[j] = a[j]; B[j] = b[j]; }    /* :702-706 */
        orc_crt_residue_multiply(A, B, R + (size_t)k * len, len, orc_crt_mods[k], threads);  /* :707 */
    }
    const uint64_t *rows[4] = {R, R + len, R + 2 * (size_t)len, R + 3 * (size_t)len};
    orc_crt_combine(C, len, rows, orc_crt_mods, 4, &pre, p64, threads);     /* :710 */
    double t1 = omp_get_wtime();
    if (seconds) *seconds = t1 - t0;
    for (int i = 0; i < 2 * n - 1; ++i) ab[i] = C[i];
    free(A); free(B); free(C); free(own);
}
