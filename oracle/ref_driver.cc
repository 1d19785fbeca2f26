// ref_driver.cc -- C-ABI shim over the REAL reference translation unit main-baseline.cc.
//
// TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists (this container); the
// output goes to oracle/_ref/ (git-ignored, travels to the GPU box as a prebuilt .so).
// The reference source is compiled where it lies (REF_TU is passed by the Makefile as an
// absolute path under /root/reference); nothing from it is copied into this repository.
//
// main-baseline.cc is the one hot-path TU of the reference that builds on this image with no
// stand-ins: main-montgomery.cc, main-barrett.cc and main-openmp.cc include <arm_neon.h>
// (absent on x86-64), main-mpi.cc needs MPI and the .cu files need CUDA, so those are
// "unbuildable here" and are pinned through nttdata/*.out and the restatement instead.
#define main ref_baseline_main
#include REF_TU
#undef main

extern "C" {

// NTT_iterative(std::vector<int>&, n, p, inv): main-baseline.cc:148-182
void ref_ntt_iterative(int *data, int n, int p, int inv)
{
    std::vector<int> v(data, data + n);
    NTT_iterative(v, n, p, inv);
    std::copy(v.begin(), v.end(), data);
}

// poly_multiply (schoolbook, accumulating into ab): main-baseline.cc:70-76
void ref_poly_multiply_naive(int *pa, int *pb, int *pab, int n, int p)
{
    poly_multiply(pa, pb, pab, n, p);
}

// The multiply body of main(): main-baseline.cc:201-232 (len = next power of two >= 2n).
void ref_poly_multiply_ntt(const int *pa, const int *pb, int *pab, int n, int p)
{
    int len = 1;
    while (len < 2 * n) len <<= 1;
    std::vector<int> a_1(len, 0), b_1(len, 0), c(len);
    for (int i = 0; i < n; ++i) { a_1[i] = pa[i]; b_1[i] = pb[i]; }
    NTT_iterative(a_1, len, p, 1);
    NTT_iterative(b_1, len, p, 1);
    for (int i = 0; i < len; ++i) c[i] = 1LL * a_1[i] * b_1[i] % p;
    NTT_iterative(c, len, p, -1);
    for (int i = 0; i < 2 * n - 1; ++i) pab[i] = c[i];
}

// power(int,int,int): main-baseline.cc:81-94
unsigned long long ref_power(int base, int exp, int mod) { return power(base, exp, mod); }

}  // extern "C"
