"""ctypes bindings for the CPU oracle (oracle/libntt_oracle.so) and, when present, the compiled
reference TU (oracle/_ref/libref_baseline.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package ntt_amd never imports this module.
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORC_PATH = os.path.join(_ROOT, "oracle", "libntt_oracle.so")
_REF_PATH = os.path.join(_ROOT, "oracle", "_ref", "libref_baseline.so")

CRT_MODS = (1004535809, 1224736769, 469762049, 998244353)   # main-openmp.cc:654

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def _load_oracle():
    if not os.path.exists(_ORC_PATH):
        raise RuntimeError(f"{_ORC_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(_ORC_PATH)
    lib.orc_power.restype = C.c_uint64
    lib.orc_power.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    lib.orc_power128.restype = C.c_uint64
    lib.orc_power128.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    lib.orc_ntt_plain.argtypes = [_u32p, C.c_int, C.c_uint32, C.c_int]
    lib.orc_ntt_plain.restype = None
    lib.orc_poly_multiply_naive.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
    lib.orc_poly_multiply_naive.restype = None
    for name in ("orc_poly_multiply_mont", "orc_poly_multiply_barrett"):
        f = getattr(lib, name)
        f.argtypes = [_u32p, _u32p, _u32p, C.c_int, C.c_uint32]
        f.restype = None
    lib.orc_poly_multiply_omp.argtypes = [_u64p, _u64p, _u64p, C.c_int, C.c_uint64, C.c_int,
                                          C.POINTER(C.c_double)]
    lib.orc_poly_multiply_omp.restype = None
    lib.orc_poly_multiply_crt.argtypes = [_u64p, _u64p, _u64p, C.c_int, C.c_uint64, C.c_int,
                                          C.c_void_p, C.POINTER(C.c_double)]
    lib.orc_poly_multiply_crt.restype = None
    return lib


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load_oracle()
    return _lib


def have_ref():
    return os.path.exists(_REF_PATH)


def ref():
    """The reference's main-baseline.cc compiled in place (oracle/ref_driver.cc)."""
    global _ref
    if _ref is None:
        r = C.CDLL(_REF_PATH)
        r.ref_ntt_iterative.argtypes = [_i32p, C.c_int, C.c_int, C.c_int]
        r.ref_ntt_iterative.restype = None
        r.ref_poly_multiply_naive.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        r.ref_poly_multiply_naive.restype = None
        r.ref_poly_multiply_ntt.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        r.ref_poly_multiply_ntt.restype = None
        r.ref_power.argtypes = [C.c_int, C.c_int, C.c_int]
        r.ref_power.restype = C.c_uint64
        _ref = r
    return _ref


# ---- convenience wrappers (numpy in, numpy out) -------------------------------------------
def ntt(a, p, inverse=False):
    """Forward / inverse NTT, natural order in and out, root 3^((p-1)/n) (main-montgomery.cc:255)."""
    a = np.ascontiguousarray(a, dtype=np.uint32).copy()
    lib().orc_ntt_plain(a, a.size, int(p), -1 if inverse else 1)
    return a


def poly_multiply(a, b, p, kind="mont"):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    n = a.size
    ab = np.zeros(max(2 * n - 1, 1), dtype=np.uint32)
    fn = lib().orc_poly_multiply_mont if kind == "mont" else lib().orc_poly_multiply_barrett
    fn(a, b, ab, n, int(p))
    return ab


def poly_multiply_naive(a, b, p):
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    n = a.size
    ab = np.zeros(max(2 * n - 1, 1), dtype=np.int32)
    lib().orc_poly_multiply_naive(a, b, ab, n, int(p))
    return ab.astype(np.uint32)


def poly_multiply_omp(a, b, p, threads=8):
    """main-openmp.cc single-prime branch; returns (ab, seconds between Start/End)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    n = a.size
    ab = np.zeros(max(2 * n - 1, 1), dtype=np.uint64)
    sec = C.c_double(0.0)
    lib().orc_poly_multiply_omp(a, b, ab, n, int(p), int(threads), C.byref(sec))
    return ab, sec.value


def poly_multiply_crt(a, b, p64, threads=8, want_residues=False):
    """main-openmp.cc large-modulus branch (4-prime CRT)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    n = a.size
    length = 1
    while length < 2 * n:
        length <<= 1
    ab = np.zeros(max(2 * n - 1, 1), dtype=np.uint64)
    res = np.zeros((4, length), dtype=np.uint64) if want_residues else None
    sec = C.c_double(0.0)
    lib().orc_poly_multiply_crt(a, b, ab, n, int(p64), int(threads),
                                res.ctypes.data if res is not None else None, C.byref(sec))
    return (ab, res, sec.value) if want_residues else (ab, sec.value)


def splitmix64_stream(seed, count):
    """coefficients from splitmix64(seed), `next() % p` applied by the caller."""
    out = np.empty(count, dtype=np.uint64)
    mask = (1 << 64) - 1
    x = seed & mask
    for i in range(count):
        x = (x + 0x9E3779B97F4A7C15) & mask
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        out[i] = z ^ (z >> 31)
    return out


def splitmix64_np(seed, count):
    """Vectorised splitmix64: element i is the (i+1)-th output of the generator seeded `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        x = np.uint64(seed & ((1 << 64) - 1)) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
