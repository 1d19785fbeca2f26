"""Pin the CPU oracle (oracle/ntt_oracle.c) against the reference's own fixtures:
  * nttdata/{0,1,2,3}.in/.out (shipped golden vectors),
  * vectors produced by the reference's main-baseline.cc compiled in place (tests/golden/
    ref_*_vectors.npz, made by tests/golden/make_golden.py),
  * live cross-check against oracle/_ref when that .so is present,
  * Python big-int schoolbook for the CRT path (unpinned by the reference's own data).
CPU only.
"""
import numpy as np
import pytest

import orc

PRIMES = (7340033, 104857601, 469762049, 998244353)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_oracle_matches_nttdata(nttdata, cid, kind):
    n, p, a, b, ab = nttdata[cid]
    got = orc.poly_multiply(a, b, p, kind=kind)
    assert got.shape == ab.shape
    assert np.array_equal(got, ab)


@pytest.mark.parametrize("cid", [0, 1])
def test_oracle_omp_matches_nttdata(nttdata, cid):
    n, p, a, b, ab = nttdata[cid]
    for threads in (1, 4):
        got, sec = orc.poly_multiply_omp(a, b, p, threads=threads)
        assert np.array_equal(got.astype(np.uint32), ab)
        assert sec >= 0.0


def test_naive_matches_nttdata0(nttdata):
    n, p, a, b, ab = nttdata[0]
    assert np.array_equal(orc.poly_multiply_naive(a, b, p), ab)


def test_oracle_ntt_matches_reference_vectors(ref_ntt_vectors):
    v = ref_ntt_vectors
    keys = [k for k in v.files if k.startswith("in_")]
    assert len(keys) == 28
    for k in keys:
        _, p, L = k.split("_")
        p, L = int(p), int(L)
        x = v[k]
        assert np.array_equal(orc.ntt(x, p), v[f"fwd_{p}_{L}"]), k
        assert np.array_equal(orc.ntt(x, p, inverse=True), v[f"inv_{p}_{L}"]), k
        assert np.array_equal(orc.ntt(orc.ntt(x, p), p, inverse=True), x), k


def test_oracle_polymul_matches_reference_vectors(ref_polymul_vectors):
    v = ref_polymul_vectors
    keys = [k for k in v.files if k.startswith("ab_")]
    assert len(keys) == 36
    for k in keys:
        _, p, n = k.split("_")
        p, n = int(p), int(n)
        a, b = v[f"a_{p}_{n}"], v[f"b_{p}_{n}"]
        for kind in ("mont", "barrett"):
            assert np.array_equal(orc.poly_multiply(a, b, p, kind=kind), v[k]), (k, kind)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_vs_live_reference():
    rng = np.random.default_rng(7)
    r = orc.ref()
    for p in PRIMES:
        for n in (7, 64, 300, 2048):
            a = rng.integers(0, p, size=n).astype(np.int32)
            b = rng.integers(0, p, size=n).astype(np.int32)
            ab = np.zeros(2 * n - 1, dtype=np.int32)
            r.ref_poly_multiply_ntt(a, b, ab, n, p)
            assert np.array_equal(orc.poly_multiply(a.astype(np.uint32), b.astype(np.uint32), p),
                                  ab.astype(np.uint32))
        for L in (32, 512, 8192):
            x = rng.integers(0, p, size=L).astype(np.int32)
            f = x.copy()
            r.ref_ntt_iterative(f, L, p, 1)
            assert np.array_equal(orc.ntt(x.astype(np.uint32), p), f.astype(np.uint32))
        assert r.ref_power(3, p - 2, p) == orc.lib().orc_power(3, p - 2, p)


def test_oracle_crt_matches_bigint(crt_vectors):
    v = crt_vectors
    p64 = int(v["p64"])
    for n in (256, 1000):
        ab, res, _ = orc.poly_multiply_crt(v[f"a_{n}"], v[f"b_{n}"], p64, threads=2, want_residues=True)
        assert np.array_equal(ab, v[f"ab_{n}"])
        for k in range(4):
            assert np.array_equal(res[k, :2 * n - 1].astype(np.uint32), v[f"res{k}_{n}"])


def test_splitmix_vectorised_matches_scalar():
    assert np.array_equal(orc.splitmix64_np(0x20250711, 100), orc.splitmix64_stream(0x20250711, 100))
    # reference value of splitmix64(seed=0) first output
    assert int(orc.splitmix64_stream(0, 1)[0]) == 0xE220A8397B1DCDAF
