"""GPU parity tests of the negacyclic product c = a * b mod (x^n + 1, p) (ntt_negacyclic_*) through the C ABI.

The reference has no negacyclic routine.  The definition is taken from the linear product its vectors pin:
c[i] = ab[i] - ab[i + n] (mod p) -- with ab the reference's own golden outputs (nttdata/<id>.out,
test_negacyclic_from_reference_golden) and, at other sizes and primes, ab = orc.poly_multiply
(oracle/ntt_oracle.c, main-montgomery.cc:317-351); plus a pure-Python schoolbook for tiny n and ring
identities (x^n = -1).
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

P = 998244353            # lazy kernels ([0, 2p) ranges), 2-adicity 23
P_STRICT = 1224736769    # > 2^30: strict kernels, 2-adicity 24
P_C = 469762049          # 2-adicity 26


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.fixture(scope="module")
def ctx(nt):
    c = nt.Context(0)
    yield c
    c.close()


def rand(rng, shape, p):
    return rng.integers(0, p, size=shape, dtype=np.int64).astype(np.uint32)


def negacyclic_from_linear(a, b, p):
    n = a.size
    ab = np.asarray(orc.poly_multiply(a, b, p)).astype(np.int64)
    lo = ab[:n].copy()
    hi = np.zeros(n, dtype=np.int64)
    hi[:n - 1] = ab[n:]
    return ((lo - hi) % p).astype(np.uint32)


def school_negacyclic(a, b, p):
    n = len(a)
    c = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n:
                c[k] += int(a[i]) * int(b[j])
            else:
                c[k - n] -= int(a[i]) * int(b[j])
    return np.array([v % p for v in c], dtype=np.uint32)


@pytest.mark.parametrize("p", [P, P_STRICT])
def test_negacyclic_tiny_schoolbook(nt, ctx, p):
    rng = np.random.default_rng(p % 97)
    for n in (2, 4, 8, 64):
        a, b = rand(rng, n, p), rand(rng, n, p)
        a[0] = p - 1
        b[n - 1] = p - 1
        for kind in (nt.MONTGOMERY, nt.BARRETT):
            plan = ctx.negacyclic_plan(p, n, 1, kind)
            assert plan.length == n
            got = plan.multiply(a, b)[0]
            plan.close()
            assert np.array_equal(got, school_negacyclic(a, b, p)), (p, n, kind)


# L = n: small path (<= 4096), one column level of 2 / 16 / 32 / 64 points, two levels (2^19)
@pytest.mark.parametrize("n", [256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288])
@pytest.mark.parametrize("p,kind", [(P, 0), (P_STRICT, 0), (P_C, 1)])
def test_negacyclic_vs_linear_oracle(nt, ctx, n, p, kind):
    rng = np.random.default_rng(n + p % 1000)
    a, b = rand(rng, n, p), rand(rng, n, p)
    a[-1] = p - 1
    b[-1] = p - 1
    plan = ctx.negacyclic_plan(p, n, 1, nt.MONTGOMERY if kind == 0 else nt.BARRETT)
    got = plan.multiply(a, b)[0]
    plan.close()
    assert np.array_equal(got, negacyclic_from_linear(a, b, p))


@pytest.mark.parametrize("seq_mode", ["1", "2", "0"])
def test_negacyclic_one_kernel_batched(nt, ctx, seq_mode, monkeypatch):
    """Full-length L = 8192 (default) and 16384 (NTT_AMD_FUSED=2 only: it ties with the three kernels) in
    seq_product_kernel (kernels_fused.hip): batches, a row stride beyond n, both reductions, lazy and strict primes;
    NTT_AMD_FUSED=0 is the three-kernel path on the same inputs."""
    monkeypatch.setenv("NTT_AMD_FUSED", seq_mode)
    rng = np.random.default_rng(31)
    for n in (8192, 16384):
        for p, kind in ((P, nt.MONTGOMERY), (P_STRICT, nt.MONTGOMERY), (P_C, nt.BARRETT), (P_STRICT, nt.BARRETT)):
            B = 5
            a, b = rand(rng, (B, n), p), rand(rng, (B, n), p)
            a[0, -1] = b[0, -1] = p - 1
            plan = ctx.negacyclic_plan(p, n, B, kind)
            got = plan.multiply(a, b, out=np.full((B, n + 3), 0xC3C3C3C3, dtype=np.uint32), out_stride=n + 3)
            plan.close()
            for r in range(B):
                assert np.array_equal(got[r, :n], negacyclic_from_linear(a[r], b[r], p)), (n, p, kind, r)
            assert np.all(got[:, n:] == 0xC3C3C3C3)


def test_negacyclic_ring_identities(ctx):
    """x^k * a rotates a by k with a sign flip on wrap (x^n = -1); 1 is the unit; (p-1) negates."""
    n, p = 65536, P
    rng = np.random.default_rng(2)
    a = rand(rng, n, p)
    plan = ctx.negacyclic_plan(p, n, 1)
    one = np.zeros(n, dtype=np.uint32)
    one[0] = 1
    assert np.array_equal(plan.multiply(a, one)[0], a)
    for k in (1, 5, n - 1):
        xk = np.zeros(n, dtype=np.uint32)
        xk[k] = 1
        want = np.concatenate([(p - a[n - k:].astype(np.int64)) % p, a[:n - k].astype(np.int64)]).astype(np.uint32)
        assert np.array_equal(plan.multiply(a, xk)[0], want), k
    neg = np.zeros(n, dtype=np.uint32)
    neg[0] = p - 1
    assert np.array_equal(plan.multiply(a, neg)[0], ((p - a.astype(np.int64)) % p).astype(np.uint32))
    plan.close()


def test_negacyclic_batched_device_resident(nt, ctx):
    """RNS shape (BASELINE.json configs[3]): n = 2^16, a batch of device-resident pairs, row stride n + 3
    (entries beyond n are not ours); several chunks and both lanes."""
    import torch
    n, B, p = 65536, 200, P_STRICT
    rng = np.random.default_rng(9)
    a, b = rand(rng, (B, n), p), rand(rng, (B, n), p)
    da = torch.from_numpy(a.view(np.int32)).cuda()
    db = torch.from_numpy(b.view(np.int32)).cuda()
    out = torch.full((B, n + 3), -3, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    plan = ctx.negacyclic_plan(p, n, B)
    plan.multiply(da, db, out=out, batch=B, out_stride=n + 3)
    ctx.synchronize()
    got = out.cpu().numpy()
    assert (got[:, n:] == -3).all()
    for i in (0, 1, 47, 48, 99, 199):
        assert np.array_equal(got[i, :n].view(np.uint32), negacyclic_from_linear(a[i], b[i], p)), i
    # linearity over the whole batch: (a + a') * b = a*b + a'*b in the ring
    s = ((a[:100].astype(np.int64) + a[100:].astype(np.int64)) % p).astype(np.uint32)
    ds = torch.from_numpy(s.view(np.int32)).cuda()
    out2 = torch.zeros((100, n), dtype=torch.int32, device="cuda")
    db2 = db[:100].contiguous()
    torch.cuda.synchronize()
    plan.multiply(ds, db2, out=out2, batch=100)
    plan.multiply(da[100:].contiguous(), db2, out=out[:100], batch=100, out_stride=n + 3)
    ctx.synchronize()
    lhs = out2.cpu().numpy().view(np.uint32).astype(np.int64)
    r1 = got[:100, :n].view(np.uint32).astype(np.int64)
    r2 = out.cpu().numpy()[:100, :n].view(np.uint32).astype(np.int64)
    assert np.array_equal(lhs, (r1 + r2) % p)
    plan.close()


def test_negacyclic_errors(nt, ctx):
    with pytest.raises(nt.NttError):
        ctx.negacyclic_plan(P, 1000)                      # not a power of two
    with pytest.raises(nt.NttError):
        ctx.negacyclic_plan(P, 1 << 23)                   # needs 2^24 | p - 1
    plan = ctx.negacyclic_plan(P, 64, 2)
    lin = ctx.plan(P, 64, 2)
    a = np.zeros((2, 64), dtype=np.uint32)
    with pytest.raises(nt.NttError):                      # entry points do not mix
        nt._check(nt._lib.ntt_poly_multiply_batched(plan._h, a.ctypes.data, a.ctypes.data,
                                                    np.zeros((2, 127), dtype=np.uint32).ctypes.data, 127, 2))
    with pytest.raises(nt.NttError):
        nt._check(nt._lib.ntt_negacyclic_multiply_batched(lin._h, a.ctypes.data, a.ctypes.data,
                                                          np.zeros((2, 64), dtype=np.uint32).ctypes.data, 64, 2))
    with pytest.raises(nt.NttError):
        plan.multiply(a, a, out=np.zeros((2, 64), dtype=np.uint32), out_stride=32)   # stride below the row length
    with pytest.raises(ValueError):
        plan.multiply(a, a, out=np.zeros((2, 32), dtype=np.uint32), out_stride=64)   # the binding's own size check
    plan.close()
    lin.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", [0, 1])
def test_negacyclic_from_reference_golden(nt, ctx, nttdata, cid, kind):
    """Anchored on the reference's own vectors: the golden linear product ab of nttdata/<id>.out reduced
    modulo x^n + 1 (c[i] = ab[i] - ab[i+n]) must equal the negacyclic product of the golden inputs."""
    n, p, a, b, ab = nttdata[cid]
    ab = ab.astype(np.int64)
    hi = np.zeros(n, dtype=np.int64)
    hi[:n - 1] = ab[n:]
    want = ((ab[:n] - hi) % p).astype(np.uint32)
    plan = ctx.negacyclic_plan(p, n, 1, nt.MONTGOMERY if kind == 0 else nt.BARRETT)
    got = plan.multiply(a.astype(np.uint32), b.astype(np.uint32))[0]
    plan.close()
    assert np.array_equal(got, want)


# ---- 64-bit lanes (ntt_negacyclic_plan64_create / ntt_negacyclic_multiply64_batched) ------------------------
P51 = 1337006139375617            # the reference's large modulus, 2-adicity 46
P62 = 4611685941117976577         # largest prime k * 2^32 + 1 below 2^62


def fold(ab, n, p):
    """c[i] = ab[i] - ab[i + n] (mod p) of a linear product of two length-n polynomials (Python integers)."""
    ab = [int(v) for v in ab]
    return np.array([(ab[i] - (ab[i + n] if i + n < len(ab) else 0)) % p for i in range(n)], dtype=np.uint64)


@pytest.mark.parametrize("n", [2, 8, 256, 1024, 2048, 4096, 8192, 65536, 131072])
@pytest.mark.parametrize("p", [P51, 1125899437080577])          # the reference's 51-bit prime, and one just below 2^50
def test_negacyclic64_vs_linear_crt_oracle(ctx, n, p, lanes64):
    """Small block (L <= 4096), one column level, two levels; against the 4-prime CRT oracle's linear product."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, p, size=n, dtype=np.uint64)
    b = rng.integers(0, p, size=n, dtype=np.uint64)
    a[-1] = p - 1
    b[-1] = p - 1
    lin, _ = orc.poly_multiply_crt(a, b, p, threads=8)
    plan = ctx.negacyclic_plan64(p, n)
    got = plan.multiply(a, b)[0]
    plan.close()
    assert np.array_equal(got, fold(lin, n, p))


def test_negacyclic64_golden_and_schoolbook(ctx, crt_vectors, lanes64):
    v = crt_vectors                       # big-integer golden linear product, n = 256
    plan = ctx.negacyclic_plan64(int(v["p64"]), 256)
    assert np.array_equal(plan.multiply(v["a_256"], v["b_256"])[0], fold(v["ab_256"], 256, int(v["p64"])))
    plan.close()
    rng = np.random.default_rng(3)
    for n in (2, 4, 32):                  # a 62-bit prime: Python schoolbook
        a = rng.integers(0, P62, size=n, dtype=np.uint64)
        b = rng.integers(0, P62, size=n, dtype=np.uint64)
        c = [0] * n
        for i in range(n):
            for j in range(n):
                t = int(a[i]) * int(b[j])
                if i + j < n:
                    c[i + j] += t
                else:
                    c[i + j - n] -= t
        plan = ctx.negacyclic_plan64(P62, n)
        got = plan.multiply(a, b)[0]
        plan.close()
        assert np.array_equal(got, np.array([x % P62 for x in c], dtype=np.uint64)), n


def test_negacyclic64_ring_identity_and_batch(nt, ctx, lanes64):
    """62-bit prime at n = 2^15, a batch on device arrays with a padded row stride: x^k * a rotates with a sign."""
    import torch
    n, B, p = 32768, 5, P62
    rng = np.random.default_rng(4)
    a = rng.integers(0, p, size=(B, n), dtype=np.uint64)
    xk = np.zeros((B, n), dtype=np.uint64)
    ks = [1, 2, 100, n - 1, 0]
    for i, k in enumerate(ks):
        xk[i, k] = 1
    da = torch.from_numpy(a.view(np.int64)).cuda()
    dx = torch.from_numpy(xk.view(np.int64)).cuda()
    out = torch.full((B, n + 2), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    plan = ctx.negacyclic_plan64(p, n, batch=8)
    plan.multiply(da, dx, out=out, batch=B, out_stride=n + 2)
    ctx.synchronize()
    got = out.cpu().numpy()
    assert (got[:, n:] == -1).all()
    for i, k in enumerate(ks):
        ai = [int(v) for v in a[i]]
        want = np.array([(p - v) % p for v in ai[n - k:]] + ai[:n - k], dtype=np.uint64) if k else a[i]
        assert np.array_equal(got[i, :n].view(np.uint64), want), k
    with pytest.raises(nt.NttError):                       # plan kinds do not mix
        nt._check(nt._lib.ntt_poly_multiply64_batched(plan._h, da.data_ptr(), dx.data_ptr(), out.data_ptr(), 2 * n, 1))
    plan.close()
    with pytest.raises(nt.NttError):
        ctx.negacyclic_plan64(p, 1000)


def test_randomised_shapes_all_rings(nt, ctx, lanes64):
    """Seeded sweep over ring, lane width, prime, size and batch: every combination against the linear
    product of the oracle (folded for x^n + 1)."""
    rng = np.random.default_rng(2025)
    primes32 = [998244353, 469762049, 1224736769, 1004535809, 104857601, 7340033]
    for trial in range(40):
        lg = int(rng.integers(1, 15))
        n = 1 << lg
        B = int(rng.integers(1, 4))
        p = primes32[int(rng.integers(0, len(primes32)))]
        if (p - 1) % (2 * n):
            continue
        a, b = rand(rng, (B, n), p), rand(rng, (B, n), p)
        kind = nt.MONTGOMERY if trial % 2 == 0 else nt.BARRETT
        plan = ctx.negacyclic_plan(p, n, B, kind)
        got = plan.multiply(a, b)
        plan.close()
        for i in range(B):
            assert np.array_equal(got[i], negacyclic_from_linear(a[i], b[i], p)), (trial, n, p, kind, i)
    for trial in range(16):
        lg = int(rng.integers(1, 16))
        n = 1 << lg
        B = int(rng.integers(1, 3))
        a = rng.integers(0, P51, size=(B, n), dtype=np.uint64)
        b = rng.integers(0, P51, size=(B, n), dtype=np.uint64)
        neg = trial % 2 == 0
        plan = ctx.negacyclic_plan64(P51, n, B) if neg else ctx.plan64(P51, n, B)
        got = plan.multiply(a, b)
        plan.close()
        for i in range(B):
            lin, _ = orc.poly_multiply_crt(a[i], b[i], P51, threads=4)
            want = fold(lin, n, P51) if neg else lin
            assert np.array_equal(got[i], want), (trial, n, neg, i)
