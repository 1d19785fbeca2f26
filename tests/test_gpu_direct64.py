"""GPU parity tests of the direct 64-bit path (ntt_plan64_*, kernels_u64.hip) through the C ABI.

What pins it: the big-integer golden vectors of tests/golden/crt_vectors.npz (p = 1337006139375617,
the modulus of the reference's fifth case), the 4-prime CRT oracle (oracle/ntt_oracle.c,
orc_poly_multiply_crt, restating main-openmp.cc:693-711) wherever n * p^2 stays below the product of
its primes, Python big-integer schoolbook products for small n, and -- for primes near 2^62, where
no oracle applies -- the evaluation identity A(x) * B(x) = AB(x) (mod p).  The reference's own
cases nttdata/0-3 (32-bit primes) run through the 64-bit lanes too and must reproduce its golden
outputs; for a modulus above 32 bits the reference holds no data (its `power` is 32-bit),
so that range is pinned by the big-integer vectors and the CRT oracle only.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

P51 = 1337006139375617            # reference case 5 (main-openmp.cc:636), 2-adicity 46
P62 = 4611685941117976577         # largest prime k * 2^32 + 1 below 2^62
P50 = 1125899437080577            # largest prime k * 2^24 + 1 below 2^50
P33 = 8588886017                  # largest prime k * 2^20 + 1 below 2^33
P51B = 2251799696244737           # largest prime k * 2^24 + 1 below 2^51: the widest the FP64 lanes take


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.fixture(scope="module")
def ctx(nt):
    c = nt.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _both_policies(lanes64):
    return lanes64


def school(a, b, p):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        x = int(x)
        for j, y in enumerate(b):
            out[i + j] += x * int(y)
    return np.array([v % p for v in out], dtype=np.uint64)


def horner(c, x, p):
    acc = 0
    for v in c[::-1]:
        acc = (acc * x + int(v)) % p
    return acc


def test_direct64_golden(ctx, crt_vectors):
    v = crt_vectors
    p64 = int(v["p64"])
    assert p64 == P51
    for n in (256, 1000):
        plan = ctx.plan64(p64, n)
        got = plan.multiply(v[f"a_{n}"], v[f"b_{n}"])[0]
        assert np.array_equal(got, v[f"ab_{n}"]), n
        plan.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 2048, 2049, 4096, 10000, 65536, 131072])
@pytest.mark.parametrize("p", [P51, P50])
def test_direct64_vs_crt_oracle(ctx, n, p):
    """Every plan shape: one LDS block (L <= 4096), one column level (L = 8192, 32768), two levels
    (L = 2^17: 16 x 2, L = 2^18: 16 x 4); the reference's 51-bit prime (general FP64 policy) and one just
    below 2^50 (the FP64 policy without the closing fold)."""
    rng = np.random.default_rng(1000 + n)
    a = rng.integers(0, p, size=n, dtype=np.uint64)
    b = rng.integers(0, p, size=n, dtype=np.uint64)
    a[0] = p - 1
    b[-1] = p - 1
    want, _ = orc.poly_multiply_crt(a, b, p, threads=8)
    plan = ctx.plan64(p, n)
    got = plan.multiply(a, b)[0]
    plan.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [7, 4096, 50000, 65536])
@pytest.mark.parametrize("p", [P51B, P50])
def test_direct64_top_of_fp64_range(ctx, n, p):
    """p just below 2^51 and just below 2^50, where the rounding bounds of the two FP64 policies are tightest
    (|x*w/p| up to 2^52): random and all-(p-1) operands against the CRT oracle (n * p^2 < product of its
    primes up to 2^17)."""
    rng = np.random.default_rng(51 + n)
    plan = ctx.plan64(p, n)
    for a, b in ((rng.integers(0, p, size=n, dtype=np.uint64), rng.integers(0, p, size=n, dtype=np.uint64)),
                 (np.full(n, p - 1, dtype=np.uint64), np.full(n, p - 1, dtype=np.uint64)),
                 (np.full(n, p - 1, dtype=np.uint64), np.ones(n, dtype=np.uint64))):
        want, _ = orc.poly_multiply_crt(a, b, p, threads=8)
        assert np.array_equal(plan.multiply(a, b)[0], want)
    plan.close()


def test_direct64_matches_crt_path_on_gpu(ctx):
    """The same caller served two ways: ntt_crt_multiply and the direct path."""
    rng = np.random.default_rng(5)
    n = 50000
    a = rng.integers(0, P51, size=n, dtype=np.uint64)
    b = rng.integers(0, P51, size=n, dtype=np.uint64)
    plan = ctx.plan64(P51, n)
    assert np.array_equal(plan.multiply(a, b)[0], ctx.crt_multiply(a, b, P51))
    plan.close()


@pytest.mark.parametrize("p", [P33, P50, P62])
def test_direct64_small_schoolbook(ctx, p):
    rng = np.random.default_rng(p % 1000)
    for n in (1, 2, 7, 33, 100):
        a = rng.integers(0, p, size=n, dtype=np.uint64)
        b = rng.integers(0, p, size=n, dtype=np.uint64)
        a[0] = p - 1
        b[0] = p - 1
        plan = ctx.plan64(p, n)
        got = plan.multiply(a, b)[0]
        plan.close()
        assert np.array_equal(got, school(a, b, p)), (p, n)


@pytest.mark.parametrize("p,n", [(P62, 3000), (P62, 131072), (P50, 20000), (P33, 100000)])
def test_direct64_evaluation_identity(ctx, p, n):
    """No oracle reaches a 62-bit prime (the CRT oracle needs n * p^2 < 2^119.6): check
    A(x) * B(x) = AB(x) (mod p) at three points, plus canonical range."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, p, size=n, dtype=np.uint64)
    b = rng.integers(0, p, size=n, dtype=np.uint64)
    plan = ctx.plan64(p, n)
    ab = plan.multiply(a, b)[0]
    plan.close()
    assert ab.max() < p
    for x in (1, 2, 0x123456789ABCDEF % p):
        assert horner(a, x, p) * horner(b, x, p) % p == horner(ab, x, p), x


@pytest.mark.parametrize("p,n", [(P51, 200000), (P62, 1 << 18), (P50, 150001),          # L = 2^19: one level of 128 points
                                 (P51, (1 << 19) + 3), (P51, 1 << 22), (P62, 1 << 21), (P51B, 1 << 20)])
def test_direct64_large_sizes(ctx, p, n):
    """Two and three column levels in front of the rows (L = 2^21 ... 2^23; the plan limit is n = 2^23):
    beyond the CRT oracle's exactness bound, so checked through A(x) * B(x) = AB(x) at x = 1, -1 (sums)
    and one Horner evaluation, plus the two end coefficients."""
    rng = np.random.default_rng(n % 1000)
    a = rng.integers(0, p, size=n, dtype=np.uint64)
    b = rng.integers(0, p, size=n, dtype=np.uint64)
    a[-1] = b[-1] = p - 1
    plan = ctx.plan64(p, n)
    ab = plan.multiply(a, b)[0]
    plan.close()
    assert ab.max() < p
    assert int(ab[0]) == int(a[0]) * int(b[0]) % p and int(ab[-1]) == 1

    def at1(c):
        return sum(int(v) for v in np.add.reduceat(c >> np.uint64(32), np.arange(0, c.size, 1 << 20)).tolist()) * (1 << 32) \
            + sum(int(v) for v in np.add.reduceat(c & np.uint64(0xFFFFFFFF), np.arange(0, c.size, 1 << 20)).tolist())

    def atm1(c):
        return at1(c[0::2]) - at1(c[1::2])
    assert at1(a) * at1(b) % p == at1(ab) % p
    assert atm1(a) * atm1(b) % p == atm1(ab) % p
    if n <= (1 << 20):
        x = 0x123456789ABCDEF % p
        assert horner(a, x, p) * horner(b, x, p) % p == horner(ab, x, p)


def test_direct64_edge_values(ctx):
    n = 4096
    plan = ctx.plan64(P51, n)
    top = np.full(n, P51 - 1, dtype=np.uint64)
    zero = np.zeros(n, dtype=np.uint64)
    one = np.zeros(n, dtype=np.uint64)
    one[0] = 1
    assert not plan.multiply(top, zero)[0].any()
    assert np.array_equal(plan.multiply(top, one)[0][:n], top)
    want, _ = orc.poly_multiply_crt(top, top, P51, threads=8)
    assert np.array_equal(plan.multiply(top, top)[0], want)
    plan.close()


def test_direct64_batched_device_resident(ctx):
    """[batch][n] device arrays, output row stride 2n, tail entry of every row untouched."""
    import torch
    n, B = 5000, 6
    rng = np.random.default_rng(77)
    a = rng.integers(0, P51, size=(B, n), dtype=np.uint64)
    b = rng.integers(0, P51, size=(B, n), dtype=np.uint64)
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.full((B, 2 * n), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    plan = ctx.plan64(P51, n, batch=8)
    plan.multiply(da, db, out=out, batch=B, out_stride=2 * n)
    ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(B):
        want, _ = orc.poly_multiply_crt(a[i], b[i], P51, threads=8)
        assert np.array_equal(got[i, :2 * n - 1].view(np.uint64), want), i
        assert got[i, 2 * n - 1] == -7
    # host pointers with a strided output keep the caller's tail entries too
    hout = np.full((B, 2 * n), 5, dtype=np.uint64)
    plan.multiply(a, b, out=hout, batch=B, out_stride=2 * n)
    assert np.array_equal(hout[:, :2 * n - 1], got[:, :2 * n - 1].view(np.uint64))
    assert (hout[:, 2 * n - 1] == 5).all()
    plan.close()


def test_direct64_errors(nt, ctx):
    with pytest.raises(nt.NttError):
        ctx.plan64(P51 + 2, 16)                 # not prime
    with pytest.raises(nt.NttError):
        ctx.plan64((1 << 61) - 1, 16)           # prime, but p - 1 = 2 * odd
    with pytest.raises(nt.NttError):
        ctx.plan64(P33, 1 << 20)                # 2-adicity 20 < 21
    with pytest.raises(nt.NttError):
        ctx.plan64(18446744069414584321, 16)    # NTT-friendly but above 2^62
    with pytest.raises(nt.NttError):
        ctx.plan64(P51, 4096, batch=65536)      # L = 8192: the column kernels take at most 65535 rows per launch
    plan = ctx.plan64(P51, 16, batch=2)
    a = np.zeros((3, 16), dtype=np.uint64)
    with pytest.raises(nt.NttError):
        plan.multiply(a, a, batch=3)
    plan.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_direct64_reference_golden(ctx, nttdata, cid):
    """The reference's own cases (nttdata/<id>.in/.out, 32-bit primes) through the 64-bit lanes: a third,
    independent route to the golden outputs."""
    n, p, a, b, ab = nttdata[cid]
    plan = ctx.plan64(p, n)
    got = plan.multiply(a.astype(np.uint64), b.astype(np.uint64))[0]
    plan.close()
    assert np.array_equal(got, ab.astype(np.uint64))
