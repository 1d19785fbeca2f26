"""GPU parity tests: the HIP path (through the C ABI of libntt_amd.so) against the CPU oracle,
the reference's shipped nttdata fixtures and the vectors produced by the compiled reference.
Bit-exact (integer arithmetic): every comparison is array equality.
"""
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

PRIMES = (7340033, 104857601, 469762049, 998244353)
KINDS = {"mont": 0, "barrett": 1}


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.fixture(scope="module")
def ctx(nt):
    c = nt.Context(0)
    yield c
    c.close()


def rand_poly(rng, n, p, batch=None):
    shape = (n,) if batch is None else (batch, n)
    return rng.integers(0, p, size=shape, dtype=np.int64).astype(np.uint32)


# ---- the reference's shipped cases (golden vectors that pin the path) -----------------
@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_nttdata_legacy_entry_point(nt, nttdata, cid):
    n, p, a, b, ab = nttdata[cid]
    got = nt.poly_multiply(a.astype(np.int32), b.astype(np.int32), p)
    assert np.array_equal(got.astype(np.uint32), ab)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_nttdata_plan(nt, ctx, nttdata, cid, kind):
    n, p, a, b, ab = nttdata[cid]
    plan = ctx.plan(p, n, 1, KINDS[kind])
    got = plan.multiply(a.reshape(1, n), b.reshape(1, n))
    assert np.array_equal(got[0], ab)
    plan.close()


def test_legacy_entry_leaves_tail_untouched(nt, nttdata):
    # ownership rule: entries >= 2n-1 of ab are not written
    import ctypes as C
    n, p, a, b, ab = nttdata[0]
    buf = np.full(300000, -7, dtype=np.int32)
    a32, b32 = a.astype(np.int32), b.astype(np.int32)
    nt._lib.poly_multiply(a32.ctypes.data, b32.ctypes.data, buf.ctypes.data, n, p)
    assert np.array_equal(buf[:2 * n - 1].astype(np.uint32), ab)
    assert np.all(buf[2 * n - 1:] == -7)


# ---- vectors made by the compiled reference (tests/golden/make_golden.py) --------------------------
@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_reference_polymul_vectors(ctx, ref_polymul_vectors, kind):
    v = ref_polymul_vectors
    for k in [k for k in v.files if k.startswith("ab_")]:
        _, p, n = k.split("_")
        p, n = int(p), int(n)
        plan = ctx.plan(p, n, 1, KINDS[kind])
        got = plan.multiply(v[f"a_{p}_{n}"].reshape(1, n), v[f"b_{p}_{n}"].reshape(1, n))
        assert np.array_equal(got[0], v[k]), (k, kind)
        plan.close()


@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_reference_ntt_vectors(ctx, ref_ntt_vectors, kind):
    v = ref_ntt_vectors
    for k in [k for k in v.files if k.startswith("in_")]:
        _, p, L = k.split("_")
        p, L = int(p), int(L)
        plan = ctx.plan(p, L // 2, 1, KINDS[kind])
        assert plan.length == L
        x = v[k].copy()
        assert np.array_equal(plan.forward(x.copy()), v[f"fwd_{p}_{L}"]), (k, kind)
        assert np.array_equal(plan.inverse(x.copy()), v[f"inv_{p}_{L}"]), (k, kind)
        plan.close()


# ---- seeded synthetic inputs against the oracle, all code paths ------------------------------------
SIZES = [1, 2, 3, 7, 100, 1000, 2048, 2049, 4096, 5000, 40000, 65536, 100000, 131072]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_multiply_vs_oracle(ctx, n, kind):
    rng = np.random.default_rng(1000 + n)
    for p in (998244353, 469762049, 1224736769, 7340033):    # lazy, lazy, strict (>2^30), small
        a, b = rand_poly(rng, n, p), rand_poly(rng, n, p)
        a[0] = p - 1
        b[-1] = p - 1
        plan = ctx.plan(p, n, 1, KINDS[kind])
        got = plan.multiply(a.reshape(1, n), b.reshape(1, n))[0]
        want = orc.poly_multiply(a, b, p, kind=kind)
        assert np.array_equal(got, want), (n, p, kind)
        plan.close()


def test_randomised_shapes_and_primes(ctx):
    """Seeded sweep over (prime, n, batch, kind, stride): ragged n around every path boundary
    (L = 4096 | 8192, column levels 1 | 2), NTT primes of different 2-adicity and size, odd strides."""
    rng = np.random.default_rng(20250711)
    # 2013265921 = 15 * 2^27 + 1 has smallest generator 31, not the reference's hard-coded 3: the
    # oracle's g = 3 transform is not valid there, so that prime is checked by evaluation instead
    primes = [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769, 2013265921, 167772161, 40961]
    ns = [1, 2, 6, 17, 129, 1023, 1025, 2047, 2048, 2049, 3000, 4095, 4097, 6000, 8191, 8193, 12345, 16385,
          33000, 70000, 131071, 131073, 262145]
    for trial in range(60):
        p = int(primes[rng.integers(len(primes))])
        n = int(ns[rng.integers(len(ns))])
        L = 1 << max(1, (2 * n - 1).bit_length())
        if (p - 1) % L:
            continue
        batch = int(rng.integers(1, 4)) if n < 40000 else 1
        kind = int(rng.integers(2))
        stride = 2 * n - 1 + int(rng.integers(0, 3))
        a, b = rand_poly(rng, n, p, batch), rand_poly(rng, n, p, batch)
        plan = ctx.plan(p, n, batch, kind)
        got = plan.multiply(a, b, out=np.full((batch, stride), 0xABCDEF01, dtype=np.uint32), out_stride=stride)
        for r in range(batch):
            if p == 2013265921:
                x = int(rng.integers(2, p))

                def ev(c):
                    acc = 0
                    for v in c[::-1].tolist():
                        acc = (acc * x + v) % p
                    return acc
                assert ev(a[r]) * ev(b[r]) % p == ev(got[r, :2 * n - 1]), (p, n, r, kind)
                assert got[r, :2 * n - 1].max() < p
            else:
                assert np.array_equal(got[r, :2 * n - 1], orc.poly_multiply(a[r], b[r], p, kind="barrett")), (p, n, r, kind)
        assert np.all(got[:, 2 * n - 1:] == 0xABCDEF01), (p, n)
        plan.close()


@pytest.mark.parametrize("seq_mode", ["1", "0"])
def test_one_kernel_medium_products(ctx, seq_mode, monkeypatch):
    """L = 8192 and 16384: whole products in one kernel with the rows one after the other (seq_product_kernel,
    kernels_fused.hip; NTT_AMD_FUSED=0 keeps the three kernels on the same inputs) -- ragged n
    on both sides of every boundary, batches, odd strides, both reductions, lazy / strict / small primes."""
    monkeypatch.setenv("NTT_AMD_FUSED", seq_mode)
    rng = np.random.default_rng(77)
    for n in (2049, 3000, 4096, 4097, 5000, 6143, 8191, 8192):
        for p, kind in ((998244353, 0), (469762049, 1), (1224736769, 0), (1224736769, 1), (7340033, 0)):
            B = 3
            stride = 2 * n - 1 + (n % 3)
            a, b = rand_poly(rng, n, p, B), rand_poly(rng, n, p, B)
            a[0, 0] = a[1, -1] = b[1, -1] = p - 1
            plan = ctx.plan(p, n, B, kind)
            got = plan.multiply(a, b, out=np.full((B, stride), 0x5A5A5A5A, dtype=np.uint32), out_stride=stride)
            plan.close()
            for r in range(B):
                assert np.array_equal(got[r, :2 * n - 1], orc.poly_multiply(a[r], b[r], p, kind="barrett")), (n, p, kind, r)
            assert np.all(got[:, 2 * n - 1:] == 0x5A5A5A5A), (n, p, kind)


def test_one_kernel_vs_three_kernels_random(ctx, monkeypatch):
    """Seeded sweep, n in (2048, 8192], linear and negacyclic: the one-kernel product (default) against the
    three-kernel path (NTT_AMD_FUSED=0) bit for bit on every row, and one row per trial against the oracle."""
    rng = np.random.default_rng(424242)
    primes = [998244353, 469762049, 1224736769, 7340033, 2013265921, 167772161]
    for trial in range(40):
        p = int(primes[rng.integers(len(primes))])
        neg = bool(rng.integers(2))
        n = int(1 << rng.integers(13, 15)) if neg else int(rng.integers(2049, 8193))
        if neg and n == 16384:
            monkeypatch.setenv("NTT_AMD_FUSED", "2")      # the full-length L = 16384 is not the default route
        else:
            monkeypatch.delenv("NTT_AMD_FUSED", raising=False)
        B = int(rng.integers(1, 6))
        kind = int(rng.integers(2))
        cols = n if neg else 2 * n - 1
        stride = cols + int(rng.integers(0, 4))
        a, b = rand_poly(rng, n, p, B), rand_poly(rng, n, p, B)
        mk = (lambda: ctx.negacyclic_plan(p, n, B, kind)) if neg else (lambda: ctx.plan(p, n, B, kind))
        plan = mk()
        one = plan.multiply(a, b, out=np.full((B, stride), 0x11111111, dtype=np.uint32), out_stride=stride)
        plan.close()
        monkeypatch.setenv("NTT_AMD_FUSED", "0")
        plan = mk()
        three = plan.multiply(a, b, out=np.full((B, stride), 0x11111111, dtype=np.uint32), out_stride=stride)
        plan.close()
        assert np.array_equal(one, three), (trial, p, n, neg, B, kind)
        if p != 2013265921:                               # (generator 31: the oracle's g = 3 transform is not valid there)
            r = int(rng.integers(B))
            ab = np.asarray(orc.poly_multiply(a[r], b[r], p, kind="barrett")).astype(np.int64)
            if neg:
                hi = np.zeros(n, dtype=np.int64)
                hi[:n - 1] = ab[n:]
                ab = (ab[:n] - hi) % p
            assert np.array_equal(one[r, :cols], ab.astype(np.uint32)), (trial, p, n, neg, r)


def test_multi_level_columns(ctx):
    # L = 2^19 and 2^20: two column levels in front of the 4096-point rows
    rng = np.random.default_rng(5)
    for n, p in ((1 << 18, 998244353), (300000, 469762049), (1 << 19, 7340033)):
        a, b = rand_poly(rng, n, p), rand_poly(rng, n, p)
        plan = ctx.plan(p, n, 1, 0)
        got = plan.multiply(a.reshape(1, n), b.reshape(1, n))[0]
        assert np.array_equal(got, orc.poly_multiply(a, b, p, kind="barrett")), (n, p)
        plan.close()


def test_multi_level_columns_batched(ctx):
    """Two column levels with a batch, a ragged n, an odd output stride and both reductions."""
    rng = np.random.default_rng(6)
    n, p, B = 300001, 469762049, 3
    stride = 2 * n + 5
    a, b = rand_poly(rng, n, p, B), rand_poly(rng, n, p, B)
    for kind in (0, 1):
        plan = ctx.plan(p, n, B, kind)
        got = plan.multiply(a, b, out=np.full((B, stride), 7, dtype=np.uint32), out_stride=stride)
        plan.close()
        for r in range(B):
            assert np.array_equal(got[r, :2 * n - 1], orc.poly_multiply(a[r], b[r], p, kind="barrett")), (kind, r)
        assert np.all(got[:, 2 * n - 1:] == 7)


def test_largest_product_of_the_headline_prime(ctx):
    """n = 2^22: L = 2^23 is the whole 2-adicity of 998244353 (three column levels + rows); the reference's
    own arrays stop at 300000 entries (main-montgomery.cc:297), far below.  Checked against the oracle's
    Barrett body (power-of-two padding, main-barrett.cc): the Montgomery body pads to a power of FOUR
    (main-montgomery.cc:317-321), would need 2^24 | p - 1 here and is not a valid transform at this size."""
    rng = np.random.default_rng(22)
    n, p = 1 << 22, 998244353
    a, b = rand_poly(rng, n, p), rand_poly(rng, n, p)
    a[-1] = b[-1] = p - 1
    plan = ctx.plan(p, n, 1, 0)
    got = plan.multiply(a.reshape(1, n), b.reshape(1, n))[0]
    plan.close()
    assert np.array_equal(got, orc.poly_multiply(a, b, p, kind="barrett"))
    assert got[-1] == 1                                   # (p - 1)^2


def test_edge_values(ctx):
    p = 998244353
    for n in (4, 4096, 8192):
        plan = ctx.plan(p, n, 1, 0)
        for a, b in ((np.zeros(n), np.zeros(n)), (np.full(n, p - 1), np.full(n, p - 1)),
                     (np.ones(n), np.full(n, p - 1)), (np.eye(1, n, n - 1)[0], np.eye(1, n, n - 1)[0])):
            a, b = a.astype(np.uint32), b.astype(np.uint32)
            got = plan.multiply(a.reshape(1, n), b.reshape(1, n))[0]
            assert np.array_equal(got, orc.poly_multiply(a, b, p)), n
        plan.close()


def test_forced_strict_path_small_prime(nt, ctx):
    # the [0,p) (non-lazy) kernels are normally used only for p >= 2^30: force them for a small prime
    os.environ["NTT_AMD_FORCE_STRICT"] = "1"
    try:
        rng = np.random.default_rng(9)
        for n in (1000, 10000):
            for kind in ("mont", "barrett"):
                p = 998244353
                a, b = rand_poly(rng, n, p), rand_poly(rng, n, p)
                plan = ctx.plan(p, n, 1, KINDS[kind])
                assert np.array_equal(plan.multiply(a.reshape(1, n), b.reshape(1, n))[0],
                                      orc.poly_multiply(a, b, p))
                plan.close()
    finally:
        del os.environ["NTT_AMD_FORCE_STRICT"]


@pytest.mark.parametrize("kind", ["mont", "barrett"])
def test_forward_inverse_vs_oracle(ctx, kind):
    rng = np.random.default_rng(11)
    for p, L in ((998244353, 8192), (469762049, 1 << 18), (1224736769, 1 << 14), (7340033, 1 << 20),
                 (998244353, 64), (998244353, 2)):
        batch = 3 if L <= (1 << 14) else 1
        x = rand_poly(rng, L, p, batch)
        plan = ctx.plan(p, L // 2, batch, KINDS[kind])
        f = plan.forward(x.copy(), batch)
        for r in range(batch):
            assert np.array_equal(f[r], orc.ntt(x[r], p)), (p, L, r)
        i = plan.inverse(x.copy(), batch)
        for r in range(batch):
            assert np.array_equal(i[r], orc.ntt(x[r], p, inverse=True)), (p, L, r)
        assert np.array_equal(plan.inverse(f, batch), x)
        plan.close()


# ---- batched, device-resident, strided output ------------------------------------------------------
def test_batched_device_resident(ctx):
    import torch
    rng = np.random.default_rng(21)
    for n, batch, p in ((131072, 5, 998244353), (65536, 9, 469762049), (1000, 33, 998244353)):
        a, b = rand_poly(rng, n, p, batch), rand_poly(rng, n, p, batch)
        ta = torch.from_numpy(a.view(np.int32)).cuda()
        tb = torch.from_numpy(b.view(np.int32)).cuda()
        stride = 2 * n
        tab = torch.full((batch, stride), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        plan = ctx.plan(p, n, batch, 0)
        plan.multiply(ta, tb, out=tab, batch=batch, out_stride=stride)
        ctx.synchronize()
        got = tab.cpu().numpy().view(np.uint32)
        for r in range(batch):
            assert np.array_equal(got[r, :2 * n - 1], orc.poly_multiply(a[r], b[r], p)), (n, r)
        assert np.all(got[:, 2 * n - 1] == 0xFFFFFFFF)          # pad column untouched
        # inputs are not modified
        assert np.array_equal(ta.cpu().numpy().view(np.uint32), a)
        plan.close()


def test_chunking_is_transparent(nt, ctx):
    # the work buffers hold `chunk` products; results must not depend on the chunk size
    rng = np.random.default_rng(22)
    n, batch, p = 8192, 7, 998244353
    a, b = rand_poly(rng, n, p, batch), rand_poly(rng, n, p, batch)
    outs = []
    for chunk in ("1", "3", "7"):
        os.environ["NTT_AMD_CHUNK"] = chunk
        plan = ctx.plan(p, n, batch, 0)
        outs.append(plan.multiply(a, b))
        plan.close()
    del os.environ["NTT_AMD_CHUNK"]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.array_equal(outs[0][6], orc.poly_multiply(a[6], b[6], p))


# ---- size-independent properties at the benchmark size ---------------------------------------------
def test_full_size_properties(ctx):
    """n = 2^17, p = 998244353, batch 64 (BASELINE configs[1] shape): checks that need no oracle run
    per row: multiplying by the constant 1 is the identity, by x^k is a shift, the product commutes,
    and the coefficient sum equals the product of the coefficient sums (evaluation at x = 1)."""
    import torch
    n, batch, p = 131072, 64, 998244353
    a = (orc.splitmix64_np(0x20250711, batch * n) % np.uint64(p)).astype(np.uint32).reshape(batch, n)
    b = (orc.splitmix64_np(0x20250711 + 77, batch * n) % np.uint64(p)).astype(np.uint32).reshape(batch, n)
    plan = ctx.plan(p, n, batch, 0)
    one = np.zeros((batch, n), dtype=np.uint32)
    one[:, 0] = 1
    got = plan.multiply(a, one)
    assert np.array_equal(got[:, :n], a) and not got[:, n:].any()
    shift = np.zeros((batch, n), dtype=np.uint32)
    shift[:, 12345] = 1
    got = plan.multiply(a, shift)
    assert np.array_equal(got[:, 12345:12345 + n], a) and not got[:, :12345].any()
    ab, ba = plan.multiply(a, b), plan.multiply(b, a)
    assert np.array_equal(ab, ba)
    sa = a.astype(np.uint64).sum(axis=1) % p
    sb = b.astype(np.uint64).sum(axis=1) % p
    sab = ab.astype(np.uint64).sum(axis=1) % p
    assert np.array_equal(sab, sa * sb % p)
    # and two rows against the oracle
    for r in (0, batch - 1):
        assert np.array_equal(ab[r], orc.poly_multiply(a[r], b[r], p))
    plan.close()


# ---- BASELINE.json configs[3]: FHE-style RNS batch (n = 2^16, 32 limbs) ----------------------------
def rns_primes(count, two_adicity, above):
    """The `count` smallest primes p > above with p = 1 mod 2^two_adicity."""
    out, step = [], 1 << two_adicity
    p = (above // step + 1) * step + 1
    while len(out) < count:
        if all(p % q for q in range(3, int(p ** 0.5) + 1, 2)):
            out.append(p)
        p += step
    return out


def test_rns_batch_config4(ctx):
    """n = 65536, 32 distinct primes p = 1 (mod 2^17) just above 2^30 (generator found per prime, strict
    [0,p) kernels since p >= 2^30), a few polynomials per limb: every row against the oracle's plain
    NTT product, which works for any generator because results are canonical."""
    n = 65536
    primes = rns_primes(32, 17, 1 << 30)
    assert len(set(primes)) == 32 and all((p - 1) % (1 << 17) == 0 and (1 << 30) < p < (1 << 31) for p in primes)
    rng = np.random.default_rng(44)
    for li, p in enumerate(primes):
        batch = 3 if li < 4 else 1
        a, b = rand_poly(rng, n, p, batch), rand_poly(rng, n, p, batch)
        plan = ctx.plan(p, n, batch, li % 2)              # alternate Montgomery / Barrett
        got = plan.multiply(a, b)
        # the oracle's NTT hard-codes g = 3 (main-montgomery.cc:257), which is not a generator for
        # every such prime, so check the evaluation identity A(x) * B(x) = AB(x) (mod p) at a
        # random point per row, plus exact convolution coefficients for one limb below
        for r in range(batch):
            x = int(rng.integers(2, p))

            def ev(c):                     # Horner in Python ints (exact)
                acc = 0
                for v in c[::-1].tolist():
                    acc = (acc * x + v) % p
                return acc
            assert ev(a[r]) * ev(b[r]) % p == ev(got[r]), (p, r, x)
        plan.close()
    # one limb fully, against an exact convolution on a short slice (first 64 and last 64 coefficients)
    p = primes[0]
    a, b = rand_poly(rng, n, p, 1), rand_poly(rng, n, p, 1)
    plan = ctx.plan(p, n, 1, 0)
    got = plan.multiply(a, b)[0]
    ai, bi = [int(v) for v in a[0]], [int(v) for v in b[0]]
    for k in list(range(64)) + list(range(2 * n - 1 - 64, 2 * n - 1)):
        lo, hi = max(0, k - (n - 1)), min(k, n - 1)
        assert sum(ai[i] * bi[k - i] for i in range(lo, hi + 1)) % p == int(got[k]), k
    plan.close()


# ---- BASELINE.json configs[4] on one GPU: a single transform of length 2^26 ------------------------
def test_single_transform_2pow26(nt, ctx):
    """L = 2^26, p = 469762049 (2-adicity 26): three column levels + 4096-point rows on ONE GPU (the
    8-GPU all-to-all split is a later row).  Checks the round-trip identity and 64 output points
    against direct O(L) evaluation X[k] = sum x[i] w^(ik)."""
    p, lg = 469762049, 26
    L = 1 << lg
    x = (orc.splitmix64_np(0x20250711, L) % np.uint64(p)).astype(np.uint32)
    plan = ctx.plan(p, L // 2, 1, 0)
    assert plan.length == L
    f = plan.forward(x.copy().reshape(1, L), 1)[0]
    w = pow(3, (p - 1) // L, p)
    rng = np.random.default_rng(5)
    ks = [0, 1, L - 1, L // 2, 4096, 4097] + [int(v) for v in rng.integers(0, L, size=58)]
    H = 1 << 13                                      # i = hi * 2^13 + lo: w^(ik) = A[hi] * B[lo]
    xm = x.reshape(H, H).astype(np.uint64)
    P = np.uint64(p)
    for k in ks:
        wk = pow(w, k, p)
        wkH = pow(wk, H, p)
        A, B = np.empty(H, dtype=np.uint64), np.empty(H, dtype=np.uint64)
        ca = cb = 1
        for j in range(H):
            A[j], B[j] = ca, cb
            ca, cb = ca * wkH % p, cb * wk % p
        rows = (xm * B % P).sum(axis=1) % P          # 8192 terms < 2^29 each
        want = int((rows * A % P).sum() % P)
        assert int(f[k]) == want, k
    # the same transform through the 8-rank decomposition of BASELINE configs[4] (virtual ranks in this
    # process, exchanges by tensor shuffles): rank r must end with X[r + 8*k2]
    import torch
    G = 8
    S, cols = L // G, L // G // G
    dplans = [nt.DistPlan(ctx, p, lg, G, r, 0) for r in range(G)]
    dx = torch.from_numpy(x.view(np.int32)).cuda()
    held = [torch.stack([dx[s * S + d * cols: s * S + (d + 1) * cols] for s in range(G)]).contiguous() for d in range(G)]
    del dx
    torch.cuda.synchronize()
    for r in range(G):
        dplans[r].colpass(held[r], False)
    ctx.synchronize()
    rows = [torch.cat([held[s][d] for s in range(G)]).contiguous() for d in range(G)]
    del held
    torch.cuda.synchronize()
    for r in range(G):
        dplans[r].local(rows[r], False)
    ctx.synchronize()
    for r in range(G):
        assert np.array_equal(rows[r].cpu().numpy().view(np.uint32), f[r::G]), r
    for dp in dplans:
        dp.close()
    del rows
    back = plan.inverse(f.reshape(1, L), 1)[0]
    assert np.array_equal(back, x)
    plan.close()


# ---- large modulus through the 4-prime CRT -----------------------------------------------------------
def test_crt_golden(ctx, crt_vectors):
    v = crt_vectors
    p64 = int(v["p64"])
    for n in (256, 1000):
        got = ctx.crt_multiply(v[f"a_{n}"], v[f"b_{n}"], p64)
        assert np.array_equal(got, v[f"ab_{n}"]), n


def test_crt_vs_oracle_large(ctx):
    p64 = 1337006139375617
    rng = np.random.default_rng(33)
    for n in (5, 5000, 131072):
        a = rng.integers(0, p64, size=n, dtype=np.uint64)
        b = rng.integers(0, p64, size=n, dtype=np.uint64)
        a[0] = p64 - 1
        b[0] = p64 - 1
        want, _ = orc.poly_multiply_crt(a, b, p64, threads=8)
        assert np.array_equal(ctx.crt_multiply(a, b, p64), want), n
    # the older modulus of the reference comments (main-montgomery.cc:302)
    p64 = 263882790666241
    a = rng.integers(0, p64, size=3000, dtype=np.uint64)
    b = rng.integers(0, p64, size=3000, dtype=np.uint64)
    want, _ = orc.poly_multiply_crt(a, b, p64, threads=4)
    assert np.array_equal(ctx.crt_multiply(a, b, p64), want)


def test_crt_merge_moduli_of_every_size(ctx):
    """The merge takes two routes (crt_device.hpp): p64 >= 2^33 sums the mixed-radix digits as one 96-bit integer
    and reduces with two 32-bit Montgomery steps, smaller moduli keep Horner with 64-bit Montgomery products.  Moduli
    of 20 to 58 bits (need not be prime), extreme coefficients, against Python big-int schoolbook; the exactness guard
    n (p64 - 1)^2 < M bounds what n may be."""
    rng = np.random.default_rng(58)
    M = 1004535809 * 1224736769 * 469762049 * 998244353
    for p64 in (1048573, (1 << 31) - 1, (1 << 32) + 15, (1 << 33) - 9, (1 << 33) + 17, (1 << 40) + 15, 263882790666241,
                1337006139375617, (1 << 51) - 129, (1 << 55) + 3, (1 << 58) - 27):
        n = 6
        while (n + 1) * (p64 - 1) ** 2 < M and n < 40:
            n += 1
        a = rng.integers(0, p64, size=n, dtype=np.uint64)
        b = rng.integers(0, p64, size=n, dtype=np.uint64)
        a[0] = b[0] = a[-1] = p64 - 1
        b[-1] = 0
        want = [0] * (2 * n - 1)
        for i in range(n):
            for j in range(n):
                want[i + j] = (want[i + j] + int(a[i]) * int(b[j])) % p64
        got = ctx.crt_multiply(a, b, p64)
        assert [int(x) for x in got] == want, p64


def test_crt_batched(ctx, crt_vectors):
    """ntt_crt_multiply_batched: host and device arrays, every row against the single-product call and the
    oracle; switching between batch sizes and back to single products reuses / rebuilds the cached plans."""
    import torch
    p64 = 1337006139375617
    rng = np.random.default_rng(12)
    n, B = 3000, 7
    a = rng.integers(0, p64, size=(B, n), dtype=np.uint64)
    b = rng.integers(0, p64, size=(B, n), dtype=np.uint64)
    got = ctx.crt_multiply_batched(a, b, p64)
    for i in range(B):
        want, _ = orc.poly_multiply_crt(a[i], b[i], p64, threads=4)
        assert np.array_equal(got[i], want), i
    assert np.array_equal(ctx.crt_multiply(a[2], b[2], p64), got[2])               # back to one product
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.zeros((B, 2 * n - 1), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.crt_multiply_batched(da[:3].contiguous(), db[:3].contiguous(), p64, out=out[:3])     # smaller batch, cached plans
    ctx.synchronize()
    assert np.array_equal(out[:3].cpu().numpy().view(np.uint64), got[:3])
    v = crt_vectors                                                                  # golden big-integer vectors
    g = ctx.crt_multiply_batched(np.stack([v["a_256"], v["a_256"]]), np.stack([v["b_256"], v["b_256"]]), int(v["p64"]))
    assert np.array_equal(g[0], v["ab_256"]) and np.array_equal(g[1], v["ab_256"])


def test_crt_sharded_world1(nt, ctx):
    """The prime-per-GPU CRT driver (ntt_amd.shard.crt_multiply_sharded) fed by the real C-ABI halves
    ntt_crt_residue_multiply / ntt_crt_merge, world_size 1 (the exchange itself is covered by the gloo
    tests on CPU); also the halves on host pointers against the golden residues."""
    import torch
    from ntt_amd import shard
    assert [nt.crt_prime(k) for k in range(4)] == list(orc.CRT_MODS)
    p64, n = 1337006139375617, 5000
    rng = np.random.default_rng(8)
    a = rng.integers(0, p64, size=n, dtype=np.uint64)
    b = rng.integers(0, p64, size=n, dtype=np.uint64)
    want, res, _ = orc.poly_multiply_crt(a, b, p64, threads=4, want_residues=True)
    rows = [ctx.crt_residue_multiply(a, b, k) for k in range(4)]           # host pointers
    for k in range(4):
        assert np.array_equal(rows[k], res[k, :2 * n - 1].astype(np.uint32)), k
    assert np.array_equal(ctx.crt_merge(rows, p64), want)
    assert np.array_equal(ctx.crt_merge([r[100:200].copy() for r in rows], p64), want[100:200])   # any slice
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()

    # the library works on its own stream: order it against torch's stream on both sides of each call
    def residue_fn(k):
        out = torch.zeros(2 * n - 1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.crt_residue_multiply(da, db, k, out=out)
        ctx.synchronize()
        return out

    def merge_fn(r4, count):
        out = torch.zeros(count, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.crt_merge(r4, p64, out=out, count=count)
        ctx.synchronize()
        return out

    g = shard.Group("nccl", device=torch.device("cuda", 0))       # WORLD_SIZE unset: no process group
    got = shard.crt_multiply_sharded(g, residue_fn, merge_fn, n, device="cuda")
    ctx.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want)
    # the per-epoch form: buffers owned by the exchange object, library on torch's stream, nothing allocated and
    # no host synchronisation inside an epoch (main-mpi.cc:706-802 is the loop it serves)
    xchg = shard.CrtExchange(g, ctx, n, p64, device="cuda")
    try:
        first = xchg.multiply(da, db).clone()
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        for _ in range(3):
            out = xchg.multiply(da, db)
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want) and torch.equal(out, first)
    finally:
        ctx.set_stream(None)                                      # the module's context goes back to its own stream


def test_dist_transform_virtual_ranks(nt, ctx):
    """The per-rank GPU pieces of the distributed transform (ntt_dist_colpass / ntt_dist_local) for
    every rank of a virtual world in ONE process: the two all-to-all exchanges are done by tensor
    shuffles here (their RCCL form is ntt_amd.shard.dist_transform, covered over gloo on CPU).
    world = 8 is BASELINE configs[4]'s node size; 32 exercises the LDS-staged column kernel."""
    import torch
    for p, lgL, G, kind in ((469762049, 18, 8, 0), (998244353, 16, 4, 1), (469762049, 18, 32, 0),
                            (1224736769, 17, 2, 0)):
        L = 1 << lgL
        S, cols = L // G, L // G // G
        x = (orc.splitmix64_np(lgL * 100 + G, L) % np.uint64(p)).astype(np.uint32)
        X = orc.ntt(x, p)
        plans = [nt.DistPlan(ctx, p, lgL, G, r, kind) for r in range(G)]
        assert plans[0].local_length == S

        def a2a(sends):           # sends[s][d] goes to rank d; rank d holds stack over s
            return [torch.stack([sends[s][d] for s in range(G)]).contiguous() for d in range(G)]

        dx = torch.from_numpy(x.view(np.int32)).cuda()
        held = a2a([dx[r * S:(r + 1) * S].reshape(G, cols) for r in range(G)])
        torch.cuda.synchronize()
        for r in range(G):
            plans[r].colpass(held[r], False)
        ctx.synchronize()
        rows = [h.reshape(S).contiguous() for h in a2a(held)]
        torch.cuda.synchronize()
        for r in range(G):
            plans[r].local(rows[r], False)
        ctx.synchronize()
        for r in range(G):
            assert np.array_equal(rows[r].cpu().numpy().view(np.uint32), X[r::G]), (p, lgL, G, r)
        # inverse from the transposed layout back to natural blocks
        for r in range(G):
            plans[r].local(rows[r], True)
        ctx.synchronize()
        held = a2a([rows[r].reshape(G, cols) for r in range(G)])
        torch.cuda.synchronize()
        for r in range(G):
            plans[r].colpass(held[r], True)
        ctx.synchronize()
        back = torch.cat([h.reshape(S) for h in a2a(held)])
        assert np.array_equal(back.cpu().numpy().view(np.uint32), x), (p, lgL, G)
        for pl in plans:
            pl.close()
    with pytest.raises(nt.NttError):
        nt.DistPlan(ctx, 998244353, 12, 8, 0)      # L < 256 * world^2
    with pytest.raises(nt.NttError):
        nt.DistPlan(ctx, 998244353, 20, 6, 0)      # world not a power of two


# ---- the harness loop as a program: ntt_cli on the reference's text formats ---------------------------
def test_cli_harness_loop(nt, nttdata, tmp_path):
    """ntt_amd/ntt_cli = main() of main-montgomery.cc:298-362 on top of the C ABI: reads <id>.in, multiplies
    on the GPU, checks against <id>.out, prints the reference's verdict and latency lines, writes files/<id>.out."""
    import subprocess
    data, out = tmp_path / "nttdata", tmp_path / "files"
    data.mkdir()
    out.mkdir()
    for cid in (0, 1, 3):
        n, p, a, b, ab = nttdata[cid]
        with open(data / f"{cid}.in", "w") as f:          # the reference's layout
            f.write(f"{n} {p}\n" + "".join(f"{v} " for v in a) + "\n" + "".join(f"{v} " for v in b))
        with open(data / f"{cid}.out", "w") as f:
            f.write("".join(f"{v} " for v in ab))
    # a 51-bit case for the CRT branch (nttdata/4.* is not shipped: synthetic, expected from the oracle)
    p64, n4 = 1337006139375617, 1000
    rng = np.random.default_rng(4)
    a4 = rng.integers(0, p64, size=n4, dtype=np.uint64)
    b4 = rng.integers(0, p64, size=n4, dtype=np.uint64)
    ab4, _ = orc.poly_multiply_crt(a4, b4, p64, threads=2)
    with open(data / "4.in", "w") as f:
        f.write(f"{n4} {p64}\n" + "".join(f"{v} " for v in a4) + "\n" + "".join(f"{v} " for v in b4))
    with open(data / "4.out", "w") as f:
        f.write("".join(f"{v} " for v in ab4))
    exe = os.path.join(os.path.dirname(nt.__file__), "ntt_cli")
    for extra in ([], ["--barrett"]):
        r = subprocess.run([exe, "--data", str(data), "--out", str(out), "--begin", "0", "--end", "4",
                            "--epochs", "3"] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        assert lines.count("多项式乘法结果正确") == 4 and "多项式乘法结果错误" not in r.stdout
        assert any(l.startswith("average latency for n = 131072 p = 469762049 : ") and l.endswith(" (us) ")
                   for l in r.stdout.split("\n"))
        assert "case 2 skipped" in r.stderr                 # missing file: reported, no spurious verdict
        n, p, a, b, ab = nttdata[1]
        got = np.loadtxt(out / "1.out", dtype=np.uint64)   # fWrite format: one value per line
        assert np.array_equal(got.astype(np.uint32), ab)
        assert np.array_equal(np.loadtxt(out / "4.out", dtype=np.uint64), ab4)
    # packed cases: --pack converts the text files, --bin runs the same loop on them; --direct64 sends the
    # 51-bit case through one 64-bit transform instead of the 4-prime CRT
    r = subprocess.run([exe, "--data", str(data), "--pack", "--begin", "0", "--end", "4"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 1 and "case 2 not packed" in r.stderr      # 2.in is missing, the others convert
    assert (data / "1.inb").exists() and (data / "4.outb").exists()
    r = subprocess.run([exe, "--data", str(data), "--out", str(out), "--begin", "0", "--end", "4", "--epochs", "2",
                        "--bin", "--direct64"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("多项式乘法结果正确") == 4 and "多项式乘法结果错误" not in r.stdout
    assert nt.check_case_bin(out, 4, ab4) is True                     # the written packed result
    assert nt.check_case_bin(out, 3, nttdata[3][4].astype(np.uint32)) is True
    # a wrong expected file must flip the verdict and the exit status
    with open(data / "0.out", "w") as f:
        f.write("4 21 30 48 39 30 9 ")
    r = subprocess.run([exe, "--data", str(data), "--out", str(out), "--begin", "0", "--end", "0", "--epochs", "1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "多项式乘法结果错误" in r.stdout


# ---- error behaviour -----------------------------------------------------------------------------------
def test_errors(nt, ctx):
    with pytest.raises(nt.NttError) as e:
        ctx.plan(998244353, 1 << 23, 1)          # L = 2^24 > 2-adicity 23
    assert e.value.code == nt.ERR_UNSUPPORTED
    with pytest.raises(nt.NttError):
        ctx.plan(998244351, 16, 1)               # not prime
    with pytest.raises(nt.NttError):
        ctx.plan(998244353, 0, 1)
    with pytest.raises(nt.NttError):
        ctx.crt_multiply(np.zeros(4, np.uint64), np.zeros(4, np.uint64), 1 << 61)   # even / too big
    plan = ctx.plan(998244353, 16, 1)
    with pytest.raises(nt.NttError):
        plan.multiply(np.zeros((1, 16), np.uint32), np.zeros((1, 16), np.uint32), out=np.zeros(31, np.uint32),
                      out_stride=10)          # stride below the row length: the library's own check
    with pytest.raises(nt.NttError):               # more products than the plan was sized for
        plan.multiply(np.zeros((2, 16), np.uint32), np.zeros((2, 16), np.uint32))
    plan.close()


def test_kernel_timing(ctx):
    n, p = 65536, 998244353
    a = np.ones((2, n), dtype=np.uint32)
    plan = ctx.plan(p, n, 2, 0)
    ctx.enable_kernel_timing(True)
    plan.multiply(a, a)
    t = ctx.kernel_times()
    ctx.enable_kernel_timing(False)
    assert t["col_fwd"][1] == 1 and t["row_mul"][1] == 1 and t["col_inv"][1] == 1
    assert t["row_mul"][0] > 0
    # events for one kernel only (what bench.py's timed region uses), then for all again
    ctx.kernel_timing_select("row_mul")
    ctx.enable_kernel_timing(True)
    plan.multiply(a, a)
    plan.multiply(a, a)
    t = ctx.kernel_times()
    assert t["row_mul"][1] == 2 and t["col_fwd"][1] == 0 and t["col_inv"][1] == 0
    ctx.kernel_timing_select(None)
    ctx.enable_kernel_timing(True)
    plan.multiply(a, a)
    t = ctx.kernel_times()
    ctx.enable_kernel_timing(False)
    assert t["row_mul"][1] == 1 and t["col_fwd"][1] == 1
    with pytest.raises(ValueError):
        ctx.kernel_timing_select("no_such_kernel")
    plan.close()


def test_concurrent_callers(nt, ctx):
    """Calls on one handle are serialised by its mutex; separate contexts run side by side.  Four Python
    threads (ctypes drops the GIL inside the C ABI) multiply through a shared context and through a second
    one at the same time; every result must be the oracle's."""
    import threading
    n, p, B = 8192, 998244353, 4
    rng = np.random.default_rng(21)
    jobs = []
    for t in range(4):
        a, b = rand_poly(rng, n, p, B), rand_poly(rng, n, p, B)
        jobs.append((a, b, [orc.poly_multiply(a[i], b[i], p) for i in range(B)]))
    ctx2 = nt.Context(0)
    plans = [ctx.plan(p, n, B), ctx.plan(p, n, B, nt.BARRETT), ctx2.plan(p, n, B), ctx2.plan(p, n, B)]
    errors = []

    def work(t):
        try:
            for _ in range(10):
                got = plans[t].multiply(jobs[t][0], jobs[t][1])
                for i in range(B):
                    if not np.array_equal(got[i], jobs[t][2][i]):
                        errors.append((t, i))
        except Exception as e:                      # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for pl in plans:
        pl.close()
    ctx2.close()
    assert not errors, errors


def test_binding_checks_shapes_before_launching(nt, ctx):
    """The Python binding refuses what would make a kernel read or write out of bounds: short arrays,
    non-contiguous tensors, wrong element width -- before any launch."""
    import torch
    n = 4096
    plan = ctx.plan(998244353, n, 2)
    a = np.ones((2, n), dtype=np.uint32)
    with pytest.raises(ValueError):
        plan.multiply(a, a[:1], batch=2)                       # b holds one row
    with pytest.raises(ValueError):
        plan.multiply(a, a, out=np.zeros(2 * n - 1, dtype=np.uint32), batch=2)
    with pytest.raises(TypeError):
        plan.multiply(a.astype(np.uint64), a, batch=2)
    with pytest.raises(TypeError):
        plan.multiply(np.ones((2, 2 * n), dtype=np.uint32)[:, ::2], a, batch=2)
    t = torch.ones((2, 2 * n), dtype=torch.int32, device="cuda")
    with pytest.raises(TypeError):
        plan.multiply(t[:, ::2], t[:, ::2], batch=2)           # strided device tensor
    with pytest.raises(TypeError):
        plan.multiply(t.to(torch.int64), t.to(torch.int64), batch=2)
    with pytest.raises(ValueError):
        plan.multiply(t[:1, :n].contiguous(), t[:1, :n].contiguous(), batch=2)
    with pytest.raises(ValueError):
        plan.forward(np.ones(plan.length - 1, dtype=np.uint32))
    plan.close()
    p64 = ctx.plan64(1337006139375617, n)
    with pytest.raises(ValueError):
        p64.multiply(np.ones(n, dtype=np.uint64), np.ones(n - 1, dtype=np.uint64))
    p64.close()
    with pytest.raises(ValueError):
        ctx.crt_multiply(np.ones(n, dtype=np.uint64), np.ones(n, dtype=np.uint64), 1337006139375617,
                         out=np.zeros(n, dtype=np.uint64))


def test_caller_stream(nt):
    """ntt_ctx_set_stream / ntt_ctx_stream: the library enqueues on a stream of the caller's (torch's) and is
    then ordered with the caller's own work by the stream alone -- inputs produced by torch kernels just before
    the call, the result consumed by a torch kernel just after, no synchronisation in between; with the
    lanes' fork and join (a batch of several chunks) on the caller's stream as well."""
    import torch
    c = nt.Context(0)
    own = c.stream
    side = torch.cuda.Stream()
    c.set_stream(side.cuda_stream)
    assert c.stream == side.cuda_stream
    n, p, B = 131072, 998244353, 6
    rng = np.random.default_rng(8)
    a, b = rand_poly(rng, n, p, B), rand_poly(rng, n, p, B)
    import os
    os.environ["NTT_AMD_CHUNK"] = "2"                           # three chunks, two lanes
    try:
        plan = c.plan(p, n, B)
    finally:
        del os.environ["NTT_AMD_CHUNK"]
    ha, hb = torch.from_numpy(a.view(np.int32)), torch.from_numpy(b.view(np.int32))
    with torch.cuda.stream(side):
        da = ha.cuda(non_blocking=False) + 0                    # produced by a torch kernel on `side`
        db = hb.cuda(non_blocking=False) + 0
        out = torch.zeros((B, 2 * n - 1), dtype=torch.int32, device="cuda")
        plan.multiply(da, db, out=out, batch=B)
        res = out.clone()                                       # consumed on `side` right behind the library's kernels
    side.synchronize()
    got = res.cpu().numpy().view(np.uint32)
    for r in range(B):
        assert np.array_equal(got[r], orc.poly_multiply(a[r], b[r], p)), r
    c.set_stream(0)                                             # back to the context's own stream
    assert c.stream == own
    plan.close()
    c.close()


def test_library_on_torch_stream(nt):
    """Context.use_torch_stream(): torch's ops and the library's kernels queue on one stream (torch's default
    stream is the null stream, handed over as hipStreamLegacy), so a product may consume a tensor torch is still
    producing and torch may consume the product, with no host synchronisation in between -- the ordering
    ntt_amd.launch and shard.CrtExchange rely on."""
    import torch
    c = nt.Context(0)
    c.use_torch_stream()
    n, batch, p = 65536, 24, 998244353
    plan = c.plan(p, n, batch)
    ref_ctx = nt.Context(0)                                  # its own stream, synchronised by hand: the reference
    ref_plan = ref_ctx.plan(p, n, batch)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    base = torch.randint(0, p, (batch, n), generator=gen, device="cuda", dtype=torch.int64)
    other = torch.randint(0, p, (batch, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    out = torch.empty((batch, 2 * n - 1), dtype=torch.int32, device="cuda")
    want = torch.empty_like(out)
    for i in range(6):
        a = ((base * (i + 3) + i) % p).to(torch.int32)       # queued on torch's stream, not waited for
        plan.multiply(a, other, out=out, batch=batch)        # queued behind it
        got = out.clone()                                    # torch reads the product behind the library's kernels
        torch.cuda.synchronize()
        ref_plan.multiply(a, other, out=want, batch=batch)
        ref_ctx.synchronize()
        assert torch.equal(got, want), i
    plan.close()
    ref_plan.close()
    c.close()
    ref_ctx.close()


@pytest.mark.parametrize("n", [40000, 70000, 131072])
def test_crt_batched_fused_tail(ctx, n):
    """ntt_crt_multiply_batched at L = 2^17 / 2^18 with batch >= 4: groups on two lanes, the last inverse column pass of
    the four primes and the merge in one kernel (kernels_crt.hip).  Every row against the single-product call (separate
    kernels), rows 0 and last against the CRT oracle (main-openmp.cc:693-711); extreme operands in row 0."""
    import torch
    p64, batch = 1337006139375617, 7
    rng = np.random.default_rng(n)
    a = rng.integers(0, p64, size=(batch, n), dtype=np.uint64)
    b = rng.integers(0, p64, size=(batch, n), dtype=np.uint64)
    a[0, :] = p64 - 1
    b[0, :] = p64 - 1
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.full((batch, 2 * n - 1), -1, dtype=torch.int64, device="cuda")
    ctx.crt_multiply_batched(da, db, p64, out=out)
    ctx.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for r in range(batch):
        assert np.array_equal(got[r], ctx.crt_multiply(a[r], b[r], p64)), r
    for r in (0, batch - 1):
        want, _ = orc.poly_multiply_crt(a[r], b[r], p64, threads=4)
        assert np.array_equal(got[r], want), r
    # a second, smaller batch through the cached workspace, host arrays this time
    got2 = ctx.crt_multiply_batched(a[2:6].copy(), b[2:6].copy(), p64)
    assert np.array_equal(got2, got[2:6])
