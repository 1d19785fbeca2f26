"""BASELINE.json configs[1] and configs[3] at their STATED sizes on one MI355X (the parity tests of
test_gpu_parity.py run the same shapes at small batches):

* configs[1]: 1024 products of n = 2^17, p = 998244353 in one call -- the 22 launch groups on two stream lanes that
  bench.py times -- every row checked by evaluation identities, rows on both sides of every kind of group / lane
  boundary against the oracle;
* configs[3]: n = 2^16, 32 RNS limbs (distinct primes just above 2^30) x 256 polynomial pairs per limb.

The per-row checks are size-independent properties evaluated on the GPU with torch int64 arithmetic (plumbing): a
product is right at a point x iff A(x) * B(x) = AB(x) (mod p); a wrong coefficient survives three random points with
probability ~ (2n / p)^3."""
import os
import sys

import numpy as np
import pytest

import orc

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def _powers(x, count, p, torch):
    """[x^0, x^1, ..., x^(count-1)] mod p as an int64 device tensor (doubling: log2(count) vector multiplies)."""
    pw = torch.ones(1, dtype=torch.int64, device="cuda")
    xk = x % p
    while pw.numel() < count:
        pw = torch.cat([pw, pw * xk % p])
        xk = xk * xk % p
    return pw[:count]


def _eval_rows(t, pw, p, torch, rows_per_step=64):
    """Horner-free evaluation of every row of the int32 tensor t (residues < 2^31) at the point behind `pw`."""
    out = []
    for r0 in range(0, t.shape[0], rows_per_step):
        blk = t[r0:r0 + rows_per_step].to(torch.int64)
        out.append(((blk * pw[:blk.shape[1]]) % p).sum(dim=1) % p)       # terms < 2^62, sums < 2^49
    return torch.cat(out)


def _check_rows(a, b, ab, out_len, p, torch, rng):
    assert int(ab[:, :out_len].max()) < p and int(ab[:, :out_len].min()) >= 0          # canonical residues
    for x in (1, p - 1, int(rng.integers(2, p - 1)), int(rng.integers(2, p - 1))):
        pw = _powers(x, out_len, p, torch)
        ea, eb, eab = _eval_rows(a, pw, p, torch), _eval_rows(b, pw, p, torch), _eval_rows(ab[:, :out_len], pw, p, torch)
        bad = torch.nonzero(ea * eb % p != eab).flatten().tolist()
        assert not bad, (x, bad[:8])


def test_config1_batch_1024():
    import torch
    import ntt_amd
    import bench

    n, B, p = 131072, 1024, 998244353
    stride = 2 * n
    ha = bench.splitmix_fill(0x20250711, B, n, p)                 # the benchmark's own inputs
    hb = bench.splitmix_fill(0x20250711 + (1 << 20), B, n, p)
    a = torch.from_numpy(ha.view(np.int32)).cuda()
    b = torch.from_numpy(hb.view(np.int32)).cuda()
    ab = torch.full((B, stride), -1, dtype=torch.int32, device="cuda")
    ctx = ntt_amd.Context(0)
    plan = ctx.plan(p, n, B)
    plan.multiply(a, b, out=ab, batch=B, out_stride=stride)
    ctx.synchronize()
    assert int((ab[:, 2 * n - 1:] != -1).sum()) == 0              # entries >= 2n-1 of a row stay untouched
    _check_rows(a, b, ab, 2 * n - 1, p, torch, np.random.default_rng(1))
    # launch groups are 56 products (18 full ones and one of 16, alternating between the two stream lanes; 48 until the
    # end of round 3): rows on both sides of the first lane-0 | lane-1 and lane-1 | lane-0 boundaries of either
    # grouping and the last product, against the oracle
    rows = (0, 47, 48, 55, 56, 111, 112, B - 1)
    for r in rows:
        got = ab[r, :2 * n - 1].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, orc.poly_multiply(ha[r], hb[r], p)), r
    plan.close()
    ctx.close()


def _rns_primes(count, two_adicity, above):
    """The `count` smallest primes p > above with p = 1 mod 2^two_adicity."""
    out, step = [], 1 << two_adicity
    p = (above // step + 1) * step + 1
    while len(out) < count:
        if all(p % q for q in range(3, int(p ** 0.5) + 1, 2)):
            out.append(p)
        p += step
    return out


def test_config4_rns_32_limbs_x_256():
    import torch
    import ntt_amd

    n, limbs, polys = 65536, 32, 256
    primes = _rns_primes(limbs, 17, 1 << 30)
    ctx = ntt_amd.Context(0)
    rng = np.random.default_rng(4)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2025)
    ab = torch.empty((polys, 2 * n - 1), dtype=torch.int32, device="cuda")
    for li, p in enumerate(primes):
        a = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        b = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        a[0, :] = p - 1                                            # the largest residues in one pair of every limb
        b[0, :] = p - 1
        plan = ctx.plan(p, n, polys, li % 2)                      # alternate Montgomery / Barrett kernels
        plan.multiply(a, b, out=ab, batch=polys)
        ctx.synchronize()
        _check_rows(a, b, ab, 2 * n - 1, p, torch, rng)
        if li in (0, limbs - 1):                                  # exact coefficients at both ends of two rows
            for r in (1, polys - 1):
                ai, bi = a[r].cpu().tolist(), b[r].cpu().tolist()
                got = ab[r].cpu().tolist()
                for k in list(range(24)) + list(range(2 * n - 1 - 24, 2 * n - 1)):
                    lo, hi = max(0, k - (n - 1)), min(k, n - 1)
                    assert sum(ai[i] * bi[k - i] for i in range(lo, hi + 1)) % p == got[k], (p, r, k)
        plan.close()
    # the same batch as negacyclic products (the ring FHE limbs multiply in): c = ab_lo - ab_hi, limb 0
    p = primes[0]
    a = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    lin = ctx.plan(p, n, polys)
    lin.multiply(a, b, out=ab, batch=polys)
    ring = ctx.negacyclic_plan(p, n, polys)
    c = torch.empty((polys, n), dtype=torch.int32, device="cuda")
    ring.multiply(a, b, out=c, batch=polys)
    ctx.synchronize()
    lo = ab[:, :n].to(torch.int64)
    hi = torch.zeros_like(lo)
    hi[:, :n - 1] = ab[:, n:].to(torch.int64)
    assert torch.equal((lo - hi) % p, c.to(torch.int64))
    lin.close()
    ring.close()
    ctx.close()


@pytest.mark.parametrize("shape", ["linear-8192", "linear-5000", "negacyclic-8192"])
def test_one_kernel_products_at_full_batch(shape):
    """The one-kernel products (seq_product_kernel, kernels_fused.hip: L = 16384 and the full-length L = 8192) at the
    batch the size sweep times them with (4096 device-resident products, one workgroup each): every row by the
    evaluation identity -- linear: A(x) B(x) = AB(x); negacyclic: at the roots of x^n + 1, the odd powers of a 2n-th
    root --, three rows against the oracle."""
    import torch
    import ntt_amd

    kind, n = shape.split("-")
    n, B, p = int(n), 4096, 998244353
    neg = kind == "negacyclic"
    rng = np.random.default_rng(n + neg)
    ha = rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.uint32)
    hb = rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.uint32)
    ha[0, :] = p - 1
    hb[0, :] = p - 1
    a = torch.from_numpy(ha.view(np.int32)).cuda()
    b = torch.from_numpy(hb.view(np.int32)).cuda()
    out_len = n if neg else 2 * n - 1
    stride = out_len + 1
    ab = torch.full((B, stride), -1, dtype=torch.int32, device="cuda")
    ctx = ntt_amd.Context(0)
    plan = ctx.negacyclic_plan(p, n, B) if neg else ctx.plan(p, n, B)
    plan.multiply(a, b, out=ab, batch=B, out_stride=stride)
    ctx.synchronize()
    plan.close()
    ctx.close()
    assert bool((ab[:, out_len:] == -1).all())                    # beyond the product: not ours
    if neg:
        # x^n = -1 at x = psi^(2k+1): there the negacyclic product evaluates like the plain one
        g = 3
        psi = pow(g, (p - 1) // (2 * n), p)
        assert int(ab[:, :out_len].max()) < p and int(ab[:, :out_len].min()) >= 0
        for k in (0, 1, int(rng.integers(2, n))):
            pw = _powers(pow(psi, 2 * k + 1, p), n, p, torch)
            ea, eb, eab = _eval_rows(a, pw, p, torch), _eval_rows(b, pw, p, torch), _eval_rows(ab[:, :n], pw, p, torch)
            bad = torch.nonzero(ea * eb % p != eab).flatten().tolist()
            assert not bad, (k, bad[:8])
    else:
        _check_rows(a, b, ab, out_len, p, torch, rng)
    got = ab.cpu().numpy().view(np.uint32)
    for r in (0, B // 2 + 1, B - 1):
        want = np.asarray(orc.poly_multiply(ha[r], hb[r], p)).astype(np.int64)
        if neg:
            hi = np.zeros(n, dtype=np.int64)
            hi[:n - 1] = want[n:]
            want = (want[:n] - hi) % p
        assert np.array_equal(got[r, :out_len], want.astype(np.uint32)), r
