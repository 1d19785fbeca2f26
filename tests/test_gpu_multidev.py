"""Multi-device contexts (ntt_ctx_create with ndev > 1, ntt_amd/csrc/multidev.cpp): the in-library form of the
reference's distributed program (main-mpi.cc:642-819), rehearsed on ONE GPU with repeated device ids -- every slot is a
full single-device context with its own streams, plans and buffers, the exchanges are the same peer copies / peer
loads and stores a node of distinct GPUs would run (peer == self is legal for both).  Through the C ABI (ctypes),
bit-exact against the single-context result and the CPU oracle:

* batch sharding of products, negacyclic products and transforms;
* BASELINE.json configs[2]: the 4-prime CRT product at n = 2^17 with one prime per slot + the slice exchange;
* configs[3]: 32 RNS limbs x 256 polynomial pairs at n = 2^16 over 8 slots (limbs per slot, and rows per slot);
* configs[4]: one transform of 2^26 points split over 8 slots, every transport of the in-library exchange.

Never run by us on more than one physical GPU."""
import numpy as np
import pytest

import orc
from test_gpu_full_configs import _check_rows, _rns_primes

pytestmark = pytest.mark.gpu

P64 = 1337006139375617


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


def _rand32(rng, rows, n, p):
    x = rng.integers(0, p, size=(rows, n), dtype=np.int64).astype(np.uint32)
    x[0, :] = p - 1
    return x


def test_group_context_shape_and_errors(nt):
    g = nt.Context(devices=[0, 0, 0])
    assert g.device_count == 3 and nt.Context(0).device_count == 1
    s1 = g.slot(1)
    assert s1.device == 0 and s1.stream                                   # a full single-device context behind every slot
    with pytest.raises(nt.NttError) as e:
        g.set_stream(0)
    assert e.value.code == nt.ERR_UNSUPPORTED
    with pytest.raises(nt.NttError):
        g.slot(3)
    with pytest.raises(nt.NttError):
        g.plan64(P64, 100)
    with pytest.raises(nt.NttError):
        g.enable_kernel_timing(True)
    with pytest.raises(nt.NttError):
        nt.Context(devices=[0, 99])                                       # no such device: nothing leaks, error reported
    # per-slot work through the slot's own context
    p, n = 998244353, 300
    rng = np.random.default_rng(0)
    a, b = _rand32(rng, 1, n, p), _rand32(rng, 1, n, p)
    plan = s1.plan(p, n, 1)
    assert np.array_equal(plan.multiply(a, b)[0], orc.poly_multiply(a[0], b[0], p))
    plan.close()
    g.close()


@pytest.mark.parametrize("ndev,batch", [(2, 7), (3, 10), (4, 2), (8, 8)])
def test_group_batch_sharding(nt, ndev, batch):
    """Rows of the batch per slot by the reference's partition rule (main-mpi.cc:657-660): host arrays and device
    arrays, linear and negacyclic products, natural-order transforms; every row against the oracle."""
    import torch
    p, n = 998244353, 5000
    rng = np.random.default_rng(ndev * 100 + batch)
    a, b = _rand32(rng, batch, n, p), _rand32(rng, batch, n, p)
    g = nt.Context(devices=[0] * ndev)
    for kind in (nt.MONTGOMERY, nt.BARRETT):
        plan = g.plan(p, n, batch, kind)
        got = plan.multiply(a, b)                                          # host arrays: every slot stages its rows
        for r in range(batch):
            assert np.array_equal(got[r], orc.poly_multiply(a[r], b[r], p)), (kind, r)
        stride = 2 * n + 1
        da = torch.from_numpy(a.view(np.int32)).cuda()
        db = torch.from_numpy(b.view(np.int32)).cuda()
        out = torch.full((batch, stride), -5, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        plan.multiply(da, db, out=out, batch=batch, out_stride=stride)     # device arrays: queued on the slots' streams
        g.synchronize()
        assert np.array_equal(out[:, :2 * n - 1].cpu().numpy().view(np.uint32), got)
        assert bool((out[:, 2 * n - 1:] == -5).all())
        if batch > 1:                                                       # fewer rows than the plan was made for
            part = plan.multiply(a[:batch - 1].copy(), b[:batch - 1].copy())
            assert np.array_equal(part, got[:batch - 1])
        plan.close()
    # negacyclic products (n a power of two) and transforms, rows over the slots
    n2 = 4096 * 4
    a2, b2 = _rand32(rng, batch, n2, p), _rand32(rng, batch, n2, p)
    ring = g.negacyclic_plan(p, n2, batch)
    c = ring.multiply(a2, b2)
    for r in (0, batch - 1):
        lin = np.asarray(orc.poly_multiply(a2[r], b2[r], p)).astype(np.int64)
        hi = np.zeros(n2, dtype=np.int64)
        hi[:n2 - 1] = lin[n2:]
        assert np.array_equal(c[r], ((lin[:n2] - hi) % p).astype(np.uint32)), r
    ring.close()
    tplan = g.plan(p, 8192, batch)                                          # L = 16384
    x = rng.integers(0, p, size=(batch, 16384), dtype=np.int64).astype(np.uint32)
    y = x.copy()
    tplan.forward(y, batch)
    for r in (0, batch - 1):
        assert np.array_equal(y[r], orc.ntt(x[r], p)), r
    tplan.inverse(y, batch)
    assert np.array_equal(y, x)
    tplan.close()
    g.close()


@pytest.mark.parametrize("ndev,n", [(4, 131072), (2, 3001), (3, 5000), (8, 4097), (4, 5)])
def test_group_crt_prime_per_slot(nt, ndev, n):
    """BASELINE.json configs[2] (ndev = 4, n = 2^17): prime k on slot k / ceil(4 / ndev), one exchange of coefficient
    slices, the merge on every slot (main-mpi.cc:657-660, 739-797).  Ragged slices (2n - 1 not divisible by ndev), slots
    without a prime (ndev = 8) and slots with two primes (ndev = 2, 3); host and device arrays; against the oracle's
    CRT product (main-openmp.cc:693-711) and the single-context result."""
    import torch
    rng = np.random.default_rng(n + ndev)
    a = rng.integers(0, P64, size=n, dtype=np.uint64)
    b = rng.integers(0, P64, size=n, dtype=np.uint64)
    a[0] = b[0] = a[-1] = P64 - 1
    want, _ = orc.poly_multiply_crt(a, b, P64, threads=8)
    g = nt.Context(devices=[0] * ndev)
    got = g.crt_multiply(a, b, P64)                                         # host arrays
    assert np.array_equal(got, want)
    single = nt.Context(0)
    assert np.array_equal(single.crt_multiply(a, b, P64), got)
    single.close()
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    out = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):                                                      # the cached exchange buffers, twice
        out.zero_()
        torch.cuda.synchronize()
        g.crt_multiply(da, db, P64, out=out)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    # a modulus that is not NTT-friendly, and a batch (rows per slot, no exchange)
    p2 = (1 << 50) - 27                                                      # odd, not NTT-friendly; n * p2^2 below the CRT range
    a2, b2 = a % np.uint64(p2), b % np.uint64(p2)
    w2, _ = orc.poly_multiply_crt(a2, b2, p2, threads=8)
    assert np.array_equal(g.crt_multiply(a2, b2, p2), w2)
    if n <= 5000:
        B = ndev + 1
        A = rng.integers(0, P64, size=(B, n), dtype=np.uint64)
        Bm = rng.integers(0, P64, size=(B, n), dtype=np.uint64)
        gb = g.crt_multiply_batched(A, Bm, P64)
        for r in range(B):
            wr, _ = orc.poly_multiply_crt(A[r], Bm[r], P64, threads=4)
            assert np.array_equal(gb[r], wr), r
    g.close()


def test_group_rns_config3_limbs_and_rows(nt):
    """BASELINE.json configs[3] over 8 slots: n = 2^16, 32 limbs (distinct primes just above 2^30) x 256 polynomial
    pairs.  Limb sharding: slot s owns limbs partition(32, 8, s) and multiplies their 256 pairs on its own context;
    row sharding: the 256 pairs of a limb through ONE multi-device plan.  All 8192 rows by the evaluation identity,
    exact coefficient slices on two limbs."""
    import torch
    n, limbs, polys, ndev = 65536, 32, 256, 8
    primes = _rns_primes(limbs, 17, 1 << 30)
    g = nt.Context(devices=[0] * ndev)
    rng = np.random.default_rng(43)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(43)
    per = -(-limbs // ndev)
    ab = torch.empty((polys, 2 * n - 1), dtype=torch.int32, device="cuda")
    for li, p in enumerate(primes):
        a = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        b = torch.randint(0, p, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        a[0, :] = p - 1
        b[0, :] = p - 1
        torch.cuda.synchronize()
        if li % 4 == 3:                                                     # this limb: rows over the 8 slots
            plan = g.plan(p, n, polys, li % 2)
            plan.multiply(a, b, out=ab, batch=polys)
            g.synchronize()
        else:                                                               # limb on its owner slot (main-mpi.cc:657-660)
            sc = g.slot(li // per)
            plan = sc.plan(p, n, polys, li % 2)
            plan.multiply(a, b, out=ab, batch=polys)
            sc.synchronize()
        _check_rows(a, b, ab, 2 * n - 1, p, torch, rng)
        if li in (3, limbs - 2):
            for r in (1, polys - 1):
                ai, bi = a[r].cpu().tolist(), b[r].cpu().tolist()
                got = ab[r].cpu().tolist()
                for k in list(range(16)) + list(range(2 * n - 1 - 16, 2 * n - 1)):
                    lo, hi = max(0, k - (n - 1)), min(k, n - 1)
                    assert sum(ai[i] * bi[k - i] for i in range(lo, hi + 1)) % p == got[k], (p, r, k)
        plan.close()
    g.close()


def _dist_roundtrip(nt, monkeypatch, lgL, ndev, transport, kind, p=469762049, chunks=None, check_single=True):
    import torch
    L, S = 1 << lgL, (1 << lgL) // ndev
    monkeypatch.setenv("NTT_AMD_DIST_TRANSPORT", transport)
    if chunks is not None:
        monkeypatch.setenv("NTT_AMD_DIST_CHUNKS", str(chunks))
    g = nt.Context(devices=[0] * ndev)
    dp = nt.DistPlan(g, p, lgL, ndev, 0, kind)
    assert dp.transport[0] == transport and dp.local_length == S
    gen = torch.Generator(device="cuda")
    gen.manual_seed(lgL * 31 + ndev)
    x = torch.randint(0, p, (L,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    x[0] = p - 1
    x[L - 1] = p - 1
    xin = [x[r * S:(r + 1) * S].clone() for r in range(ndev)]
    keep = [t.clone() for t in xin]
    X = [torch.full((S,), -1, dtype=torch.int32, device="cuda") for _ in range(ndev)]
    back = [torch.full((S,), -1, dtype=torch.int32, device="cuda") for _ in range(ndev)]
    torch.cuda.synchronize()
    dp.forward(xin, X)
    dp.inverse(X, back)                                                     # queued behind the forward transform
    g.synchronize()
    for r in range(ndev):
        assert torch.equal(xin[r], keep[r]), r                              # the input blocks are not modified
        assert torch.equal(back[r], keep[r]), r                             # inverse(forward(x)) = x
    if check_single:                                                        # the whole transform on one context, natural order
        single = nt.Context(0)
        plan = single.plan(p, L // 2, 1, kind)
        assert plan.length == L
        y = x.clone()
        torch.cuda.synchronize()
        plan.forward(y, 1)
        single.synchronize()
        for r in range(ndev):
            assert torch.equal(X[r], y[r::ndev]), r                         # slot r holds X[r + ndev * k2]
        plan.close()
        single.close()
    first = [t.clone() for t in X]
    # a second pair of calls on the same plan (its staging arrays and events are reused), other data
    xin2 = [torch.randint(0, p, (S,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32) for _ in range(ndev)]
    torch.cuda.synchronize()
    dp.forward(xin2, X)
    dp.inverse(X, back)
    g.synchronize()
    for r in range(ndev):
        assert torch.equal(back[r], xin2[r]), r
    dp.close()
    g.close()
    return x, first


@pytest.mark.parametrize("transport", ["p2p", "copyin", "copy"])
@pytest.mark.parametrize("ndev", [2, 4, 8, 16])
def test_group_dist_transform(nt, monkeypatch, ndev, transport):
    """One transform of 2^20 points split over the slots with the exchanges inside the library, every transport:
    against the single-context transform (itself oracle-checked) and, at this size, the CPU oracle directly."""
    x, X = _dist_roundtrip(nt, monkeypatch, 20, ndev, transport, nt.MONTGOMERY if ndev != 4 else nt.BARRETT)
    want = orc.ntt(x.cpu().numpy().view(np.uint32), 469762049)
    for r in (0, ndev - 1):
        assert np.array_equal(X[r].cpu().numpy().view(np.uint32), want[r::ndev]), r


def test_group_dist_transform_strict_prime_and_chunks(nt, monkeypatch):
    """A prime above 2^30 (strict kernels; 2^22 | p - 1) and other chunk counts of the copy transports."""
    p = 1224736769
    assert (p - 1) % (1 << 22) == 0
    _dist_roundtrip(nt, monkeypatch, 20, 4, "copy", nt.MONTGOMERY, p=p, chunks=2)
    _dist_roundtrip(nt, monkeypatch, 18, 2, "copyin", nt.MONTGOMERY, p=p, chunks=8)
    _dist_roundtrip(nt, monkeypatch, 16, 4, "p2p", nt.BARRETT, p=p)


@pytest.mark.parametrize("transport", ["p2p", "copyin", "copy"])
def test_group_dist_transform_2pow26_over_8_slots(nt, monkeypatch, transport):
    """BASELINE.json configs[4]: L = 2^26, p = 469762049 over 8 slots, in-library exchange; against the single-context
    transform of the same array (natural order) and four points by direct evaluation on the CPU."""
    p, lgL, ndev = 469762049, 26, 8
    x, X = _dist_roundtrip(nt, monkeypatch, lgL, ndev, transport, nt.MONTGOMERY)
    L = 1 << lgL
    xs = x.cpu().numpy().view(np.uint32).astype(np.uint64)
    w = pow(3, (p - 1) // L, p)
    rng = np.random.default_rng(26)
    for k in [1, L - 1] + [int(v) for v in rng.integers(2, L - 1, size=2)]:
        wk = pow(w, k, p)
        # sum x[i] wk^i by blocks of 2^13 with vectorised powers (exact: products < 2^60)
        blk = 1 << 13
        pw = np.ones(blk, dtype=np.uint64)
        for i in range(1, blk):
            pw[i] = pw[i - 1] * np.uint64(wk) % np.uint64(p)
        step = pow(wk, blk, p)
        sums = (xs.reshape(-1, blk) * pw % np.uint64(p)).sum(axis=1) % np.uint64(p)          # < 2^13 * 2^29
        acc, cur = 0, 1
        for s in sums.tolist():
            acc = (acc + s * cur) % p
            cur = cur * step % p
        assert int(X[k % ndev][k // ndev].item()) == acc, k


def test_group_dist_rccl_transport_is_refused_on_repeated_devices(nt, monkeypatch):
    """NTT_AMD_DIST_TRANSPORT=rccl opens librccl (dlopen) and asks for one communicator per slot; RCCL needs distinct
    devices, so the one-GPU rehearsal must get a clean NTT_ERR_UNSUPPORTED carrying RCCL's message -- and the context
    must stay usable (the next plan takes the default transport).  The data path of this transport has never run."""
    monkeypatch.setenv("NTT_AMD_DIST_TRANSPORT", "rccl")
    g = nt.Context(devices=[0, 0])
    with pytest.raises(nt.NttError) as e:
        nt.DistPlan(g, 469762049, 18, 2, 0)
    assert e.value.code == nt.ERR_UNSUPPORTED and ("ncclCommInitAll" in str(e.value) or "librccl" in str(e.value)), str(e.value)
    monkeypatch.setenv("NTT_AMD_DIST_TRANSPORT", "bogus")
    with pytest.raises(nt.NttError) as e:
        nt.DistPlan(g, 469762049, 18, 2, 0)
    assert e.value.code == nt.ERR_INVALID
    monkeypatch.delenv("NTT_AMD_DIST_TRANSPORT")
    dp = nt.DistPlan(g, 469762049, 18, 2, 0)
    assert dp.transport[0] == "p2p"
    dp.close()
    g.close()


@pytest.mark.parametrize("lgL,ndev,transport,kind_name,p", [
    (16, 2, "p2p", "MONTGOMERY", 998244353), (16, 4, "copy", "BARRETT", 998244353), (18, 4, "copyin", "MONTGOMERY", 469762049),
    (20, 8, "p2p", "MONTGOMERY", 1224736769), (20, 16, "p2p", "BARRETT", 469762049), (14, 2, "p2p", "MONTGOMERY", 7340033),
    (14, 4, "p2p", "MONTGOMERY", 7340033), (12, 2, "copy", "BARRETT", 104857601)])
def test_group_dist_multiply(nt, monkeypatch, lgL, ndev, transport, kind_name, p):
    """ntt_dist_multiply: the product of two polynomials of L / 2 coefficients held as blocks on the slots -- forward
    split transform twice, pointwise product, inverse split transform, the spectra never reordered -- against the
    oracle's poly_multiply (main-barrett.cc:249-282 body; a lazy, a wide-range and a strict prime, both kernel families,
    a local length of 4096 and below included) and unchanged operands."""
    import torch
    monkeypatch.setenv("NTT_AMD_DIST_TRANSPORT", transport)
    kind = getattr(nt, kind_name)
    L, S, n = 1 << lgL, (1 << lgL) // ndev, (1 << lgL) // 2
    rng = np.random.default_rng(lgL * 7 + ndev)
    a = rng.integers(0, p, size=n, dtype=np.int64).astype(np.uint32)
    b = rng.integers(0, p, size=n, dtype=np.int64).astype(np.uint32)
    a[:100] = p - 1
    b[:100] = p - 1
    g = nt.Context(devices=[0] * ndev)
    dp = nt.DistPlan(g, p, lgL, ndev, 0, kind)
    half = ndev // 2
    ab_ = [torch.from_numpy(a[r * S:(r + 1) * S].view(np.int32).copy()).cuda() for r in range(half)]
    bb_ = [torch.from_numpy(b[r * S:(r + 1) * S].view(np.int32).copy()).cuda() for r in range(half)]
    out = [torch.full((S,), -1, dtype=torch.int32, device="cuda") for _ in range(ndev)]
    torch.cuda.synchronize()
    for _ in range(2):                                       # twice on one plan: spectra buffers and events reused
        dp.multiply(ab_, bb_, out)
    g.synchronize()
    got = torch.cat(out).cpu().numpy().view(np.uint32)
    want = orc.poly_multiply(a, b, p, kind="barrett")
    assert np.array_equal(got[:L - 1], want) and got[L - 1] == 0
    for r in range(half):
        assert np.array_equal(ab_[r].cpu().numpy().view(np.uint32), a[r * S:(r + 1) * S])
    dp.close()
    g.close()


def test_group_dist_multiply_2pow26_over_8_slots(nt):
    """The split product at L = 2^26 (two polynomials of 2^25 coefficients) over 8 slots against the single-context
    product of the same operands (itself oracle-checked at this size by test_largest_product...) and the evaluation
    identity A(x) B(x) = AB(x) at three points."""
    import torch
    from bench import _eval_point_t
    p, lgL, ndev = 469762049, 26, 8
    L, S, n = 1 << lgL, (1 << lgL) // ndev, (1 << lgL) // 2
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2626)
    a = torch.randint(0, p, (n,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, p, (n,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    g = nt.Context(devices=[0] * ndev)
    dp = nt.DistPlan(g, p, lgL, ndev, 0)
    ab_ = [a[r * S:(r + 1) * S].clone() for r in range(ndev // 2)]
    bb_ = [b[r * S:(r + 1) * S].clone() for r in range(ndev // 2)]
    out = [torch.full((S,), -1, dtype=torch.int32, device="cuda") for _ in range(ndev)]
    torch.cuda.synchronize()
    dp.multiply(ab_, bb_, out)
    g.synchronize()
    got = torch.cat(out)
    dp.close()
    g.close()
    assert int(got[L - 1].item()) == 0 and int(got.max().item()) < p and int(got.min().item()) >= 0
    for x in (1, p - 1, 0x2545F491 % p):
        ea, eb, eab = _eval_point_t(torch, a, x, p), _eval_point_t(torch, b, x, p), _eval_point_t(torch, got, x, p)
        assert ea * eb % p == eab, x
    single = nt.Context(0)
    plan = single.plan(p, n, 1)
    want = torch.full((2 * n - 1,), -1, dtype=torch.int32, device="cuda")
    plan.multiply(a.view(1, n), b.view(1, n), out=want.view(1, -1), batch=1)
    single.synchronize()
    assert torch.equal(got[:L - 1], want)
    plan.close()
    single.close()


@pytest.mark.parametrize("devices", ["0,0,0,0", "0,0"])
def test_cli_on_a_multi_device_context(nt, nttdata, tmp_path, devices):
    """ntt_cli --devices: the rank layout of main-mpi.cc:642-819 in one process -- cases with a 32-bit modulus on slot 0
    (the other slots idle, :710-738), the 51-bit case with its four CRT primes partitioned over the slots and the slice
    exchange (:739-797).  The reference's cases 0 and 1 against their golden outputs, a synthetic 51-bit case against
    the CRT oracle; verdict lines, latency lines and written files as the single-device harness."""
    import os
    import subprocess
    data, out = tmp_path / "nttdata", tmp_path / "files"
    data.mkdir()
    out.mkdir()
    for cid in (0, 1):
        n, p, a, b, ab = nttdata[cid]
        with open(data / f"{cid}.in", "w") as f:
            f.write(f"{n} {p}\n" + "".join(f"{v} " for v in a) + "\n" + "".join(f"{v} " for v in b))
        with open(data / f"{cid}.out", "w") as f:
            f.write("".join(f"{v} " for v in ab))
    n4 = 3001
    rng = np.random.default_rng(len(devices))
    a4 = rng.integers(0, P64, size=n4, dtype=np.uint64)
    b4 = rng.integers(0, P64, size=n4, dtype=np.uint64)
    ab4, _ = orc.poly_multiply_crt(a4, b4, P64, threads=4)
    with open(data / "2.in", "w") as f:
        f.write(f"{n4} {P64}\n" + "".join(f"{v} " for v in a4) + "\n" + "".join(f"{v} " for v in b4))
    with open(data / "2.out", "w") as f:
        f.write("".join(f"{v} " for v in ab4))
    exe = os.path.join(os.path.dirname(nt.__file__), "ntt_cli")
    r = subprocess.run([exe, "--data", str(data), "--out", str(out), "--begin", "0", "--end", "2", "--epochs", "3",
                        "--devices", devices], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("多项式乘法结果正确") == 3 and "多项式乘法结果错误" not in r.stdout
    assert any(l.startswith(f"average latency for n = {n4} p = {P64} : ") for l in r.stdout.split("\n"))
    assert np.array_equal(np.loadtxt(out / "2.out", dtype=np.uint64), ab4)
    assert np.array_equal(np.loadtxt(out / "1.out", dtype=np.uint64).astype(np.uint32), nttdata[1][4])
    r = subprocess.run([exe, "--devices", "0,x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2


def test_group_multiply_with_arrays_per_slot(nt):
    """ntt_poly_multiply_batched_slots: every slot multiplies arrays of its own (the batch-per-GPU layout for one
    process), uneven counts and a slot without work; rows against the oracle, capacity and null checks."""
    import torch
    p, n, ndev = 998244353, 40000, 3
    rng = np.random.default_rng(17)
    g = nt.Context(devices=[0] * ndev)
    plan = g.plan(p, n, 12)                                   # 4 products per slot
    counts = [4, 0, 3]
    host, dev_a, dev_b, outs = [], [], [], []
    for c in counts:
        a = rng.integers(0, p, size=(max(c, 1), n), dtype=np.int64).astype(np.uint32)
        b = rng.integers(0, p, size=(max(c, 1), n), dtype=np.int64).astype(np.uint32)
        host.append((a, b))
        dev_a.append(torch.from_numpy(a.view(np.int32)).cuda() if c else None)
        dev_b.append(torch.from_numpy(b.view(np.int32)).cuda() if c else None)
        outs.append(torch.full((c, 2 * n - 1), -1, dtype=torch.int32, device="cuda") if c else None)
    torch.cuda.synchronize()
    plan.multiply_slots(dev_a, dev_b, outs, counts)
    g.synchronize()
    for s, c in enumerate(counts):
        for r in range(c):
            assert np.array_equal(outs[s][r].cpu().numpy().view(np.uint32), orc.poly_multiply(host[s][0][r], host[s][1][r], p)), (s, r)
    # host arrays per slot work as well (each slot stages its own)
    houts = [np.zeros((c, 2 * n - 1), dtype=np.uint32) if c else None for c in counts]
    plan.multiply_slots([h[0] if c else None for h, c in zip(host, counts)], [h[1] if c else None for h, c in zip(host, counts)], houts, counts)
    for s, c in enumerate(counts):
        if c:
            assert np.array_equal(houts[s], outs[s].cpu().numpy().view(np.uint32))
    small = g.plan(p, n, 9)                                   # 3 products per slot
    with pytest.raises(nt.NttError) as e:
        small.multiply_slots(dev_a, dev_b, outs, counts)      # 4 on slot 0: more than ceil(9 / 3)
    assert e.value.code == nt.ERR_INVALID
    small.close()
    single = nt.Context(0)
    sp = single.plan(p, n, 4)
    with pytest.raises(nt.NttError):
        sp.multiply_slots([dev_a[0]], [dev_b[0]], [outs[0]], [4])
    sp.close()
    single.close()
    plan.close()
    g.close()
