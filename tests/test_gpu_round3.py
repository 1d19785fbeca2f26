"""Round-3 route and test gaps: branches of the CRT batch that no test reached, the
public merge on non-canonical residues, and the per-device ordering of the no-alloc entry point's scratch array.
All through the C ABI (ctypes), against the CPU oracle (main-openmp.cc:693-711 restated in oracle/ntt_oracle.c)."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

P64 = 1337006139375617                     # the reference's fifth modulus (main-openmp.cc:636)


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.fixture()
def ctx(nt):
    c = nt.Context(0)
    yield c
    c.close()


def _pairs(seed, batch, n):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, P64, size=(batch, n), dtype=np.uint64)
    b = rng.integers(0, P64, size=(batch, n), dtype=np.uint64)
    a[0, :] = P64 - 1
    b[0, :] = P64 - 1
    return a, b


def _check_rows(ctx, a, b, got, oracle_rows):
    for r in range(a.shape[0]):
        assert np.array_equal(got[r], ctx.crt_multiply(a[r], b[r], P64)), r      # the separate-kernel single product
    for r in oracle_rows:
        want, _ = orc.poly_multiply_crt(a[r], b[r], P64, threads=8)
        assert np.array_equal(got[r], want), r


@pytest.mark.parametrize("n,batch", [(10000, 6), (300000, 5)])
def test_crt_batch_unfused_grouped_branch(ctx, n, batch):
    """L = 2^15 (column level of 8 points) and L = 2^20 (a 256-point level): the grouped two-lane CRT batch without the
    fused head and tail -- reduce_all with the group offset, the two-slot work area, launch_crt_merge on a group slice
    (crt.cpp) -- every row against the single-product call, first and last row against the oracle."""
    a, b = _pairs(n, batch, n)
    got = ctx.crt_multiply_batched(a, b, P64)
    _check_rows(ctx, a, b, got, (0, batch - 1))


def test_crt_batch_fused_tail_without_fused_head(ctx):
    """A workspace created with batch == 4 keeps the LDS-staged forward column pass (ntt_plan::col16_fwd), so the
    batch runs with the fused tail but the separate reduce sweep (fuse && !fuse_in in crt.cpp)."""
    n = 70000
    a, b = _pairs(4, 4, n)
    got = ctx.crt_multiply_batched(a, b, P64)
    _check_rows(ctx, a, b, got, (0, 3))


def test_crt_batch_routes_ab_in_one_process(nt, monkeypatch):
    """NTT_AMD_CRT_FUSE / NTT_AMD_CRT_GROUP are read per call: the same batch through the fused and the unfused route
    and with groups of 2 gives identical arrays (and the oracle's rows)."""
    n, batch = 70000, 7
    a, b = _pairs(7, batch, n)
    c = nt.Context(0)
    fused = c.crt_multiply_batched(a, b, P64)
    monkeypatch.setenv("NTT_AMD_CRT_FUSE", "0")
    unfused = c.crt_multiply_batched(a, b, P64)
    monkeypatch.setenv("NTT_AMD_CRT_GROUP", "2")
    small_groups = c.crt_multiply_batched(a, b, P64)
    monkeypatch.delenv("NTT_AMD_CRT_FUSE")
    small_groups_fused = c.crt_multiply_batched(a, b, P64)
    c.close()
    assert np.array_equal(fused, unfused) and np.array_equal(fused, small_groups) and np.array_equal(fused, small_groups_fused)
    for r in (0, batch - 1):
        want, _ = orc.poly_multiply_crt(a[r], b[r], P64, threads=8)
        assert np.array_equal(fused[r], want), r


def test_crt_merge_takes_any_u32_residue(nt, ctx):
    """ntt_crt_merge is public: residues that are not canonical (r_k + m_k, where that still fits 32 bits, and the
    all-ones word) merge to the value of their canonical form (CRT_combine_parallel reduces with %, main-openmp.cc:476)."""
    rng = np.random.default_rng(5)
    count = 1000
    mods = [nt.crt_prime(k) for k in range(4)]
    canon = [rng.integers(0, m, size=count, dtype=np.int64).astype(np.uint32) for m in mods]
    want = ctx.crt_merge(canon, P64)
    loose = []
    for r, m in zip(canon, mods):
        x = r.astype(np.uint64) + np.uint64(m) * rng.integers(0, 4, size=count).astype(np.uint64)
        loose.append(np.where(x < (1 << 32), x, r.astype(np.uint64)).astype(np.uint32))
    assert any(int((l != c).sum()) > 0 for l, c in zip(loose, canon))
    assert np.array_equal(ctx.crt_merge(loose, P64), want)
    # against big integers on a few coefficients
    M = 1
    for m in mods:
        M *= m
    for i in range(0, count, 97):
        x = 0
        for r, m in zip(canon, mods):
            Mi = M // m
            x += int(r[i]) * Mi * pow(Mi, -1, m)
        assert int(want[i]) == x % M % P64


def test_noalloc_calls_of_two_contexts_are_ordered(nt):
    """Two contexts on one GPU share the device's scratch array (engine.cpp): calls issued back to back on their two
    streams, without any synchronisation between them, must each give the batched entry point's result."""
    import torch
    p, n, batch = 998244353, 131072, 24
    rng = np.random.default_rng(3)
    c0, c1 = nt.Context(0), nt.Context(0)
    plans = [c.plan(p, n, batch) for c in (c0, c1)]
    ops, want, got = [], [], []
    for i in range(2):
        a = torch.from_numpy(rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
        b = torch.from_numpy(rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32).view(np.int32)).cuda()
        ops.append((a, b))
        w = torch.zeros((batch, 2 * n - 1), dtype=torch.int32, device="cuda")
        plans[i].multiply(a, b, out=w, batch=batch)
        want.append(w)
        got.append(torch.zeros((batch, 2 * n - 1), dtype=torch.int32, device="cuda"))
    c0.synchronize()
    c1.synchronize()
    torch.cuda.synchronize()
    for rep in range(3):                                   # six calls in flight, alternating contexts
        for i in range(2):
            plans[i].multiply_noalloc(ops[i][0], ops[i][1], got[i], batch)
    c0.synchronize()
    c1.synchronize()
    for i in range(2):
        assert torch.equal(got[i], want[i]), i
    row = got[1][batch - 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(row, orc.poly_multiply(ops[1][0][batch - 1].cpu().numpy().view(np.uint32),
                                                 ops[1][1][batch - 1].cpu().numpy().view(np.uint32), p))
    for pl in plans:
        pl.close()
    c0.close()
    c1.close()


def test_crt_multiply_routes_to_direct_plan(nt, monkeypatch):
    """Default route of ntt_crt_multiply[_batched] (crt.cpp): p64 = 1337006139375617 is a prime with 2^46 | p64 - 1,
    so the product runs as ONE transform with 64-bit lanes; a modulus that is not such a prime keeps the four-prime
    path.  Same arrays either way, and the oracle's (main-openmp.cc:693-711)."""
    import torch
    rng = np.random.default_rng(64)
    c = nt.Context(0)
    for n, batch in ((1000, 3), (131072, 5)):
        a, b = _pairs(n + 1, batch, n)
        monkeypatch.setenv("NTT_AMD_CRT_DIRECT", "0")
        four = c.crt_multiply_batched(a, b, P64)
        monkeypatch.delenv("NTT_AMD_CRT_DIRECT")
        c.enable_kernel_timing(True)
        direct = c.crt_multiply_batched(a, b, P64)
        times = c.kernel_times()
        c.enable_kernel_timing(False)
        assert times["crt_merge"][1] == 0 and times["crt_reduce"][1] == 0        # no CRT kernel ran ...
        assert times["col64"][1] + times["lds64"][1] > 0                           # ... the 64-bit lanes did
        assert np.array_equal(direct, four)
        want, _ = orc.poly_multiply_crt(a[0], b[0], P64, threads=8)
        assert np.array_equal(direct[0], want)
        assert np.array_equal(c.crt_multiply(a[1], b[1], P64), four[1])           # batch 1 through the cached plan
        da = torch.from_numpy(a.view(np.int64)).cuda()
        db = torch.from_numpy(b.view(np.int64)).cuda()
        out = torch.zeros((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        c.crt_multiply_batched(da, db, P64, out=out)                               # device arrays, asynchronous
        c.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), four)
    # 2^51 - 129 is prime but 2^10 does not divide 2^51 - 130; 263882790666241 = 15 * 2^44 + 1 is prime: the first stays on four primes
    for p64, routed in (((1 << 51) - 129, False), (263882790666241, True)):
        n = 300
        a = rng.integers(0, p64, size=n, dtype=np.uint64)
        b = rng.integers(0, p64, size=n, dtype=np.uint64)
        c.enable_kernel_timing(True)
        got = c.crt_multiply(a, b, p64)
        times = c.kernel_times()
        c.enable_kernel_timing(False)
        assert (times["crt_merge"][1] == 0) == routed, p64
        want = [0] * (2 * n - 1)
        for i in range(n):
            ai = int(a[i])
            for j in range(n):
                want[i + j] += ai * int(b[j])
        assert [int(x) for x in got] == [w % p64 for w in want], p64
    c.close()


@pytest.mark.parametrize("n", [40000, 131072])
def test_forward_column_pass_prefetch_variant(nt, monkeypatch, n):
    """The walking first forward pass with the next product's loads in flight (col_walk_prefetch_kernel, the default for
    batches above 16 at column levels of 32 / 64 points) against the plain walking kernel (NTT_AMD_COL_PREFETCH=0) and
    the oracle: Montgomery and Barrett, a lazy and a strict prime, a batch that leaves workgroups with 2 and 3 products."""
    import torch
    batch = 41
    for p, kind in ((998244353, nt.MONTGOMERY), (998244353, nt.BARRETT), (1224736769, nt.MONTGOMERY), (469762049, nt.BARRETT)):
        rng = np.random.default_rng(n + p % 1000 + kind)
        a = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        b = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        a[0, :] = p - 1
        b[0, :] = p - 1
        da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
        outs = []
        c = nt.Context(0)
        plan = c.plan(p, n, batch, kind)
        for flag in ("1", "0"):
            monkeypatch.setenv("NTT_AMD_COL_PREFETCH", flag)
            out = torch.full((batch, 2 * n - 1), -1, dtype=torch.int32, device="cuda")
            plan.multiply(da, db, out=out, batch=batch)
            c.synchronize()
            outs.append(out)
        plan.close()
        c.close()
        assert torch.equal(outs[0], outs[1]), (p, kind)
        got = outs[0].cpu().numpy().view(np.uint32)
        for r in (0, 16, 17, 33, batch - 1):                # first, second and third product of a workgroup's walk
            assert np.array_equal(got[r], orc.poly_multiply(a[r], b[r], p, kind="barrett")), (p, kind, r)


def _largest_prime_below(limit, two_adicity):
    """The largest prime p < limit with 2^two_adicity | p - 1 (trial division: limit is 2^29 here)."""
    step = 1 << two_adicity
    p = (limit - 2) // step * step + 1
    while p > step:
        if all(p % q for q in range(3, int(p ** 0.5) + 1, 2)):
            return p
        p -= step
    raise AssertionError("no such prime")


@pytest.mark.parametrize("n,batch", [(2048, 3), (5000, 2), (20000, 3), (40000, 20), (131072, 1), (131072, 19), (1 << 21, 1)])
def test_wide_range_kernels_for_primes_below_2pow29(nt, monkeypatch, n, batch):
    """p < 2^29 (8p <= 2^32): the row kernels and forward register column passes keep sums unreduced up to 4p and hand
    out values below 8p (device_ntt.hpp, WIDE; NTT_AMD_WIDE=0 keeps the [0, 2p) kernels).  The reference's three shipped
    primes (main-montgomery.cc:301-306) and the largest prime below 2^29 with 2^18 | p - 1 (the tightest ranges), operands
    of all p - 1 in the first row, Montgomery and Barrett: identical to the [0, 2p) kernels, rows against the oracle
    (generator 3 is the reference's; for the fourth prime the evaluation identity stands in).  Column levels of 16, 32
    and 64 points and, at n = 2^21, two levels of 32."""
    import torch
    from test_gpu_full_configs import _check_rows
    tight = _largest_prime_below(1 << 29, 18)
    assert tight > (1 << 29) - (1 << 24)
    for p in (7340033, 104857601, 469762049, tight):
        if (p - 1) % (1 << (2 * n - 1).bit_length()) != 0:
            continue                                         # 7340033 = 7 * 2^20 + 1 serves L <= 2^20: every size here
        for kind in (nt.MONTGOMERY, nt.BARRETT):
            rng = np.random.default_rng(n + batch + p % 997 + kind)
            a = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
            b = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
            a[0, :] = p - 1
            b[0, :] = p - 1
            da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
            outs = []
            for flag in ("1", "0"):
                monkeypatch.setenv("NTT_AMD_WIDE", flag)        # read when the plan is created
                c = nt.Context(0)
                plan = c.plan(p, n, batch, kind)
                out = torch.full((batch, 2 * n - 1), -1, dtype=torch.int32, device="cuda")
                plan.multiply(da, db, out=out, batch=batch)
                c.synchronize()
                outs.append(out)
                if flag == "1" and n == 5000:                   # the natural-order transforms use the same row kernels
                    x = rng.integers(0, p, size=plan.length, dtype=np.int64).astype(np.uint32)
                    y = x.copy()
                    plan.forward(y, 1)
                    if p != tight:
                        assert np.array_equal(y, orc.ntt(x, p)), p
                    plan.inverse(y, 1)
                    assert np.array_equal(y, x), p
                plan.close()
                c.close()
            assert torch.equal(outs[0], outs[1]), (p, kind)
            _check_rows(da, db, outs[0], 2 * n - 1, p, torch, rng)
            if p != tight:
                got = outs[0].cpu().numpy().view(np.uint32)
                for r in sorted({0, batch - 1}):
                    assert np.array_equal(got[r], orc.poly_multiply(a[r], b[r], p, kind="barrett")), (p, kind, r)


def test_pinned_host_arrays_are_used_in_place(nt, ctx):
    """Small calls on page-locked host arrays (ntt_host_alloc) run zero-copy: the first pass reads the operands and the last
    pass writes the product over PCIe (engine.cpp, kZeroCopyBytes); larger calls and pageable arrays are staged.  Same
    results every way, the oracle's; entries beyond 2n - 1 of a pinned output row stay untouched."""
    p = 998244353
    rng = np.random.default_rng(91)
    for n, batch in ((131072, 1), (131072, 2), (1000, 3), (4096, 1), (2048, 2), (131072, 5)):
        a, b = nt.host_array((batch, n)), nt.host_array((batch, n))
        a[:] = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        b[:] = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        a[0, :] = p - 1
        stride = 2 * n + 2
        out = nt.host_array((batch, stride))
        out[:] = 0xDEADBEEF
        plan = ctx.plan(p, n, batch)
        plan.multiply(a, b, out=out, batch=batch, out_stride=stride)              # pinned in, pinned out
        paged = plan.multiply(np.array(a), np.array(b))                           # pageable copies of the same operands
        mixed = plan.multiply(a, np.array(b))                                     # one of each
        plan.close()
        assert np.array_equal(out[:, :2 * n - 1], paged) and np.array_equal(mixed, paged), (n, batch)
        assert bool((out[:, 2 * n - 1:] == 0xDEADBEEF).all())
        for r in (0, batch - 1):
            assert np.array_equal(paged[r], orc.poly_multiply(a[r], b[r], p)), (n, r)
    # the reference's entry point on pinned int arrays
    n = 131072
    ia, ib = nt.host_array(n, np.int32), nt.host_array(n, np.int32)
    ia[:] = rng.integers(0, p, size=n)
    ib[:] = rng.integers(0, p, size=n)
    iab = nt.host_array(2 * n - 1, np.int32)
    nt._lib.poly_multiply(ia.ctypes.data, ib.ctypes.data, iab.ctypes.data, n, p)
    assert np.array_equal(iab.view(np.uint32), orc.poly_multiply(ia.view(np.uint32), ib.view(np.uint32), p))


def test_object_lifetimes_return_their_device_memory(nt):
    """Every kind of object the library hands out, created, used and closed in a loop: the device's free memory comes back
    to where it was after the first (warm-up) cycle -- plans with their work buffers and tables, the CRT workspaces of a
    context and of a multi-device context, streamed-product rings, 64-bit plans, split-transform plans with their
    exchange buffers and streams, a communicator with its staging."""
    import torch
    p, p64 = 998244353, 1337006139375617
    rng = np.random.default_rng(5)
    a = rng.integers(0, p, size=(6, 20000), dtype=np.int64).astype(np.uint32)
    b = rng.integers(0, p, size=(6, 20000), dtype=np.int64).astype(np.uint32)
    a64 = rng.integers(0, p64, size=(3, 9000), dtype=np.uint64)
    b64 = rng.integers(0, p64, size=(3, 9000), dtype=np.uint64)
    first = {}

    def cycle(tag):
        ctx = nt.Context(0)
        pl = ctx.plan(p, 20000, 6)
        r1 = pl.multiply(a, b)
        r2 = pl.multiply_stream(a, b)
        ng = ctx.negacyclic_plan(p, 16384, 2).multiply(a[:2, :16384].copy(), b[:2, :16384].copy())
        c1 = ctx.crt_multiply_batched(a64, b64, p64)
        p6 = ctx.plan64(p64, 9000, 3)
        c2 = p6.multiply(a64, b64)
        dp = nt.DistPlan(ctx, 469762049, 16, 4, 1)
        d = torch.zeros(dp.local_length, dtype=torch.int32, device="cuda")
        dp.colpass(d)
        comm = nt.Comm.host(ctx, lambda s, r: r.__setitem__(Ellipsis, s), 1, 0)
        c3 = comm.crt_multiply(a64[0], b64[0], p64)
        ctx.synchronize()
        del d
        ctx.close()                                                          # closes pl, p6, dp, comm with it
        grp = nt.Context(devices=[0, 0])
        gp = grp.plan(p, 20000, 6)
        r3 = gp.multiply(a, b)
        c4 = grp.crt_multiply(a64[0], b64[0], p64)
        gd = nt.DistPlan(grp, 469762049, 16, 2, 0)
        xs = [torch.zeros(gd.local_length, dtype=torch.int32, device="cuda") for _ in range(4)]
        gd.forward(xs[:2], xs[2:])
        grp.synchronize()
        del xs
        grp.close()
        got = (r1, r2, ng, c1, c2, c3, r3, c4)
        if not first:
            first[0] = got
        else:
            assert all(np.array_equal(x, y) for x, y in zip(got, first[0])), tag
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    assert np.array_equal(cycle("warm-up") and first[0][0][2], orc.poly_multiply(a[2], b[2], p))
    base = cycle("base")
    for i in range(6):
        free = cycle(i)
    assert free >= base - (4 << 20), f"{(base - free) >> 20} MiB of device memory did not come back over 6 cycles"


@pytest.mark.parametrize("n", [1000, 5000, 20000, 131072])
def test_multiply_limbs_matches_per_limb_calls(nt, n):
    """ntt_poly_multiply_limbs: several moduli in one call, each limb whole on a stream lane (lazy, wide-range and strict
    primes side by side; L = 2048 -> the 4096-point row kernel, 16384 fused, 65536 and 2^18 through work buffers): every
    row equals the per-limb ntt_poly_multiply_batched result, rows of the small sizes the oracle's."""
    import torch
    primes = [998244353, 469762049, 1224736769, 7340033, 104857601]
    batch = 7 if n < 100000 else 3
    rng = np.random.default_rng(n)
    ctx = nt.Context(0)
    plans = [ctx.plan(p, n, batch) for p in primes]
    ha = [rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32) for p in primes]
    hb = [rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32) for p in primes]
    for x, p in zip(ha, primes):
        x[0, :] = p - 1
    for x, p in zip(hb, primes):
        x[0, :] = p - 1
    da = [torch.from_numpy(x.view(np.int32)).cuda() for x in ha]
    db = [torch.from_numpy(x.view(np.int32)).cuda() for x in hb]
    stride = 2 * n + 3
    outs = [torch.full((batch, stride), -1, dtype=torch.int32, device="cuda") for _ in primes]
    torch.cuda.synchronize()
    for _ in range(2):
        nt.multiply_limbs(plans, da, db, outs, batch, out_stride=stride)
    ctx.synchronize()
    for i, p in enumerate(primes):
        got = outs[i].cpu().numpy().view(np.uint32)
        assert np.all(got[:, 2 * n - 1:] == 0xFFFFFFFF), "wrote past the row"
        want = plans[i].multiply(ha[i], hb[i], batch=batch)
        assert np.array_equal(got[:, :2 * n - 1], want), (p, n)
        if n <= 20000:
            for r in (0, batch - 1):
                assert np.array_equal(want[r], orc.poly_multiply(ha[i][r], hb[i][r], p)), (p, n, r)
    # argument checks
    with pytest.raises(nt.NttError) as e:
        nt.multiply_limbs([plans[0], plans[0]], da[:2], db[:2], outs[:2], batch, out_stride=stride)      # a plan twice
    assert e.value.code == nt.ERR_INVALID
    with pytest.raises(nt.NttError):
        nt.multiply_limbs(plans[:2], ha[:2], db[:2], outs[:2], batch, out_stride=stride)                 # host array
    other = nt.Context(0)
    foreign = other.plan(primes[1], n, batch)
    with pytest.raises(nt.NttError):
        nt.multiply_limbs([plans[0], foreign], da[:2], db[:2], outs[:2], batch, out_stride=stride)       # two contexts
    small = ctx.plan(primes[0], n, batch - 1)
    with pytest.raises(nt.NttError):
        nt.multiply_limbs([small, plans[1]], da[:2], db[:2], outs[:2], batch, out_stride=stride)         # beyond a plan's capacity
    other.close()
    ctx.close()


@pytest.mark.parametrize("n", [2048, 8192, 65536])
def test_multiply_limbs_negacyclic(nt, n):
    """The limbs of an x^n + 1 ring in one call: every row equals ntt_negacyclic_multiply_batched's (itself checked against
    the folded oracle product in tests/test_gpu_negacyclic.py)."""
    import torch
    primes = [998244353, 469762049, 1224736769, 7340033, 104857601]
    batch = 5
    rng = np.random.default_rng(n + 1)
    ctx = nt.Context(0)
    plans = [ctx.negacyclic_plan(p, n, batch) for p in primes]
    ha = [rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32) for p in primes]
    hb = [rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32) for p in primes]
    ha[0][0, :] = primes[0] - 1
    hb[0][0, :] = primes[0] - 1
    da = [torch.from_numpy(x.view(np.int32)).cuda() for x in ha]
    db = [torch.from_numpy(x.view(np.int32)).cuda() for x in hb]
    outs = [torch.full((batch, n + 2), -1, dtype=torch.int32, device="cuda") for _ in primes]
    torch.cuda.synchronize()
    nt.multiply_limbs(plans, da, db, outs, batch, out_stride=n + 2)
    ctx.synchronize()
    for i, p in enumerate(primes):
        got = outs[i].cpu().numpy().view(np.uint32)
        assert np.all(got[:, n:] == 0xFFFFFFFF), "wrote past the row"
        assert np.array_equal(got[:, :n], plans[i].multiply(ha[i], hb[i], batch=batch)), (p, n)
    # the fold itself, once, on the smallest size: c = (a * b)[0:n] - (a * b)[n:2n-1]
    if n == 2048:
        lin = orc.poly_multiply(ha[1][2], hb[1][2], primes[1]).astype(np.int64)
        fold = (lin[:n] - np.concatenate([lin[n:], [0]])) % primes[1]
        assert np.array_equal(outs[1][2, :n].cpu().numpy().view(np.uint32), fold.astype(np.uint32))
    ctx.close()


def test_small_calls_on_pageable_arrays(nt, monkeypatch):
    """poly_multiply / ntt_poly_multiply_batched on the harness's kind of arrays: pageable ones go through page-locked
    buffers of the context and a CPU memcpy each way (NTT_AMD_PAGEABLE_STAGE=0: the hipMemcpy staging); page-locked and
    pageable arrays mixed in one call; a row stride with a gap keeps the caller's bytes."""
    p = 998244353
    rng = np.random.default_rng(77)
    ctx = nt.Context(0)
    for n, batch in ((131072, 1), (30000, 3), (5, 2)):
        a = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        b = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
        want = np.stack([orc.poly_multiply(a[r], b[r], p) for r in range(batch)])
        plan = ctx.plan(p, n, batch)
        stride = 2 * n + 5
        for stage in ("1", "0"):
            monkeypatch.setenv("NTT_AMD_PAGEABLE_STAGE", stage)
            for _ in range(2):                                  # twice: the context's buffers are reused
                out = np.full((batch, stride), 0xABCD1234, dtype=np.uint32)
                plan.multiply(a, b, out=out, batch=batch, out_stride=stride)
                assert np.array_equal(out[:, :2 * n - 1], want), (n, stage)
                assert np.all(out[:, 2 * n - 1:] == 0xABCD1234), "bytes between the rows were overwritten"
        monkeypatch.delenv("NTT_AMD_PAGEABLE_STAGE")
        ha = nt.host_array((batch, n))
        ha[...] = a
        mixed = np.zeros((batch, 2 * n - 1), dtype=np.uint32)   # page-locked a, pageable b and result
        plan.multiply(ha, b, out=mixed, batch=batch)
        assert np.array_equal(mixed, want)
        hout = nt.host_array((batch, 2 * n - 1))                # pageable operands, page-locked result
        plan.multiply(a, b, out=hout, batch=batch)
        assert np.array_equal(hout, want)
        plan.close()
    # the reference's own entry point on int arrays
    ia = rng.integers(0, p, size=70000, dtype=np.int64).astype(np.int32)
    ib = rng.integers(0, p, size=70000, dtype=np.int64).astype(np.int32)
    assert np.array_equal(nt.poly_multiply(ia, ib, p).view(np.uint32), orc.poly_multiply(ia.view(np.uint32), ib.view(np.uint32), p))
    ctx.close()


@pytest.mark.parametrize("limit,n,batch", [(1 << 30, 1500, 5), (1 << 30, 40000, 3), (1 << 30, 131072, 50),
                                           (1 << 31, 1500, 5), (1 << 31, 40000, 3), (1 << 31, 131072, 50)])
def test_two_products_per_reduction_at_the_range_ends(nt, ctx, limit, n, batch):
    """The radix-4 steps and fused passes of the Montgomery kernels add two signed products d0 w0 + d1 w1 before ONE
    reduction (modarith.hpp: mul2_acc; device_ntt.hpp: acc4_step, reg_ntt16_tw): the bound 2 * 2p * p / 2 <= p * 2^31 is
    tightest for the largest lazy prime (below 2^30) and the largest strict one (below 2^31).  Largest primes of both
    ranges with 2^18 | p - 1, rows of all p - 1, of alternating 0 / p - 1 (the largest differences) and random rows; the
    one-kernel product (L = 4096), a 32-point and a 64-point column level.  No oracle reaches these primes with its fixed
    generator 3: evaluation identities on every row plus an exact schoolbook check of the short product."""
    import torch
    from test_gpu_full_configs import _check_rows
    p = _largest_prime_below(limit, 18)
    assert p > limit - (1 << 26)
    rng = np.random.default_rng(n + batch + (limit >> 20))
    a = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
    b = rng.integers(0, p, size=(batch, n), dtype=np.int64).astype(np.uint32)
    a[0, :] = p - 1
    b[0, :] = p - 1
    a[1, ::2], a[1, 1::2] = p - 1, 0
    b[1, ::2], b[1, 1::2] = 0, p - 1
    a[2, :] = p - 1
    b[2, :] = 1
    da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
    plan = ctx.plan(p, n, batch, nt.MONTGOMERY)
    out = torch.full((batch, 2 * n - 1), -1, dtype=torch.int32, device="cuda")
    plan.multiply(da, db, out=out, batch=batch)
    ctx.synchronize()
    plan.close()
    _check_rows(da, db, out, 2 * n - 1, p, torch, rng)
    got = out.cpu().numpy().view(np.uint32)
    # row 0 in closed form: (p - 1)^2 = 1, so coefficient k of the product is min(k, 2n - 2 - k) + 1
    k = np.arange(2 * n - 1, dtype=np.int64)
    assert np.array_equal(got[0].astype(np.int64), (np.minimum(k, 2 * n - 2 - k) + 1) % p)
    # row 2: all p - 1 times all ones: coefficient k is -(min(k, 2n - 2 - k) + 1) mod p
    assert np.array_equal(got[2].astype(np.int64), (-(np.minimum(k, 2 * n - 2 - k) + 1)) % p)
    # exact schoolbook for the first 64 coefficients of two random rows (big-integer arithmetic)
    for r in (batch - 2, batch - 1):
        ar, br = [int(v) for v in a[r][:64]], [int(v) for v in b[r][:64]]
        assert [int(v) for v in got[r][:64]] == [sum(ar[i] * br[kk - i] for i in range(kk + 1)) % p for kk in range(64)]
