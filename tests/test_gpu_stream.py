"""GPU tests of ntt_poly_multiply_stream (host operands streamed through a ring of device slots) against the
batched device path and the oracle: pinned and pageable host memory, counts that are not a multiple of
the slice, fewer slices than ring slots, strided output whose tail entries must survive."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

P = 998244353


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.fixture(scope="module")
def ctx(nt):
    c = nt.Context(0)
    yield c
    c.close()


def rand(rng, shape, p=P):
    return rng.integers(0, p, size=shape, dtype=np.int64).astype(np.uint32)


@pytest.mark.parametrize("count,cap", [(1, 4), (4, 4), (13, 4), (37, 8), (7, 16)])
def test_stream_pageable_matches_batched(ctx, count, cap):
    n = 4096
    rng = np.random.default_rng(count)
    a, b = rand(rng, (count, n)), rand(rng, (count, n))
    plan = ctx.plan(P, n, cap)
    got = plan.multiply_stream(a, b)
    big = ctx.plan(P, n, count)
    want = big.multiply(a, b)
    assert np.array_equal(got, want)
    w0 = orc.poly_multiply(a[count - 1], b[count - 1], P)
    assert np.array_equal(got[count - 1], np.asarray(w0).astype(np.uint32))
    plan.close()
    big.close()


def test_stream_pinned_strided_and_reuse(nt, ctx):
    """Pinned arrays (DMA straight from / into the caller's memory), output row stride 2n: entry 2n-1 of
    every row is not ours.  Two calls in a row on one context reuse the ring."""
    n, count, cap = 8192, 29, 8
    rng = np.random.default_rng(3)
    a = nt.host_array((count, n))
    b = nt.host_array((count, n))
    out = nt.host_array((count, 2 * n))
    a[:] = rand(rng, (count, n))
    b[:] = rand(rng, (count, n))
    out[:] = 0xABCD
    plan = ctx.plan(P, n, cap)
    plan.multiply_stream(a, b, out=out, out_stride=2 * n)
    ref = ctx.plan(P, n, count)
    want = ref.multiply(np.array(a), np.array(b))
    assert np.array_equal(out[:, :2 * n - 1], want)
    assert (out[:, 2 * n - 1] == 0xABCD).all()
    # second call, smaller count, pageable output with stride
    out2 = np.full((5, 2 * n), 7, dtype=np.uint32)
    plan.multiply_stream(a[:5], b[:5], out=out2, out_stride=2 * n)
    assert np.array_equal(out2[:, :2 * n - 1], want[:5])
    assert (out2[:, 2 * n - 1] == 7).all()
    plan.close()
    ref.close()


def test_stream_headline_shape_properties(nt, ctx):
    """n = 2^17 at a count above the ring (slices of 32): every 16th product against the oracle, all rows
    against the evaluation at x = 1."""
    n, count, cap = 131072, 100, 32
    a = (orc.splitmix64_np(11, count * n) % np.uint64(P)).astype(np.uint32).reshape(count, n)
    b = (orc.splitmix64_np(12, count * n) % np.uint64(P)).astype(np.uint32).reshape(count, n)
    plan = ctx.plan(P, n, cap)
    got = plan.multiply_stream(a, b)
    plan.close()
    sa = a.astype(np.uint64).sum(axis=1) % P
    sb = b.astype(np.uint64).sum(axis=1) % P
    sab = got.astype(np.uint64).sum(axis=1) % P
    assert np.array_equal(sa * sb % P, sab)
    for i in range(0, count, 16):
        want = orc.poly_multiply(a[i], b[i], P)
        assert np.array_equal(got[i], np.asarray(want).astype(np.uint32)), i


def test_stream_errors(nt, ctx):
    import torch
    plan = ctx.plan(P, 64, 2)
    a = np.zeros((3, 64), dtype=np.uint32)
    with pytest.raises(nt.NttError):
        plan.multiply_stream(a, a, out=np.zeros((3, 127), dtype=np.uint32), out_stride=100)    # stride < 2n-1
    d = torch.zeros((3, 64), dtype=torch.int32, device="cuda")
    dout = torch.zeros((3, 127), dtype=torch.int32, device="cuda")
    with pytest.raises(nt.NttError):          # device operands belong to ntt_poly_multiply_batched
        plan.multiply_stream(d, d, out=dout, out_stride=127)
    plan.close()
