"""ntt_poly_multiply_batched_noalloc (include/ntt_amd.h): the entry point that allocates nothing, through the
C ABI, against ntt_poly_multiply_batched on the same device arrays -- bit for bit, on two calls in a row, the
second with a smaller input, so that whatever the first call left in the static scratch array is there when
the second runs.  Collected last (file name): it shares the GPU's one scratch array with nothing else."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _rand(rng, rows, n, p):
    return rng.integers(0, p, size=(rows, n), dtype=np.int64).astype(np.uint32)


def test_noalloc_entry_matches_batched_entry():
    import torch
    import ntt_amd

    ctx = ntt_amd.Context(0)
    rng = np.random.default_rng(77)
    p = 998244353
    free_before = None
    # (n, batch of the first call, batch of the second): L = 2^18 takes 2 MiB of scratch per product, so 40 products
    # go through the 64 MiB array in two groups; L = 16384 has two column levels; n = 1000 and 4000 are the
    # one-kernel shapes that use no scratch at all
    for n, b1, b2 in ((131072, 40, 3), (5000, 9, 2), (4000, 5, 1), (1000, 4, 2)):
        ref_plan = ctx.plan(p, n, b1)
        plan = ctx.plan(p, n, 1)                       # capacity 1: the no-alloc entry does not use the plan's buffers
        stride = 2 * n + 3
        a = torch.from_numpy(_rand(rng, b1, n, p).view(np.int32)).cuda()
        b = torch.from_numpy(_rand(rng, b1, n, p).view(np.int32)).cuda()
        a[0, :] = p - 1                                # extreme values in the first product
        b[0, :] = p - 1
        want = torch.full((b1, stride), -7, dtype=torch.int32, device="cuda")
        got = torch.full((b1, stride), -7, dtype=torch.int32, device="cuda")
        ref_plan.multiply(a, b, out=want, batch=b1, out_stride=stride)
        ctx.synchronize()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        plan.multiply_noalloc(a, b, got, b1, out_stride=stride)          # first call
        ctx.synchronize()
        assert torch.equal(got, want), (n, b1)
        # one row against the CPU oracle as well (the batched entry point it is compared with is oracle-checked elsewhere;
        # this keeps the test from being a self-comparison): the row of extreme operands
        assert np.array_equal(got[0, :2 * n - 1].cpu().numpy().view(np.uint32),
                              orc.poly_multiply(a[0].cpu().numpy().view(np.uint32), b[0].cpu().numpy().view(np.uint32), p)), n
        # second call, smaller input: the last rows only, into a fresh output
        a2, b2_ = a[b1 - b2:].contiguous(), b[b1 - b2:].contiguous()
        torch.cuda.synchronize()
        free_mid = torch.cuda.mem_get_info()[0]
        got2 = torch.full((b2, stride), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        free_mid2 = torch.cuda.mem_get_info()[0]
        plan.multiply_noalloc(a2, b2_, got2, b2, out_stride=stride)     # second call
        ctx.synchronize()
        free_after = torch.cuda.mem_get_info()[0]
        assert torch.equal(got2, want[b1 - b2:]), (n, b2)
        assert free_after == free_mid2                                   # the call itself took no device memory
        assert free_mid <= free_before                                   # (only this test's own tensors did)
        # against the oracle-checked path's semantics: entries beyond 2n-1 of a row stay untouched
        assert int((got[:, 2 * n - 1:] != -7).sum()) == 0
        plan.close()
        ref_plan.close()

    # Barrett kernels and a prime above 2^30 (strict kernels) through the same entry
    for pp, kind in ((998244353, ntt_amd.BARRETT), (1224736769, ntt_amd.MONTGOMERY)):
        n, bt = 20000, 6
        ref_plan = ctx.plan(pp, n, bt, kind)
        a = torch.from_numpy(_rand(rng, bt, n, pp).view(np.int32)).cuda()
        b = torch.from_numpy(_rand(rng, bt, n, pp).view(np.int32)).cuda()
        want = torch.zeros((bt, 2 * n - 1), dtype=torch.int32, device="cuda")
        got = torch.zeros((bt, 2 * n - 1), dtype=torch.int32, device="cuda")
        ref_plan.multiply(a, b, out=want, batch=bt)
        ref_plan.multiply_noalloc(a, b, got, bt)
        ctx.synchronize()
        assert torch.equal(got, want), (pp, kind)
        ref_plan.close()

    # limits: host arrays and transforms beyond the scratch maximum are rejected, nothing is launched
    plan = ctx.plan(p, 4096, 1)
    host = np.zeros(4096, dtype=np.uint32)
    with pytest.raises(ntt_amd.NttError) as e:
        plan.multiply_noalloc(host, host, np.zeros(8191, dtype=np.uint32), 1)
    assert e.value.code == ntt_amd.ERR_INVALID and "device pointers" in str(e.value)
    plan.close()
    big = ctx.plan(469762049, 1 << 23, 1)              # L = 2^24: 128 MiB of intermediates per product
    x = torch.zeros(1 << 23, dtype=torch.int32, device="cuda")
    out = torch.zeros((1 << 24) - 1, dtype=torch.int32, device="cuda")
    with pytest.raises(ntt_amd.NttError) as e:
        big.multiply_noalloc(x, x, out, 1)
    assert e.value.code == ntt_amd.ERR_UNSUPPORTED and "scratch" in str(e.value)
    big.close()
    ctx.close()
