"""Worker for tests/test_gpu_multiprocess.py: one process per rank, every rank computing on the GPU through
the C ABI (all ranks share cuda:0 on a one-GPU box), the exchange steps over gloo on host tensors.  The
same drivers (ntt_amd.shard.crt_multiply_sharded, dist_transform) run over RCCL with device tensors when
every rank owns a GPU; this covers the process-per-rank plumbing around the real kernels."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ntt_amd  # noqa: E402
import orc  # noqa: E402
from ntt_amd import shard  # noqa: E402


def main():
    out_path = sys.argv[1]
    g = shard.Group("gloo")
    G, rank = g.world, g.rank
    ctx = ntt_amd.Context(0)
    ok = True

    # ---- 4-prime CRT, primes sharded over the ranks (main-mpi.cc:657-660), all-to-all of residue slices
    p64, n = 1337006139375617, 3000
    a = orc.splitmix64_np(1, n) % np.uint64(p64)
    b = orc.splitmix64_np(2, n) % np.uint64(p64)

    def residue_fn(k):
        return torch.from_numpy(ctx.crt_residue_multiply(a, b, k).view(np.int32))

    def merge_fn(rows, count):
        out = ctx.crt_merge([r.numpy().view(np.uint32) for r in rows], p64, count=count)
        return torch.from_numpy(out.view(np.int64))

    full = shard.crt_multiply_sharded(g, residue_fn, merge_fn, n)
    want, _ = orc.poly_multiply_crt(a, b, p64, threads=2)
    ok = ok and bool(np.array_equal(full.numpy().view(np.uint64), want))
    # the per-epoch form of the same exchange (preallocated buffers, products written in exchange order)
    xchg = shard.CrtExchange(g, ctx, n, p64)
    ta, tb = torch.from_numpy(a.view(np.int64).copy()), torch.from_numpy(b.view(np.int64).copy())
    for _ in range(2):
        ok = ok and bool(np.array_equal(xchg.multiply(ta, tb).numpy().view(np.uint64), want))

    # ---- one transform of length 2^16 split over the ranks: two all-to-alls around the GPU pieces
    p, lgL = 469762049, 16
    L = 1 << lgL
    S = L // G
    x = (orc.splitmix64_np(3, L) % np.uint64(p)).astype(np.uint32)
    X = orc.ntt(x, p)
    dp = ntt_amd.DistPlan(ctx, p, lgL, G, rank)

    def colpass(arr, inv):
        d = arr.contiguous().cuda()
        torch.cuda.synchronize()
        dp.colpass(d, inverse=inv)
        ctx.synchronize()
        return d.cpu()

    def local(row, inv):
        d = row.contiguous().cuda()
        torch.cuda.synchronize()
        dp.local(d, inverse=inv)
        ctx.synchronize()
        return d.cpu()

    a2a = shard.torch_all_to_all(g)
    mine = torch.from_numpy(x[rank * S:(rank + 1) * S].view(np.int32).copy())
    fwd = shard.dist_transform(G, rank, a2a, colpass, local, mine, inverse=False)
    ok = ok and bool(np.array_equal(fwd.numpy().view(np.uint32), X[rank::G]))
    back = shard.dist_transform(G, rank, a2a, colpass, local, fwd, inverse=True)
    ok = ok and bool(np.array_equal(back.numpy().view(np.uint32), x[rank * S:(rank + 1) * S]))
    # the same with preallocated ping-pong buffers (shard.DistExchange); exchanges through host mirrors over gloo
    dx = shard.DistExchange(g, ctx, dp, on_device_exchange=False)
    dmine = mine.cuda()
    for _ in range(2):
        f2 = dx.forward(dmine)
        torch.cuda.synchronize()
        ok = ok and bool(np.array_equal(f2.cpu().numpy().view(np.uint32), X[rank::G]))
        b2 = dx.inverse(f2.clone())
        torch.cuda.synchronize()
        ok = ok and bool(np.array_equal(b2.cpu().numpy().view(np.uint32), x[rank * S:(rank + 1) * S]))
    ctx.set_stream(None)
    dp.close()

    # ---- batch sharding as in bench.py: partition(), per-rank seeds, max over ranks
    units = 10
    start, count = shard.partition(units, G, rank)
    plan = ctx.plan(998244353, 4096, max(count, 1))
    done = 0
    for u in range(start, start + count):
        pa = (orc.splitmix64_np(100 + u, 4096) % np.uint64(998244353)).astype(np.uint32)
        pb = (orc.splitmix64_np(200 + u, 4096) % np.uint64(998244353)).astype(np.uint32)
        ok = ok and bool(np.array_equal(plan.multiply(pa, pb)[0], orc.poly_multiply(pa, pb, 998244353)))
        done += 1
    plan.close()
    ok = ok and int(g.sum_over_ranks(done)) == units
    ctx.close()

    flags = [None] * G
    g.dist.all_gather_object(flags, ok)
    if rank == 0:
        open(out_path, "w").write("ok" if all(flags) else "mismatch %r" % (flags,))
    g.close()


if __name__ == "__main__":
    main()
