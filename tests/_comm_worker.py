"""Worker for tests/test_gpu_comm.py: one process per rank (torch.distributed.run), every rank with its own
single-device context and ONE ntt_comm in host mode -- the library runs the exchanges of the split transform and
of the prime-per-rank large-modulus product itself and calls back into the host program for the all-to-all on host
memory, here torch.distributed over gloo standing in for the MPI_Alltoall of a C++ rank (main-mpi.cc's shape).
All ranks share cuda:0 on a one-GPU box."""
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
    calls = []

    def alltoall(send, recv):                              # uint8 [world, bytes_per_rank] views of pinned host memory
        calls.append(send.shape[1])
        g.dist.all_to_all_single(torch.from_numpy(recv).view(-1), torch.from_numpy(send).view(-1))

    comm = ntt_amd.Comm.host(ctx, alltoall, G, rank)
    ok = comm.world == G and comm.rank == rank and comm.kind == "host"
    why = []

    def check(cond, what):
        nonlocal ok
        if not cond:
            ok = False
            why.append(what)

    # ---- the reference's 51-bit case shape: primes sharded by main-mpi.cc:657-660, slices exchanged, merged, gathered
    p64 = 1337006139375617
    for n in (3000, 131072, 1):
        a = orc.splitmix64_np(1 + n, n) % np.uint64(p64)
        b = orc.splitmix64_np(2 + n, n) % np.uint64(p64)
        got = comm.crt_multiply(a, b, p64)                 # host arrays in, host array out: complete on return
        if n <= 3000:
            want, _ = orc.poly_multiply_crt(a, b, p64, threads=2)
        else:                                              # the single-context product of the same library (oracle-checked elsewhere)
            want = ctx.crt_multiply(a, b, p64)
        check(np.array_equal(got, want), f"crt host arrays n={n}")
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dout = torch.full((2 * n - 1,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        comm.crt_multiply(da, db, p64, out=dout)
        ctx.synchronize()
        check(np.array_equal(dout.cpu().numpy().view(np.uint64), want), f"crt device arrays n={n}")
    # every rank rejects together, before any exchange
    try:
        comm.crt_multiply(a, b, 1 << 61)
        check(False, "even modulus accepted")
    except ntt_amd.NttError as e:
        check(e.code == ntt_amd.ERR_UNSUPPORTED, "even modulus: wrong code")

    # ---- one transform split over the ranks, both all-to-alls inside ntt_comm_dist_forward / _inverse
    for p, lgL, kind in ((469762049, 16, ntt_amd.MONTGOMERY), (998244353, 18, ntt_amd.MONTGOMERY), (1224736769, 16, ntt_amd.BARRETT)):
        L = 1 << lgL
        S = L // G
        x = (orc.splitmix64_np(3 + lgL, L) % np.uint64(p)).astype(np.uint32)
        X = orc.ntt(x, p)
        dp = ntt_amd.DistPlan(ctx, p, lgL, G, rank, kind)
        mine = torch.from_numpy(x[rank * S:(rank + 1) * S].view(np.int32).copy()).cuda()
        keep = mine.clone()
        fwd = torch.zeros(S, dtype=torch.int32, device="cuda")
        back = torch.zeros(S, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):                                 # twice: the staging buffers are reused
            ncalls = len(calls)
            comm.dist_forward(dp, mine, fwd)
            ctx.synchronize()
            check(len(calls) == ncalls + 2 and calls[-1] == 4 * S // G, "two all-to-alls of S / world words per peer")
            check(np.array_equal(fwd.cpu().numpy().view(np.uint32), X[rank::G]), f"forward p={p} lgL={lgL}")
            check(torch.equal(mine, keep), "forward modified its input")
            comm.dist_inverse(dp, fwd, back)
            ctx.synchronize()
            check(np.array_equal(back.cpu().numpy().view(np.uint32), x[rank * S:(rank + 1) * S]), f"inverse p={p} lgL={lgL}")
        # the split product: two polynomials of L / 2 coefficients, blocks on the lower half of the ranks
        half = L // 2
        pa = (orc.splitmix64_np(5 + lgL, half) % np.uint64(p)).astype(np.uint32)
        pb = (orc.splitmix64_np(6 + lgL, half) % np.uint64(p)).astype(np.uint32)
        pa[:3] = p - 1
        pb[-3:] = p - 1
        want = np.concatenate([orc.poly_multiply(pa, pb, p), np.zeros(1, dtype=np.uint32)])
        blk = lambda v: torch.from_numpy(v[rank * S:(rank + 1) * S].view(np.int32).copy()).cuda() if rank < G // 2 else None
        da, db = blk(pa), blk(pb)
        dab = torch.full((S,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            comm.dist_multiply(dp, da, db, dab)
            ctx.synchronize()
            check(np.array_equal(dab.cpu().numpy().view(np.uint32), want[rank * S:(rank + 1) * S]), f"split product p={p} lgL={lgL}")
        # argument checks: host arrays, in == out, a plan of another rank
        for bad in (lambda: comm.dist_forward(dp, mine, mine), lambda: comm.dist_forward(dp, mine.cpu().numpy().view(np.uint32), fwd)):
            try:
                bad()
                check(False, "bad arguments accepted")
            except ntt_amd.NttError as e:
                check(e.code == ntt_amd.ERR_INVALID, "bad arguments: wrong code")
        other = ntt_amd.DistPlan(ctx, p, lgL, G, (rank + 1) % G, kind)
        try:
            comm.dist_forward(other, mine, fwd)
            check(False, "plan of another rank accepted")
        except ntt_amd.NttError as e:
            check(e.code == ntt_amd.ERR_INVALID, "plan of another rank: wrong code")
        other.close()
        dp.close()

    # a failing callback fails the call (NTT_ERR_IO), on every rank alike
    def broken(send, recv):
        raise RuntimeError("link down (test)")
    bad = ntt_amd.Comm.host(ctx, broken, G, rank)
    try:
        sys.stderr = open(os.devnull, "w")                 # the wrapper prints the traceback of the callback
        bad.crt_multiply(a, b, p64)
        check(False, "failing all-to-all ignored")
    except ntt_amd.NttError as e:
        check(e.code == ntt_amd.ERR_IO, "failing all-to-all: wrong code")
    finally:
        sys.stderr = sys.__stderr__
    bad.close()
    comm.close()
    ctx.close()

    flags = [None] * G
    g.dist.all_gather_object(flags, (ok, why))
    if rank == 0:
        open(out_path, "w").write("ok" if all(f[0] for f in flags) else "mismatch %r" % (flags,))
    g.close()


if __name__ == "__main__":
    main()
