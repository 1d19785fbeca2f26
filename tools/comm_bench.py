#!/usr/bin/env python3
"""tools/comm_bench.py -- the process-per-GPU communicator (ntt_comm, ntt_amd/csrc/comm.cpp) timed on its own job:
launched with torch.distributed.run, one rank per GPU; rank 0 prints one JSON line.

torch.distributed is only the host program here (rendezvous, handing the RCCL id round, the final check gather): with
--backend nccl every rank joins an RCCL communicator INSIDE the library (ntt_comm_create_rccl) and the exchanges are
grouped ncclSend / ncclRecv on device arrays; with --backend gloo (ranks may share a GPU) the library calls back for
an all-to-all on host memory (ntt_comm_create_host), gloo standing in for MPI_Alltoall.

bench.py --gpus N runs this as a CHILD job of rank 0 at the end of its `also` block: RCCL mode has only ever run at
world = 1 by us, so a fault here must not take the headline line along.  Workloads: (2) the 4-prime CRT
product at n = 2^17, prime per rank; (3) one 2^26-point transform forward + inverse, block r on rank r.  Checks are
oracle-free: C(1), C(-1) of the product with Python integers; the transform's round trip and one output point."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"])
    ap.add_argument("--lgL", type=int, default=26)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import ntt_amd
    import bench
    from ntt_amd import shard

    rank, world, local_rank = shard.env_rank_world()
    dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    g = shard.Group("gloo")                                  # host-side only: the data path is the library's
    ctx = ntt_amd.Context(dev)
    if args.backend == "nccl":
        ids = [ntt_amd.Comm.rccl_id() if rank == 0 else None]
        if g.dist is not None:
            g.dist.broadcast_object_list(ids, src=0)         # MPI_Bcast of 128 bytes in a C++ host program
        comm = ntt_amd.Comm.rccl(ctx, ids[0], world, rank)
    else:
        def alltoall(send, recv):
            if g.dist is None:
                recv[...] = send
            else:
                g.dist.all_to_all_single(torch.from_numpy(recv).view(-1), torch.from_numpy(send).view(-1))
        comm = ntt_amd.Comm.host(ctx, alltoall, world, rank)
    out = {"comm_kind": comm.kind, "comm_world": world}
    failed = []

    # (2) large-modulus product, prime per rank
    n, p64 = 131072, 1337006139375617
    a = bench.splitmix64(0xC47A, n) % np.uint64(p64)
    b = bench.splitmix64(0xC47B, n) % np.uint64(p64)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    dout = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    comm.crt_multiply(da, db, p64, out=dout)
    ctx.synchronize()
    got = dout.cpu().numpy().view(np.uint64)

    def at(c, sign):
        v = c.tolist()
        return (sum(v[0::2]) + sign * sum(v[1::2])) % p64
    if int(got.max()) >= p64 or any(at(a, s) * at(b, s) % p64 != at(got, s) for s in (1, -1)):
        failed.append("crt identity")
    g.barrier()
    t0 = time.perf_counter()
    for _ in range(args.reps * 4):
        comm.crt_multiply(da, db, p64, out=dout)
    ctx.synchronize()
    out["comm_crt4_n2p17_ms"] = 1e3 * g.max_over_ranks(time.perf_counter() - t0) / (args.reps * 4)

    # (3) one transform split over the ranks
    if world >= 2 and world & (world - 1) == 0 and world <= 16:
        p, lgL = 469762049, args.lgL
        L, S = 1 << lgL, (1 << lgL) // world
        w = pow(3, (p - 1) // L, p)
        dp = ntt_amd.DistPlan(ctx, p, lgL, world, rank)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(26 + rank)
        x = torch.randint(0, p, (S,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        X, back = torch.zeros_like(x), torch.zeros_like(x)
        torch.cuda.synchronize()
        comm.dist_forward(dp, x, X)
        comm.dist_inverse(dp, X, back)
        ctx.synchronize()
        if not torch.equal(back, x):
            failed.append("ntt round trip")
        # X[rank + world * k2] on this rank: one point evaluated directly needs the whole input -- gathered over the host
        parts = [None] * world
        g.dist.all_gather_object(parts, x.cpu().numpy())
        if rank == 0:
            k2 = 0x2B3C5D7 % S
            full = torch.from_numpy(np.concatenate(parts)).cuda()
            if int(X[k2].item()) != bench._eval_point_t(torch, full, pow(w, world * k2, p), p):
                failed.append("ntt point")
            del full
        g.barrier()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            comm.dist_forward(dp, x, X)
            comm.dist_inverse(dp, X, back)
        ctx.synchronize()
        dt = g.max_over_ranks(time.perf_counter() - t0) / args.reps
        out[f"comm_ntt2p{lgL}_fwd_inv_ms"] = 1e3 * dt
        out[f"comm_ntt2p{lgL}_GBps"] = 16 * L / dt / 1e9
        dp.close()
    comm.close()
    ctx.close()
    nfailed = g.sum_over_ranks(float(len(failed)))
    out["comm_checks"] = "ok" if nfailed == 0 else "FAILED on some rank: " + ", ".join(failed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    g.close()
    return 0 if nfailed == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
