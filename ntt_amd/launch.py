"""Process-per-GPU harness with the rank layout of the reference's MPI program (main-mpi.cc:642-819,
SURVEY 8f rank 4), on top of the C ABI:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
        -m ntt_amd.launch --data nttdata --out files [--begin 0 --end 4 --epochs 50]

Per case (main-mpi.cc:676-814): rank 0 reads ``<id>.in`` (fRead, :680-683); n, p, a and b are broadcast
(:685-688); then ``epochs`` timed repeats (:706-802):

* p < 2^50 (:710-738): rank 0 alone multiplies, the other ranks idle -- here through ``ntt_plan`` when p
  fits 32 bits and ``ntt_plan64`` otherwise;
* larger p (:739-797): the four CRT primes are partitioned over the ranks (:657-660), every rank multiplies
  the residues of its primes, and where the reference gathers the residue polynomials to rank 0
  (:772-791) and merges there (:793-796), the ranks exchange coefficient slices in one all-to-all, every
  rank merges its slice on its GPU and the slices are all-gathered (``shard.crt_multiply_sharded``).

Rank 0 then checks against ``<id>.out`` (fCheck, :810), prints the reference's latency line (:811) and
writes ``<out>/<id>.out`` (fWrite, :812).  Unlike the reference the exit status reports a wrong or missing
case.  Backend "nccl" (RCCL, one GPU per rank, device tensors end to end) is the default on a GPU box;
"gloo" moves the exchanges through host tensors, which also lets several ranks share one GPU for a
rehearsal (rank r uses GPU r modulo the device count).
"""
import argparse
import sys
import time

import numpy as np


def _bcast_case(group, case, device):
    """main-mpi.cc:685-688: n and p, then the operands, from rank 0 to every rank (u64 as int64 bit patterns)."""
    import torch
    head = torch.zeros(2, dtype=torch.int64, device=device)
    if group.rank == 0 and case is not None:
        head[0], head[1] = case[0], np.uint64(case[1]).astype(np.int64).item()
    if group.dist is not None:
        group.dist.broadcast(head, src=0)
    n, p = int(head[0].item()), int(np.int64(head[1].item()).astype(np.uint64))
    if n <= 0:
        return 0, 0, None, None
    ops = torch.zeros((2, n), dtype=torch.int64, device=device)
    if group.rank == 0:
        ops[0] = torch.from_numpy(case[2].astype(np.uint64).view(np.int64)).to(device)
        ops[1] = torch.from_numpy(case[3].astype(np.uint64).view(np.int64)).to(device)
    if group.dist is not None:
        group.dist.broadcast(ops, src=0)
    return n, p, ops[0], ops[1]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ntt_amd.launch")
    ap.add_argument("--data", default="nttdata", help="directory of <id>.in / <id>.out (reference: /nttdata/)")
    ap.add_argument("--out", default="files", help="directory for <id>.out results (reference: files/)")
    ap.add_argument("--begin", type=int, default=0)
    ap.add_argument("--end", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=50, help="timed repeats per case (main-mpi.cc:706)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--bin", action="store_true", help="packed <id>.inb / <id>.outb cases (ntt_pack_case)")
    args = ap.parse_args(argv)

    import torch
    import ntt_amd
    from ntt_amd import shard

    rank, world, local_rank = shard.env_rank_world()
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("ntt_amd.launch: no GPU visible", file=sys.stderr)
        return 2
    if args.backend == "nccl" and world > ndev:
        print(f"ntt_amd.launch: {world} ranks over RCCL need {world} GPUs, {ndev} visible "
              "(use --backend gloo to share GPUs)", file=sys.stderr)
        return 2
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    on_gpu = args.backend == "nccl"
    device = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")
    group = shard.Group(args.backend, device=device if on_gpu else None)
    ctx = ntt_amd.Context(dev_index)
    if on_gpu:
        # one stream for torch's collectives / fills and the library's kernels: the all-to-all's output is
        # ordered before the merge that reads it and the residue kernels before the exchange that sends them,
        # without a host synchronisation in between
        ctx.use_torch_stream()
    if rank == 0:
        print(f"{world} ranks, one GPU each ({args.backend})" if on_gpu else
              f"{world} ranks on {min(world, ndev)} GPU(s) ({args.backend})", flush=True)

    failures = 0
    for cid in range(args.begin, args.end + 1):
        case = None
        if rank == 0:
            try:
                case = (ntt_amd.read_case_bin if args.bin else ntt_amd.read_case)(args.data, cid)
            except ntt_amd.NttError as e:
                print(f"case {cid} skipped: {e}", file=sys.stderr, flush=True)
        n, p, a, b = _bcast_case(group, case, device)
        if n == 0:
            failures += 1
            continue

        single = p < (1 << 50)                                   # main-mpi.cc:710
        plan, ab = None, None
        if single and rank == 0:
            if p < (1 << 32):
                plan = ctx.plan(p, n)
                a32, b32 = a.to(torch.int32), b.to(torch.int32)  # values < 2^32: the low words
                ab = torch.zeros(2 * n - 1, dtype=torch.int32, device=device)
                a_op, b_op = (a32, b32) if on_gpu else (a32.numpy().view(np.uint32), b32.numpy().view(np.uint32))
            else:
                plan = ctx.plan64(p, n)
                ab = torch.zeros(2 * n - 1, dtype=torch.int64, device=device)
                a_op, b_op = (a, b) if on_gpu else (a.numpy().view(np.uint64), b.numpy().view(np.uint64))
            out_op = ab if on_gpu else ab.numpy().view(np.uint32 if p < (1 << 32) else np.uint64)
        if not single:
            if world == 1:                                       # all four primes and the merge on this rank
                a_op, b_op = (a, b) if on_gpu else (a.numpy().view(np.uint64), b.numpy().view(np.uint64))
                ab = torch.zeros(2 * n - 1, dtype=torch.int64, device=device)
                crt_out = ab if on_gpu else ab.numpy().view(np.uint64)
            else:                                                # buffers of the exchange allocated once per case
                xchg = shard.CrtExchange(group, ctx, n, p, device=device)

        def epoch():
            nonlocal ab
            if single:
                if rank == 0:
                    plan.multiply(a_op, b_op, out=out_op)
                    ctx.synchronize()
            elif world == 1:
                ctx.crt_multiply(a_op, b_op, p, out=crt_out)
                ctx.synchronize()
            else:
                ab = xchg.multiply(a, b)                       # residues -> all-to-all -> merge -> all-gather, one stream
                if on_gpu:
                    torch.cuda.synchronize()

        if on_gpu:
            torch.cuda.synchronize()
        epoch()                                                  # untimed: code objects, tables, workspaces
        total_ms = 0.0
        for _ in range(args.epochs):
            group.barrier()
            t0 = time.perf_counter()                             # main-mpi.cc:708
            epoch()
            total_ms += 1e3 * (time.perf_counter() - t0)         # main-mpi.cc:799-801
        if plan is not None:
            plan.close()

        if rank == 0:                                            # main-mpi.cc:804-813
            res = ab.cpu().numpy()
            res = res.view(np.uint32) if res.dtype == np.int32 else res.view(np.uint64)
            sys.stdout.flush()
            try:
                if args.bin:
                    ok = ntt_amd.check_case_bin(args.data, cid, res if p >= (1 << 32) else res.astype(np.uint32))
                else:
                    ok = ntt_amd.check_case(args.data, cid, res.astype(np.uint64))
            except ntt_amd.NttError as e:                        # no (readable) expected output: not a pass
                print(f"case {cid} not checked: {e}", file=sys.stderr, flush=True)
                ok = False
            failures += 0 if ok else 1
            print(f"average latency for n = {n} p = {p} : {total_ms / max(args.epochs, 1):g} (us)", flush=True)
            if args.bin:
                ntt_amd.write_case_bin(args.out, cid, res if p >= (1 << 32) else res.astype(np.uint32), p)
            else:
                ntt_amd.write_case(args.out, cid, res.astype(np.uint64))

    failures = int(group.max_over_ranks(failures))
    ctx.close()
    group.close()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
