#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched poly_multiply at n = 2^17 with a 32-bit NTT prime.

A "step" is one pass of the hot path over one batch: B = 1024 independent products of two
degree-(n-1) polynomials, n = 131072, p = 998244353 (BASELINE.json configs[1]), inputs and outputs
resident in HBM.  With N GPUs every rank runs its own batch (independent units, no data-path
collective: "weak" scaling); the barrier and the max-over-ranks time go through torch.distributed
(RCCL when N > 1).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline     -- HBM roofline of the dominant kernel, timed with HIP events inside the library on
                  the stream the kernels run on (ntt_ctx_enable_kernel_timing)
  cpu_baseline -- the CPU oracle's restatement of the reference's OpenMP path (main-openmp.cc) timed
                  on this box's host cores on a bounded sample of the same workload (rank 0, N = 1);
                  this leg is the only place the oracle is touched: it also compares its product of
                  pair 0 with the GPU's.  Every rank checks A(x) * B(x) = AB(x) on its own output first.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def splitmix64(seed, count):
    """The synthetic-input generator of SURVEY 8(d): element i is the (i+1)-th splitmix64 output for `seed`."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed & ((1 << 64) - 1)) + np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def splitmix_fill(seed, rows, n, p):
    out = np.empty((rows, n), dtype=np.uint32)
    for r in range(rows):        # seed 0x20250711 + pair index, next() % p
        out[r] = (splitmix64(seed + r, n) % np.uint64(p)).astype(np.uint32)
    return out


def evaluation_identity(a, b, ab, p, points=(1, 3, 0x2545F491)):
    """Oracle-free check every rank runs on its own output: A(x) * B(x) = AB(x) (mod p) at a few points."""
    def horner(c, x):
        acc = 0
        for v in c[::-1].tolist():
            acc = (acc * x + v) % p
        return acc
    return int(ab.max()) < p and all(horner(a, x % p) * horner(b, x % p) % p == horner(ab, x % p) for x in points)


def cpu_baseline(n, p, threads, budget_s, gpu_row0):
    """The oracle's port of main-openmp.cc (per-stage block-parallel radix-2 NTT, 64-bit Montgomery),
    timed between the reference's Start/End stamps.  Checker code used as the reported baseline only;
    its product of pair 0 (the same seeds as row 0 of rank 0's batch) is also compared with the GPU's."""
    import orc
    a = (splitmix64(0x20250711, n) % np.uint64(p)).astype(np.uint64)
    b = (splitmix64(0x20250711 + (1 << 20), n) % np.uint64(p)).astype(np.uint64)
    want, _ = orc.poly_multiply_omp(a, b, p, threads=threads)          # warm-up
    if gpu_row0 is not None and not np.array_equal(np.asarray(want).astype(np.uint64), gpu_row0.astype(np.uint64)):
        print("bench: GPU result differs from the oracle", file=sys.stderr)
        sys.exit(1)
    t0 = time.time()
    secs, reps = 0.0, 0
    while time.time() - t0 < budget_s and reps < 200:
        _, s = orc.poly_multiply_omp(a, b, p, threads=threads)
        secs += s
        reps += 1
    out = {"value": reps / secs, "unit": "poly-mults/s", "cores": threads, "kind": "port",
           "sample": f"{reps} products of n={n}, p={p} (oracle port of main-openmp.cc:678-692, "
                     f"timer between its Start/End stamps), {threads} OpenMP threads",
           "ms_per_product": 1e3 * secs / reps}
    # also the reference's own hard-coded num_threads(8) and the single-thread
    # Montgomery / Barrett harness bodies (main-montgomery.cc:317-351, main-barrett.cc:249-282); a few
    # products each, whole call timed
    others = {}
    s8, r8 = 0.0, 0
    for _ in range(10):
        _, s = orc.poly_multiply_omp(a, b, p, threads=min(8, threads))
        s8 += s
        r8 += 1
    others[f"openmp_{min(8, threads)}_threads_ms"] = 1e3 * s8 / r8
    a32, b32 = a.astype(np.uint32), b.astype(np.uint32)
    for kind in ("mont", "barrett"):
        orc.poly_multiply(a32, b32, p, kind=kind)
        t1 = time.time()
        for _ in range(3):
            orc.poly_multiply(a32, b32, p, kind=kind)
        others[f"single_thread_{kind}_ms"] = 1e3 * (time.time() - t1) / 3
    out["others_ms_per_product"] = others
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--p", type=int, default=998244353)
    ap.add_argument("--barrett", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="products per kernel launch (0 = library default)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the result checks (timing-only ablation builds whose results are wrong on purpose)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="rendezvous backend for N > 1; gloo lets ranks share a GPU to rehearse the multi-rank flow on a one-GPU box")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (no roofline block)")
    args = ap.parse_args()

    import torch                      # device memory, streams and torch.distributed: plumbing
    import ntt_amd                    # the product: HIP kernels behind the C ABI
    from ntt_amd import shard

    rank, world, local_rank = shard.env_rank_world()
    if args.backend == "gloo":                        # rehearsal: ranks may share a GPU, exchanges on host tensors
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    group = shard.Group(args.backend, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)   # RCCL when world > 1

    B, n, p = args.batch, args.n, args.p
    stride = 2 * n
    seed = shard.rank_seed(0x20250711, rank, B)      # every rank multiplies different polynomials
    ha = splitmix_fill(seed, B, n, p)
    hb = splitmix_fill(seed + (1 << 20), B, n, p)
    da = torch.from_numpy(ha.view(np.int32)).cuda()
    db = torch.from_numpy(hb.view(np.int32)).cuda()
    dab = torch.zeros((B, stride), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    if args.chunk > 0:
        os.environ["NTT_AMD_CHUNK"] = str(args.chunk)
    ctx = ntt_amd.Context(local_rank)
    kind = ntt_amd.BARRETT if args.barrett else ntt_amd.MONTGOMERY
    plan = ctx.plan(p, n, B, kind)

    def step():
        plan.multiply(da, db, out=dab, batch=B, out_stride=stride)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    barrier = group.barrier

    for _ in range(args.warmup):
        step()
    sync()

    # every rank checks the last product of its warm-up output before timing anything (no oracle here:
    # the oracle comparison of pair 0 happens inside the cpu_baseline leg below)
    gpu_row0 = None
    if not args.no_check:
        row = B - 1
        got = dab[row, :2 * n - 1].cpu().numpy().view(np.uint32)
        if not evaluation_identity(ha[row], hb[row], got, p):
            print(f"bench: rank {rank}: A(x) * B(x) != AB(x) on the GPU result", file=sys.stderr)
            sys.exit(1)
        gpu_row0 = dab[0, :2 * n - 1].cpu().numpy().view(np.uint32).copy()

    # which kernel dominates (all kernels bracketed with events over two untimed steps); the timed region
    # then pays only for the events of the kernel it reports
    warm_ktimes = {}
    if not args.no_kernel_timing:
        ctx.enable_kernel_timing(True)
        for _ in range(2):
            step()
        sync()
        warm_ktimes = {k: v for k, v in ctx.kernel_times().items() if v[1] > 0}
        dominant = max(warm_ktimes.items(), key=lambda kv: kv[1][0])[0]
        ctx.kernel_timing_select(dominant)
    ctx.enable_kernel_timing(not args.no_kernel_timing)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = ctx.kernel_times()
    ctx.enable_kernel_timing(False)

    elapsed = group.max_over_ranks(elapsed)

    if rank == 0:
        units = world * B * args.steps
        value = units / elapsed
        alg_bytes = 16 * n - 4                      # read a, read b, write ab
        # dominant kernel = largest share of device time in the timed region
        if args.no_kernel_timing:
            print(json.dumps({"value": value, "ms_per_step": 1e3 * elapsed / args.steps, "chunk": args.chunk}), flush=True)
            plan.close()
            ctx.close()
            group.close()
            return
        name, (tot_ms, launches) = max(((k, v) for k, v in ktimes.items() if v[1] > 0),
                                       key=lambda kv: kv[1][0])
        units_per_launch = B * args.steps / launches
        avg_ms = tot_ms / launches
        achieved = alg_bytes * units_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            per_unit = json.load(open(tpath)).get(name, {}).get("bytes_per_unit")
            if per_unit:                 # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE) per launch, GB like achieved
                traffic = per_unit * units_per_launch
        out = {
            "metric": "poly-mults/sec at n=2^17 (32-bit p)",
            "value": value,
            "unit": "poly-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"batched {B}x poly_multiply, n={n}, p={p}, "
                                   f"{'Barrett' if args.barrett else 'Montgomery'} NTT, L={plan.length} "
                                   "(BASELINE.json configs[1]); splitmix64 coefficients, device-resident",
                       "batch_per_gpu": B, "n": n, "p": p, "sharding": f"batch x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": name, "avg_launch_ms": avg_ms, "launches": launches,
                         "units_per_launch": units_per_launch, "algorithmic_bytes_per_unit": alg_bytes,
                         "pipeline_achieved": value / world * alg_bytes / 1e9,
                         "pipeline_frac": value / world * alg_bytes / 1e9 / HBM_PEAK_GBPS,
                         "kernel_ms_per_step_untimed": {k: v[0] / 2 for k, v in warm_ktimes.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(n, p, threads, args.cpu_seconds, gpu_row0)
        print(json.dumps(out), flush=True)

    plan.close()
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
