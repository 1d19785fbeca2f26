#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched poly_multiply at n = 2^17 with a 32-bit NTT prime.

A "step" is one pass of the hot path over one batch: B = 1024 independent products of two
degree-(n-1) polynomials, n = 131072, p = 998244353 (BASELINE.json configs[1]), inputs and outputs
resident in HBM.  With N GPUs every rank runs its own batch (independent units, no data-path
collective: "weak" scaling); the barrier and the max-over-ranks time go through torch.distributed
(RCCL when N > 1).

`python bench.py --gpus N` starts its N ranks itself when it is not already running under
torch.distributed.run (WORLD_SIZE unset): it launches `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a CHILD process before anything has touched the GPU, forwards
the child's output and exits with its status -- the way the reference's distributed program brings
up its own ranks (main-mpi.cc:642-665).  A rank whose torch.distributed world differs from --gpus
exits non-zero.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline     -- HBM roofline of the dominant kernel, timed with HIP events inside the library on
                  the stream the kernels run on (ntt_ctx_enable_kernel_timing); `valu` = the integer
                  VALU roofline beside it (the bound DESIGN.md section 5 argues), `kernels_one_lane`
                  = per-kernel time on one stream lane (uncontended), `sustained` = the same step
                  loop over >= 2 s after the timed region (board power settled)
  also         -- the other sharded workloads of SURVEY 8(e) behind the headline loop, timed the same way (barrier,
                  synchronise, max over ranks), flat scalars: the prime-per-rank CRT product at n = 2^17
                  (shard.CrtExchange; main-mpi.cc:706-802), 32 RNS limbs x 256 pairs at n = 2^16 split by limb, one
                  2^26-point transform forward + inverse split over the ranks (shard.DistExchange); every rank checks
                  an oracle-free identity, rank 0 one oracle row each.  --no-also skips it; a watchdog prints the line
                  without it if an exchange hangs
  cpu_baseline -- the CPU oracle's restatement of the reference's OpenMP path (main-openmp.cc) timed
                  on this box's host cores on a bounded sample of the same workload (rank 0, N = 1);
                  this leg is the only place the oracle is touched: it also compares its product of
                  pair 0 with the GPU's.  Every rank checks A(x) * B(x) = AB(x) on rows of its own
                  output first.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# MI355X_MICROARCH.md "Wave scheduling": 256 CUs x 4 SIMD-32 at 2.4 GHz; a wave64 VALU instruction
# issues over 2 cycles (v_add/v_sub/logic) or 4 (v_mul_lo/hi, v_mad_u64_u32, v_min, shifts:
# profiles/ubench_intmul_r01.txt, ubench_valu2_r01.txt)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
# measured floors of the two resources the pipeline shares (one MI355X, round 2):
#   profiles/ubench_mix_r02.txt          the butterfly's own instruction mix (add, borrow-subtract, select, sub, 3
#                                        multiply instructions) in a register-only loop, 7 x 4 waves per CU:
#                                        40.6 T lane-op/s at an in-kernel clock of 2.35 GHz
#   profiles/ubench_pipeline_mem_r02.txt three kernels that move the pipeline's bytes with its access patterns and
#                                        hand-off buffers and compute nothing: 1.24 us per product (48 per launch, 2 lanes)
MEASURED_MIX_PEAK_TLANEOPS = 40.6
MEASURED_TRAFFIC_FLOOR_US = 1.244
#   profiles/ubench_clock_r02.txt        the same butterfly loop BESIDE a second stream that keeps the memory system as
#                                        busy as this pipeline does: 33.6 T (Infinity-Cache-resident copy) to 34.6 T
#                                        (HBM copy) lane-op/s -- the VALU peak the pipeline can reach while it moves
#                                        its bytes ("coupled" peak; mean of the two)
COUPLED_MIX_PEAK_TLANEOPS = 34.1


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


def usable_cores():
    """(cores this process may run on, what limits them, CPUs the machine reports).  On the GPU boxes
    os.cpu_count() is the whole machine; the job's share is what the affinity mask, the cgroup quota
    and the OMP_NUM_THREADS the box exports allow."""
    total = os.cpu_count() or 1
    limits = {"os.cpu_count": total}
    try:
        limits["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            limits["cgroup cpu.max"] = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        limits["OMP_NUM_THREADS"] = int(env)
    name = min(limits, key=lambda k: limits[k])
    return limits[name], name, total


def cpu_baseline(n, p, budget_s, gpu_row0, with_ref=False):
    """The oracle's port of main-openmp.cc (per-stage block-parallel radix-2 NTT, 64-bit Montgomery),
    timed between the reference's Start/End stamps.  Checker code used as the reported baseline only;
    its product of pair 0 (the same seeds as row 0 of rank 0's batch) is also compared with the GPU's.
    Reported at every core this job may use (SURVEY 8d ii) and at the reference's own num_threads(8)."""
    import orc
    cores, limited_by, machine = usable_cores()
    a = (splitmix64(0x20250711, n) % np.uint64(p)).astype(np.uint64)
    b = (splitmix64(0x20250711 + (1 << 20), n) % np.uint64(p)).astype(np.uint64)
    want, _ = orc.poly_multiply_omp(a, b, p, threads=cores)          # warm-up
    if gpu_row0 is not None and not np.array_equal(np.asarray(want).astype(np.uint64), gpu_row0.astype(np.uint64)):
        print("bench: GPU result differs from the oracle", file=sys.stderr)
        sys.exit(1)

    def timed(threads, budget, cap):
        orc.poly_multiply_omp(a, b, p, threads=threads)
        t0, secs, reps = time.time(), 0.0, 0
        while time.time() - t0 < budget and reps < cap:
            _, s = orc.poly_multiply_omp(a, b, p, threads=threads)
            secs += s
            reps += 1
        return secs / reps, reps

    per, reps = timed(cores, budget_s * 0.6, 200)
    out = {"value": 1.0 / per, "unit": "poly-mults/s", "cores": cores, "kind": "port",
           "sample": f"{reps} products of n={n}, p={p} (oracle port of main-openmp.cc:678-692, "
                     f"timer between its Start/End stamps), {cores} OpenMP threads = every core this job may use "
                     f"(limited by {limited_by}; the machine reports {machine} CPUs)",
           "ms_per_product": 1e3 * per, "machine_cpus": machine, "cores_limited_by": limited_by}
    t8 = min(8, cores)
    per8, reps8 = timed(t8, budget_s * 0.25, 100)
    out["reference_threads"] = {"threads": t8, "value": 1.0 / per8, "ms_per_product": 1e3 * per8, "products": reps8,
                                "note": "main-openmp.cc hard-codes num_threads(8) (:397)"}
    # single-thread Montgomery / Barrett harness bodies (main-montgomery.cc:317-351, main-barrett.cc:249-282),
    # a few products each, whole call timed
    others = {}
    a32, b32 = a.astype(np.uint32), b.astype(np.uint32)
    for kind in ("mont", "barrett"):
        orc.poly_multiply(a32, b32, p, kind=kind)
        t1 = time.time()
        for _ in range(2):
            orc.poly_multiply(a32, b32, p, kind=kind)
        others[f"single_thread_{kind}_ms"] = 1e3 * (time.time() - t1) / 2
    if with_ref and orc.have_ref() and p < (1 << 31):
        # the reference's own main-baseline.cc multiply body, compiled in place (oracle/_ref, kind "reference")
        ia, ib = a32.view(np.int32).copy(), b32.view(np.int32).copy()
        iab = np.zeros(2 * n - 1, dtype=np.int32)
        orc.ref().ref_poly_multiply_ntt(ia, ib, iab, n, int(p))
        t1 = time.time()
        orc.ref().ref_poly_multiply_ntt(ia, ib, iab, n, int(p))
        others["reference_main_baseline_cc_single_thread_ms"] = 1e3 * (time.time() - t1)
        if gpu_row0 is not None and not np.array_equal(iab.view(np.uint32), gpu_row0):
            print("bench: GPU result differs from the compiled reference (main-baseline.cc)", file=sys.stderr)
            sys.exit(1)
    out["others_ms_per_product"] = others
    return out


def rns_primes(count, two_adicity, above):
    """The `count` smallest primes p > above with p = 1 mod 2^two_adicity (SURVEY 8d, configs[3])."""
    def is_prime(m):
        d, r = m - 1, 0
        while d % 2 == 0:
            d //= 2
            r += 1
        for a in (2, 3, 5, 7, 11, 13, 17):             # deterministic below 3.4e14
            x = pow(a, d, m)
            if x in (1, m - 1):
                continue
            for _ in range(r - 1):
                x = x * x % m
                if x == m - 1:
                    break
            else:
                return False
        return True
    out, step = [], 1 << two_adicity
    p = (above // step + 1) * step + 1
    while len(out) < count:
        if is_prime(p):
            out.append(p)
        p += step
    return out


def _powers_t(torch, x, count, p):
    """[x^0 .. x^(count-1)] mod p as an int64 device tensor (doubling)."""
    pw = torch.ones(1, dtype=torch.int64, device="cuda")
    xk = x % p
    while pw.numel() < count:
        pw = torch.cat([pw, pw * xk % p])
        xk = xk * xk % p
    return pw[:count]


def _eval_rows_t(torch, t, x, p, rows_per_step=64):
    """Every row of the int32 device tensor t (residues < p < 2^31) evaluated at x, mod p (terms < 2^62, sums < 2^49)."""
    pw = _powers_t(torch, x, t.shape[1], p)
    out = []
    for r0 in range(0, t.shape[0], rows_per_step):
        out.append(((t[r0:r0 + rows_per_step].to(torch.int64) * pw) % p).sum(dim=1) % p)
    return torch.cat(out)


def _eval_point_t(torch, x, wk, p, blk=1 << 13):
    """sum x[i] wk^i mod p for a long int32 device vector (length a multiple of blk): one output point of a transform."""
    pw = _powers_t(torch, wk, blk, p)
    sums = ((x.view(-1, blk).to(torch.int64) * pw) % p).sum(dim=1) % p
    pws = _powers_t(torch, pow(wk, blk, p), sums.numel(), p)
    return int(((sums * pws) % p).sum() % p)


def also_block(args, group, rank, world, local_rank, out=None):
    """The sharded workloads of SURVEY 8(e) beside the headline one (flat dict of scalars): (a) BASELINE.json
    configs[2], the 4-prime CRT product at n = 2^17 with the primes partitioned over the ranks and the slice exchange
    (main-mpi.cc:657-660, 739-797) through shard.CrtExchange; (b) configs[3], 32 RNS limbs x 256 pairs at n = 2^16, limbs
    partitioned over the ranks, no exchange; (c) configs[4], one 2^26-point transform forward + inverse split over the
    ranks through shard.DistExchange (one rank: the single-GPU transform).  Each: milliseconds (max over ranks),
    algorithmic GB/s, bytes a rank sends per unit.  Every rank checks an oracle-free identity; rank 0 also compares one
    row / point with the CPU oracle.  Runs on a context of its own (CrtExchange moves it to torch's stream)."""
    import torch
    import ntt_amd
    import orc
    from ntt_amd import shard
    out, failed = ({} if out is None else out), []         # `out` fills as the entries finish (the caller's watchdog reads it)
    on_device = args.backend == "nccl" or group.dist is None
    ctx = ntt_amd.Context(local_rank)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    def timed(fn, reps, warm=1):
        for _ in range(warm):
            fn()
        sync()
        group.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        sync()
        group.barrier()
        return group.max_over_ranks(time.perf_counter() - t0) / reps

    # (a) large-modulus product, primes over the ranks + one exchange
    n, p64 = 131072, 1337006139375617
    a = splitmix64(0xC47A, n) % np.uint64(p64)
    b = splitmix64(0xC47B, n) % np.uint64(p64)
    ta, tb = torch.from_numpy(a.view(np.int64)), torch.from_numpy(b.view(np.int64))
    if on_device:
        ta, tb = ta.cuda(), tb.cuda()
    ex = shard.CrtExchange(group, ctx, n, p64, device="cuda" if on_device else "cpu")
    got = ex.multiply(ta, tb)
    sync()
    got = got.cpu().numpy().view(np.uint64).copy()

    def at(c, sign):                                  # C(1) and C(-1) mod p64 with Python integers
        v = c.tolist()
        return (sum(v[0::2]) + sign * sum(v[1::2])) % p64
    if int(got.max()) >= p64 or any(at(a, s) * at(b, s) % p64 != at(got, s) for s in (1, -1)):
        failed.append("crt identity")
    if rank == 0:
        want, _ = orc.poly_multiply_crt(a, b, p64, threads=usable_cores()[0])
        if not np.array_equal(got, want):
            failed.append("crt oracle")
    t = timed(lambda: ex.multiply(ta, tb), 10)
    out["crt4_n2p17_ms"] = 1e3 * t
    out["crt4_GBps"] = (32 * n - 8) / t / 1e9                    # read a, b (u64), write ab (u64)
    out["crt4_sent_bytes_per_rank"] = (world - 1) * ex.sl * (4 * ex.count + 8)      # residue slices out + u64 slice all-gather
    out["crt4_primes_per_rank"] = -(-4 // world)

    # (b) RNS limbs over the ranks, no exchange
    n, limbs, polys = 65536, 32, 256
    primes = rns_primes(limbs, 17, 1 << 30)
    start, count = shard.partition(limbs, world, rank)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4 + rank)
    ra = torch.randint(0, 1 << 30, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    rb = torch.randint(0, 1 << 30, (polys, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
    rab_all = torch.empty((max(count, 1), polys, 2 * n - 1), dtype=torch.int32, device="cuda")    # the limbs run side by side
    rab = rab_all[max(count, 1) - 1]
    plans = [ctx.plan(primes[start + j], n, polys) for j in range(count)]

    def rns():                                            # every limb of this rank in ONE call (ntt_poly_multiply_limbs)
        if plans:
            ntt_amd.multiply_limbs(plans, [ra] * count, [rb] * count, [rab_all[j] for j in range(count)], polys)
    if plans:
        rns()
        sync()
        pl_p = primes[start + count - 1]                  # rab holds the last limb's products
        for x in (1, pl_p - 1, 0x2545F491):
            if not torch.equal(_eval_rows_t(torch, ra, x, pl_p) * _eval_rows_t(torch, rb, x, pl_p) % pl_p, _eval_rows_t(torch, rab, x, pl_p)):
                failed.append("rns identity")
                break
        if rank == 0:
            # exact coefficients at both ends of one row (Python integers).  The CPU oracle restates the reference, whose
            # generator is the literal 3 and whose Montgomery body pads to a power of FOUR (main-montgomery.cc:257,
            # :317-321): neither holds for primes that are only = 1 mod 2^17, so this workload has no oracle row
            r = polys - 1
            ai, bi, ci = ra[r].cpu().tolist(), rb[r].cpu().tolist(), rab[r].cpu().tolist()
            for k in list(range(12)) + list(range(2 * n - 1 - 12, 2 * n - 1)):
                lo, hi = max(0, k - (n - 1)), min(k, n - 1)
                if sum(ai[i] * bi[k - i] for i in range(lo, hi + 1)) % pl_p != ci[k] % (1 << 32):
                    failed.append("rns exact coefficients")
                    break
    t = timed(rns, 3)
    out["rns_32x256_n2p16_ms"] = 1e3 * t
    out["rns_products_per_s"] = limbs * polys / t
    out["rns_GBps"] = limbs * polys * (16 * n - 4) / t / 1e9
    out["rns_limbs_per_rank"] = -(-limbs // world)
    for pl in plans:
        pl.close()
    del ra, rb, rab, rab_all

    # (c) one transform of 2^26 points, forward + inverse
    lgL, p = 26, 469762049
    L = 1 << lgL
    w = pow(3, (p - 1) // L, p)
    if world == 1:
        plan = ctx.plan(p, L // 2, 1)
        gen.manual_seed(26)
        x = torch.randint(0, p, (L,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        y = x.clone()
        plan.forward(y, 1)
        sync()
        k = 0x2B3C5D7 % L
        if int(y[k].item()) != _eval_point_t(torch, x, pow(w, k, p), p):
            failed.append("ntt point")
        plan.inverse(y, 1)
        sync()
        if not torch.equal(y, x):
            failed.append("ntt round trip")
        t = timed(lambda: (plan.forward(y, 1), plan.inverse(y, 1)), 3)
        out["ntt2p26_fwd_inv_ms"] = 1e3 * t
        out["ntt2p26_GBps"] = 16 * L / t / 1e9
        out["ntt2p26_sent_bytes_per_rank"] = 0
        plan.close()
    elif world & (world - 1) == 0 and world <= 64:
        S = L // world
        dp = ntt_amd.DistPlan(ctx, p, lgL, world, rank)
        dx = shard.DistExchange(group, ctx, dp, on_device_exchange=on_device)
        gen.manual_seed(26 + rank)
        x = torch.randint(0, p, (S,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
        X = dx.forward(x).clone()                         # X[k2] = transform point rank + world * k2
        back = dx.inverse(X)
        sync()
        if not torch.equal(back, x):
            failed.append("ntt round trip")
        if rank == 0:                                     # one output point by direct evaluation over every rank's block
            blocks = []
            for r in range(world):
                gen.manual_seed(26 + r)
                blocks.append(torch.randint(0, p, (S,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32))
            k2 = 0x2B3C5D7 % S
            if int(X[k2].item()) != _eval_point_t(torch, torch.cat(blocks), pow(w, world * k2, p), p):
                failed.append("ntt point")
            del blocks
        t = timed(lambda: dx.inverse(dx.forward(x)), 3)
        out["ntt2p26_fwd_inv_ms"] = 1e3 * t
        out["ntt2p26_GBps"] = 16 * L / t / 1e9
        out["ntt2p26_sent_bytes_per_rank"] = 4 * (world - 1) * (S // world) * 4       # two all-to-alls each way
        dp.close()
    else:
        out["ntt2p26_skipped"] = "needs a power-of-two number of ranks"
    ctx.close()
    # (d) the same exchanges INSIDE the C++ library: one process over all the job's GPUs (ntt_ctx_create with ndev > 1,
    # multidev.cpp) -- peer copies for the CRT slices, peer loads / stores in the column kernel for the split transform.
    # Rank 0 runs it as a CHILD process while the other ranks wait: those paths have only ever run with repeated device
    # ids on one GPU, and a fault there must not take this line along (tools/mdev_bench.py)
    if world > 1:
        torch.cuda.synchronize()
        group.barrier()
        if rank == 0:
            ndev_seen = max(1, torch.cuda.device_count())
            devs = [r % ndev_seen for r in range(world)] if on_device else [local_rank % ndev_seen] * world
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mdev_bench.py"), "--devices",
                                    ",".join(str(d) for d in devs)], capture_output=True, text=True,
                                   timeout=max(20.0, args.also_timeout / 3))
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if lines:
                    md = json.loads(lines[-1])
                    out.update({k: v for k, v in md.items() if k.startswith("mdev_")})
                    out["mdev_devices"] = ",".join(str(d) for d in md["devices"])
                    if md.get("mdev_checks") != "ok":
                        failed.append("in-library multi-device path")
                else:
                    out["mdev_error"] = f"exit {r.returncode}: {r.stderr[-200:]}"
            except subprocess.TimeoutExpired:
                out["mdev_error"] = "timeout"
            # (e) and once more with ONE PROCESS PER GPU inside the library: ntt_comm (comm.cpp), the communicator a C++
            # host program shaped like main-mpi.cc would use -- RCCL joined by the library itself when the ranks own
            # GPUs, the host program's all-to-all on host memory otherwise.  A child JOB of `world` ranks, same reason
            try:
                port = int(os.environ.get("MASTER_PORT", "29511")) + 37
                r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                    "--master-addr", "127.0.0.1", "--master-port", str(port),
                                    os.path.join(ROOT, "tools", "comm_bench.py"), "--backend", "nccl" if on_device else "gloo"],
                                   capture_output=True, text=True, timeout=max(30.0, args.also_timeout / 3),
                                   env={k: v for k, v in os.environ.items() if k not in
                                        ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT",
                                         "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")})
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if lines:
                    cj = json.loads(lines[-1])
                    out.update({k: v for k, v in cj.items() if k.startswith("comm_")})
                    if cj.get("comm_checks") != "ok":
                        failed.append("in-library process-per-GPU communicator")
                else:
                    out["comm_error"] = f"exit {r.returncode}: {r.stderr[-200:]}"
            except subprocess.TimeoutExpired:
                out["comm_error"] = "timeout"
            group.host_signal("ntt_bench_mdev_done")
        else:
            group.host_wait("ntt_bench_mdev_done", args.also_timeout)     # on the host: this rank's GPU stays idle for the child
        group.barrier()
    nfailed = group.sum_over_ranks(float(len(failed)))
    out["checks"] = "ok" if nfailed == 0 else "FAILED on some rank: " + ", ".join(failed)
    return out


def kernel_source_stamp():
    """sha256 over the kernel sources: profiles/*.json carry the stamp of the build they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ntt_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # the device code: kernels and the headers they include (engine.hpp, rccl_dl.hpp are host-only: no kernel sees them)
        if f.endswith((".hip", ".hpp")) and f not in ("engine.hpp", "rccl_dl.hpp"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--p", type=int, default=998244353)
    ap.add_argument("--barrett", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="products per kernel launch (0 = library default)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the result checks (timing-only ablation builds whose results are wrong on purpose)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="rendezvous backend for N > 1; gloo lets ranks share a GPU to rehearse the multi-rank flow on a one-GPU box")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (no roofline block)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5,
                    help="length of the untimed-by-the-driver loop behind `roofline.sustained` (0 = skip)")
    ap.add_argument("--no-one-lane", action="store_true", help="skip the one-lane per-kernel table")
    ap.add_argument("--timing-stride", type=int, default=4,
                    help="bracket every N-th launch of the dominant kernel with events in the timed region (1: all)")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps of the same work for this long in front of the timed region (power / clock steady state)")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` block (CRT, RNS and 2^26 workloads behind the headline loop)")
    ap.add_argument("--also-timeout", type=float, default=150.0, help="watchdog of the `also` block, seconds")
    ap.add_argument("--with-ref", action="store_true",
                    help="cpu_baseline also times oracle/_ref (the reference's main-baseline.cc compiled in the build container); "
                         "off by default: the default run executes nothing built from /root/reference")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch, rendezvous, barrier and the max-over-ranks only: no GPU work, value = null "
                         "(exercises the N-rank start-up on a box without a GPU)")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`--gpus N` outside torch.distributed.run: start the N ranks as a child job (main-mpi.cc:642-665 brings up
    its own ranks the same way through mpiexec).  Nothing in this process has touched the GPU or imported torch."""
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench: --gpus {args.gpus} but torch.distributed.run started {world_env} ranks", file=sys.stderr)
        return 2

    import torch                      # device memory, streams and torch.distributed: plumbing
    from ntt_amd import shard   # host-side rank helpers (importing the package loads the C-ABI library; no GPU call)

    rank, world, local_rank = shard.env_rank_world()
    if args.dry_run:
        group = shard.Group("gloo")
        if group.world != args.gpus:
            return 2
        group.barrier()
        t0 = time.perf_counter()
        group.barrier()
        elapsed = group.max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "poly-mults/sec at n=2^17 (32-bit p)", "value": None, "unit": "poly-mults/s",
                              "n_gpus": group.world, "steps": 0, "warmup": 0, "ms_per_step": 1e3 * elapsed,
                              "dry_run": True}), flush=True)
        group.close()
        return 0

    import ntt_amd                    # the product: HIP kernels behind the C ABI

    if args.backend == "gloo":                        # rehearsal: ranks may share a GPU, exchanges on host tensors
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    group = shard.Group(args.backend, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)   # RCCL when world > 1
    if group.world != args.gpus:
        print(f"bench: rank {rank}: world {group.world} != --gpus {args.gpus}", file=sys.stderr)
        return 2

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

    # every rank checks products of its warm-up output before timing anything (no oracle here: the oracle
    # comparison of pair 0 happens inside the cpu_baseline leg below): the first, the last and one in between
    gpu_row0 = None
    if not args.no_check:
        bad = 0.0
        for row in sorted({0, B // 2, B - 1}):
            got = dab[row, :2 * n - 1].cpu().numpy().view(np.uint32)
            if not evaluation_identity(ha[row], hb[row], got, p):
                print(f"bench: rank {rank}: A(x) * B(x) != AB(x) on the GPU result, row {row}", file=sys.stderr)
                bad = 1.0
        if group.max_over_ranks(bad) > 0:            # every rank learns of a failure on any rank: nobody waits at a barrier
            group.close()
            return 1
        gpu_row0 = dab[0, :2 * n - 1].cpu().numpy().view(np.uint32).copy()
        if world > 1 and rank == 0:
            # one oracle row with N > 1 as well (with one rank the cpu_baseline leg below does it): checker only
            import orc
            want, _ = orc.poly_multiply_omp(ha[0].astype(np.uint64), hb[0].astype(np.uint64), p, threads=usable_cores()[0])
            if not np.array_equal(np.asarray(want).astype(np.uint64), gpu_row0.astype(np.uint64)):
                print("bench: rank 0: GPU result differs from the oracle", file=sys.stderr)
                bad = 1.0
        if group.max_over_ranks(bad) > 0:
            group.close()
            return 1

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
        ctx.kernel_timing_stride(args.timing_stride)   # a sample of the timed region's launches (every 4th of 22 per step)
    # settle: the board's power controller takes ~0.5 s of load to reach its steady clock (the loop
    # over >= 2 s behind the timed region used to read 4-6 % above a timed region that started cold).  Untimed steps of
    # the same work until --settle-seconds have passed, so that the K timed steps below see the steady state
    settle_steps = 0
    if args.settle_seconds > 0:
        ctx.enable_kernel_timing(False)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.settle_seconds:
            for _ in range(10):
                step()
            sync()
            settle_steps += 10
        settle_steps = int(group.max_over_ranks(settle_steps))
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

    # the same loop over >= 2 s (the board's power controller settles in ~0.5 s; the timed region above may be shorter)
    sustained = None
    if args.sustained_seconds > 0 and not args.no_kernel_timing:
        reps = max(args.steps, int(args.sustained_seconds / max(elapsed / args.steps, 1e-6)) + 1)
        barrier()
        sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            step()
        sync()
        barrier()
        dt = group.max_over_ranks(time.perf_counter() - t1)
        sustained = {"value": world * B * reps / dt, "seconds": dt, "steps": reps, "ms_per_step": 1e3 * dt / reps}

    # uncontended per-kernel times: a second context with ONE stream lane, every kernel bracketed (untimed)
    one_lane = None
    if rank == 0 and not args.no_kernel_timing and not args.no_one_lane:
        prev = os.environ.get("NTT_AMD_STREAMS")
        os.environ["NTT_AMD_STREAMS"] = "1"
        ctx1 = ntt_amd.Context(local_rank)
        plan1 = ctx1.plan(p, n, B, kind)
        if prev is None:
            del os.environ["NTT_AMD_STREAMS"]
        else:
            os.environ["NTT_AMD_STREAMS"] = prev
        plan1.multiply(da, db, out=dab, batch=B, out_stride=stride)
        ctx1.synchronize()
        ctx1.enable_kernel_timing(True)
        for _ in range(3):
            plan1.multiply(da, db, out=dab, batch=B, out_stride=stride)
        ctx1.synchronize()
        one_lane = {k: {"us_per_product": 1e3 * v[0] / (3 * B), "avg_launch_us": 1e3 * v[0] / v[1]}
                    for k, v in ctx1.kernel_times().items() if v[1] > 0}
        ctx1.enable_kernel_timing(False)
        plan1.close()
        ctx1.close()

    line = None
    if rank == 0 and args.no_kernel_timing:         # sweeps (tools/): the rate only
        print(json.dumps({"value": world * B * args.steps / elapsed, "n_gpus": world,
                          "ms_per_step": 1e3 * elapsed / args.steps, "chunk": args.chunk}), flush=True)
    elif rank == 0:
        units = world * B * args.steps
        value = units / elapsed
        alg_bytes = 16 * n - 4                      # read a, read b, write ab
        # dominant kernel = largest share of device time in the timed region
        name, (tot_ms, launches) = max(((k, v) for k, v in ktimes.items() if v[1] > 0),
                                       key=lambda kv: kv[1][0])
        # `launches` = the launches that carried events (every --timing-stride-th); the launches per step come from the
        # two untimed steps in which every launch was bracketed
        launches_per_step = warm_ktimes[name][1] / 2 if name in warm_ktimes else launches / args.steps
        units_per_launch = B / launches_per_step
        avg_ms = tot_ms / launches
        achieved = alg_bytes * units_per_launch / (avg_ms * 1e-3) / 1e9
        stamp = kernel_source_stamp()
        traffic, traffic_info = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            per_unit = tj.get(name, {}).get("bytes_per_unit")
            if per_unit:                 # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE) per launch, bytes
                traffic = per_unit * units_per_launch
                traffic_info = {"bytes_per_unit": per_unit, "measured_on_build": tj.get("_build"),
                                "this_build": stamp, "same_build": tj.get("_build") == stamp,
                                "kernel_own_GBps": per_unit * units_per_launch / (avg_ms * 1e-3) / 1e9}
        # integer-VALU roofline (the bound DESIGN.md section 5 argues for): issue cycles per product from the
        # instruction mix of the three kernels (profiles/valu_model.json, tools/valu_model.py) against the
        # chip's SIMD cycles per second
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu_model.json")
        if os.path.exists(vpath) and n == 131072 and not args.barrett:
            vj = json.load(open(vpath))
            cyc = sum(k["issue_cycles_per_product"] for k in vj["kernels"].values())
            lane_ops = sum(k["lane_ops_per_product"] for k in vj["kernels"].values())
            floor_us = 1e6 * cyc / SIMD_CYCLES_PER_S
            valu = {"bound": "valu-int", "lane_ops_per_product": lane_ops, "issue_cycles_per_product": cyc,
                    "peak": lane_ops / (cyc / SIMD_CYCLES_PER_S) / 1e12, "achieved": lane_ops * value / world / 1e12,
                    "unit": "T lane-op/s", "floor_us_per_product": floor_us,
                    "frac": floor_us / (1e6 * world / value),
                    "per_kernel_issue_cycles": {k: v["issue_cycles_per_product"] for k, v in vj["kernels"].items()},
                    "measured_mix_peak": MEASURED_MIX_PEAK_TLANEOPS,
                    "frac_of_measured_mix_peak": lane_ops * value / world / 1e12 / MEASURED_MIX_PEAK_TLANEOPS,
                    "floor_us_at_measured_mix_peak": lane_ops / MEASURED_MIX_PEAK_TLANEOPS / 1e6,
                    "traffic_floor_us_per_product": MEASURED_TRAFFIC_FLOOR_US,
                    "measured_on_build": vj.get("_build"), "same_build": vj.get("_build") == stamp,
                    "note": "peak = this instruction mix at 2 / 4 issue cycles per wave64 instruction on 256 CUs x 4 SIMDs at 2.4 GHz"}
        out = {
            "metric": f"poly-mults/sec at n=2^{n.bit_length() - 1} (32-bit p)" if n & (n - 1) == 0
                      else f"poly-mults/sec at n={n} (32-bit p)",
            "value": value,
            "unit": "poly-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps_untimed": settle_steps,
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
                         "kernel": name, "avg_launch_ms": avg_ms, "launches": launches, "launches_timed_every": args.timing_stride,
                         "units_per_launch": units_per_launch, "algorithmic_bytes_per_unit": alg_bytes,
                         "traffic_detail": traffic_info,
                         "pipeline_achieved": value / world * alg_bytes / 1e9,
                         "pipeline_frac": value / world * alg_bytes / 1e9 / HBM_PEAK_GBPS,
                         # what actually binds: integer VALU issue.  Scalars, so that they
                         # survive next to `frac`: the pipeline's VALU lane-op rate against the rate the butterfly's
                         # instruction mix reaches beside this much memory traffic (COUPLED_MIX_PEAK_TLANEOPS,
                         # profiles/ubench_clock_r02.txt), against the same mix alone, and against the issue-cycle model
                         "binding": "valu-int" if valu else "unknown (no VALU model for this shape)",
                         "valu_frac_of_coupled_peak": valu and valu["achieved"] / COUPLED_MIX_PEAK_TLANEOPS,
                         "valu_frac_of_mix_peak": valu and valu["frac_of_measured_mix_peak"],
                         "valu_frac_of_issue_model": valu and valu["frac"],
                         "valu_coupled_peak_TlaneOps": COUPLED_MIX_PEAK_TLANEOPS,
                         "valu_achieved_TlaneOps": valu and valu["achieved"],
                         "sustained_value": sustained and sustained["value"],
                         "one_lane_col_fwd_us": one_lane and one_lane.get("col_fwd", {}).get("us_per_product"),
                         "one_lane_row_mul_us": one_lane and one_lane.get("row_mul", {}).get("us_per_product"),
                         "one_lane_col_inv_us": one_lane and one_lane.get("col_inv", {}).get("us_per_product"),
                         # the dominant kernel alone on one stream lane (uncontended): `frac` above is computed from
                         # its duration in the timed region, where the other lane's row kernel runs beside it
                         "frac_one_lane": one_lane and name in one_lane and
                                          alg_bytes / (one_lane[name]["us_per_product"] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                         "kernel_ms_per_step_untimed": {k: v[0] / 2 for k, v in warm_ktimes.items()},
                         "kernels_one_lane": one_lane, "valu": valu, "sustained": sustained},
        }
        line = out

    # the other sharded workloads, behind the headline loop and after the headline line is complete: a watchdog keeps a
    # hung exchange from taking the line along -- it prints the line with whatever part of the block had finished
    also = None
    if not args.no_also and not args.no_kernel_timing:
        import threading
        partial = {}

        def give_up():
            if rank == 0:
                line["also"] = dict(partial, error=f"the also block did not finish within {args.also_timeout} s; entries so far")
                print(json.dumps(line), flush=True)
            os._exit(0)
        dog = threading.Timer(args.also_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            also = also_block(args, group, rank, world, local_rank, partial)
        except Exception as e:                       # the headline measurement stands; say what happened
            also = dict(partial, error=f"{type(e).__name__}: {e}"[:300])
        dog.cancel()

    rc = 0
    if also is not None and (str(also.get("checks", "ok")) != "ok"):
        rc = 1
    if line is not None:
        if also is not None:
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, p, args.cpu_seconds, gpu_row0, args.with_ref)
        print(json.dumps(line), flush=True)

    barrier()                       # rank 0 ran the one-lane table meanwhile: every rank leaves the group together
    plan.close()
    ctx.close()
    group.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
