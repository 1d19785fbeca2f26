#!/usr/bin/env python3
"""tools/mdev_bench.py --devices 0,1,2,3 -- the in-library multi-GPU path (ntt_ctx_create with ndev > 1,
ntt_amd/csrc/multidev.cpp) timed in ONE process that drives every listed GPU; prints one JSON line.

bench.py --gpus N runs this as a CHILD process of rank 0 at the end of its `also` block (the other ranks wait): the
library's own exchanges -- peer copies for the CRT slices, peer loads / stores inside the column kernel for the split
transform -- have only ever run with repeated device ids on one GPU, so a fault here must not take the headline line
along.  Workloads: (2) the 4-prime CRT product at n = 2^17, prime per device + slice exchange
(main-mpi.cc:657-660, 739-797); (3) one 2^26-point transform forward + inverse, block r on device r, transports p2p and
copy, and the split product of two polynomials of 2^25 coefficients (ntt_dist_multiply).  Checks are oracle-free: C(1), C(-1) of the CRT product with Python integers; the transform's round trip and one
output point evaluated directly."""
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
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--lgL", type=int, default=26)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="products per slot of the batch-per-device workload")
    args = ap.parse_args()
    devs = [int(x) for x in args.devices.split(",")]
    import torch
    import ntt_amd
    import bench

    out = {"devices": devs, "distinct_devices": len(set(devs))}
    failed = []
    grp = ntt_amd.Context(devices=devs)
    nd = len(devs)

    # (1) the headline workload, batch per device (BASELINE.json configs[1]): every slot multiplies B products of n = 2^17
    # held in its own HBM; one process, one host thread per slot, no exchange
    n, p, B = 131072, 998244353, args.batch
    plan = grp.plan(p, n, B * nd)
    da, db, dab = [], [], []
    for r, d in enumerate(devs):
        with torch.cuda.device(d):
            gen = torch.Generator(device="cuda")
            gen.manual_seed(1 + r)
            da.append(torch.randint(0, p, (B, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32))
            db.append(torch.randint(0, p, (B, n), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32))
            dab.append(torch.zeros((B, 2 * n), dtype=torch.int32, device="cuda"))
            torch.cuda.synchronize()
    counts = [B] * nd
    for _ in range(3):
        plan.multiply_slots(da, db, dab, counts, out_stride=2 * n)
    grp.synchronize()
    for r, d in enumerate(devs):                                # A(x) B(x) = AB(x) on the last row of every slot
        with torch.cuda.device(d):
            for x in (1, 0x2545F491 % p):
                if bench._eval_point_t(torch, da[r][B - 1].contiguous(), x, p, blk=1 << 12) * bench._eval_point_t(torch, db[r][B - 1].contiguous(), x, p, blk=1 << 12) % p != \
                        bench._eval_point_t(torch, torch.cat([dab[r][B - 1, :2 * n - 1], torch.zeros(1, dtype=torch.int32, device="cuda")]), x, p, blk=1 << 12):
                    failed.append(f"batch identity (slot {r})")
                    break
    t0 = time.perf_counter()
    for _ in range(args.reps * 2):
        plan.multiply_slots(da, db, dab, counts, out_stride=2 * n)
    grp.synchronize()
    dt = (time.perf_counter() - t0) / (args.reps * 2)
    out["mdev_batch_products_per_s"] = nd * B / dt
    out["mdev_batch_per_slot"] = B
    plan.close()
    del da, db, dab

    # (2) large-modulus product, prime per device
    n, p64 = 131072, 1337006139375617
    a = bench.splitmix64(0xC47A, n) % np.uint64(p64)
    b = bench.splitmix64(0xC47B, n) % np.uint64(p64)
    got = grp.crt_multiply(a, b, p64)

    def at(c, sign):
        v = c.tolist()
        return (sum(v[0::2]) + sign * sum(v[1::2])) % p64
    if int(got.max()) >= p64 or any(at(a, s) * at(b, s) % p64 != at(got, s) for s in (1, -1)):
        failed.append("crt identity")
    with torch.cuda.device(devs[0]):
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dout = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
    grp.crt_multiply(da, db, p64, out=dout)                      # operands on device 0: the peers read them over the links
    if not np.array_equal(dout.cpu().numpy().view(np.uint64), got):
        failed.append("crt device arrays")
    t0 = time.perf_counter()
    for _ in range(args.reps * 4):
        grp.crt_multiply(da, db, p64, out=dout)                  # complete on return
    out["mdev_crt4_n2p17_ms"] = 1e3 * (time.perf_counter() - t0) / (args.reps * 4)

    # (3) one transform split over the devices, exchanges inside the library
    if nd & (nd - 1) == 0 and 2 <= nd <= 16:
        p, lgL = 469762049, args.lgL
        L, S = 1 << lgL, (1 << lgL) // nd
        w = pow(3, (p - 1) // L, p)
        xin, xout, back = [], [], []
        for r, d in enumerate(devs):
            with torch.cuda.device(d):
                gen = torch.Generator(device="cuda")
                gen.manual_seed(26 + r)
                xin.append(torch.randint(0, p, (S,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32))
                xout.append(torch.zeros(S, dtype=torch.int32, device="cuda"))
                back.append(torch.zeros(S, dtype=torch.int32, device="cuda"))
                torch.cuda.synchronize()
        for transport in ("p2p", "copy"):
            os.environ["NTT_AMD_DIST_TRANSPORT"] = transport
            try:
                dp = ntt_amd.DistPlan(grp, p, lgL, nd, 0)
            except ntt_amd.NttError as e:                         # p2p without peer access between the devices
                out[f"mdev_ntt_{transport}_skipped"] = str(e)[:120]
                continue
            dp.forward(xin, xout)
            dp.inverse(xout, back)
            grp.synchronize()
            if not all(torch.equal(back[r], xin[r]) for r in range(nd)):
                failed.append(f"ntt round trip ({transport})")
            k2 = 0x2B3C5D7 % S                                     # X[0 + nd * k2] lives on slot 0
            with torch.cuda.device(devs[0]):
                full = torch.cat([t.to(f"cuda:{devs[0]}") for t in xin])
                if int(xout[0][k2].item()) != bench._eval_point_t(torch, full, pow(w, nd * k2, p), p):
                    failed.append(f"ntt point ({transport})")
                del full
            t0 = time.perf_counter()
            for _ in range(args.reps):
                dp.forward(xin, xout)
                dp.inverse(xout, back)
            grp.synchronize()
            dt = (time.perf_counter() - t0) / args.reps
            out[f"mdev_ntt2p{lgL}_{transport}_fwd_inv_ms"] = 1e3 * dt
            out[f"mdev_ntt2p{lgL}_{transport}_GBps"] = 16 * L / dt / 1e9
            dp.close()
        os.environ.pop("NTT_AMD_DIST_TRANSPORT", None)
        # the split PRODUCT of two polynomials of L / 2 coefficients (default transport): forward x 2, pointwise, inverse
        dp = ntt_amd.DistPlan(grp, p, lgL, nd, 0)
        half = nd // 2
        dp.multiply(xin[:half], [xin[(r + 1) % half] for r in range(half)], xout)
        grp.synchronize()
        with torch.cuda.device(devs[0]):
            d0 = f"cuda:{devs[0]}"
            fa = torch.cat([xin[r].to(d0) for r in range(half)])
            fb = torch.cat([xin[(r + 1) % half].to(d0) for r in range(half)])
            fab = torch.cat([t.to(d0) for t in xout])
            for x in (1, p - 1, 0x2545F491 % p):
                if bench._eval_point_t(torch, fa, x, p) * bench._eval_point_t(torch, fb, x, p) % p != bench._eval_point_t(torch, fab, x, p):
                    failed.append("split product identity")
                    break
            del fa, fb, fab
        t0 = time.perf_counter()
        for _ in range(args.reps):
            dp.multiply(xin[:half], [xin[(r + 1) % half] for r in range(half)], xout)
        grp.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        out[f"mdev_product_n2p{lgL - 1}_ms"] = 1e3 * dt
        out[f"mdev_product_n2p{lgL - 1}_GBps"] = (16 * (L // 2) - 4) / dt / 1e9       # read a, b, write ab
        dp.close()
    grp.close()
    out["mdev_checks"] = "ok" if not failed else "FAILED: " + ", ".join(failed)
    print(json.dumps(out), flush=True)
    return 0 if not failed else 1


if __name__ == "__main__":
    sys.exit(main())
