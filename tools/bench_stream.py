#!/usr/bin/env python3
"""PCIe-inclusive rate of ntt_poly_multiply_stream at the headline shape (n = 2^17, p = 998244353):
1024 products from pinned and from pageable host arrays, slices of --slice products."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slice", type=int, default=64)
    ap.add_argument("--count", type=int, default=1024)
    ap.add_argument("--stride2n", action="store_true", help="output row stride 2n instead of 2n-1")
    args = ap.parse_args()
    import torch  # noqa: F401  (one HSA runtime for the process)
    import ntt_amd
    import orc
    n, p, B = 131072, 998244353, args.count
    ctx = ntt_amd.Context(0)
    plan = ctx.plan(p, n, args.slice)
    stride = 2 * n if args.stride2n else 2 * n - 1
    pa, pb = ntt_amd.host_array((B, n)), ntt_amd.host_array((B, n))
    pab = ntt_amd.host_array((B, stride))
    pa[:] = (orc.splitmix64_np(1, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    pb[:] = (orc.splitmix64_np(2, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    pab[:] = 0
    qa, qb = np.array(pa), np.array(pb)
    qab = np.zeros((B, stride), dtype=np.uint32)
    for name, (xa, xb, xo) in (("pinned", (pa, pb, pab)), ("pageable", (qa, qb, qab))):
        plan.multiply_stream(xa[:2 * args.slice], xb[:2 * args.slice], out=xo[:2 * args.slice], out_stride=stride)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            plan.multiply_stream(xa, xb, out=xo, out_stride=stride)
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"host_memory": name, "slice": args.slice, "count": B, "out_stride": stride,
                          "poly_mults_per_s": B / best, "GBps_each_way": B * 8 * n / best / 1e9}), flush=True)
    assert np.array_equal(pab, qab)
    want = orc.poly_multiply(pa[B - 1], pb[B - 1], p)
    assert np.array_equal(pab[B - 1, :2 * n - 1], np.asarray(want).astype(np.uint32))
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
