#!/usr/bin/env python3
"""us per product at the sizes around the one-kernel / three-kernel boundary (device-resident batches, default
lanes), with the one-kernel products (seq_product_kernel, kernels_fused.hip) off and on: NTT_AMD_FUSED = 0 (three
kernels for every L above 4096), 1 (default: one kernel for L = 8192 and the linear L = 16384), 2 (also for the
full-length L = 16384).  The three builds are checked against each other on the same inputs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

ctx = ntt_amd.Context(0)
p = 469762049
rng = np.random.default_rng(0)
CASES = (("linear", 11), ("linear", 12), ("linear", 13), ("linear", 14), ("negacyclic", 12), ("negacyclic", 13),
         ("negacyclic", 14), ("negacyclic", 15))
for name, lgn in CASES:
    n = 1 << lgn
    B = (1 << 25) // n
    a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    stride = 2 * n if name == "linear" else n
    out = torch.zeros((B, stride), dtype=torch.int32, device="cuda")
    res = {}
    ref = None
    for mode in ("0", "1", "2"):
        os.environ["NTT_AMD_FUSED"] = mode
        plan = ctx.plan(p, n, B) if name == "linear" else ctx.negacyclic_plan(p, n, B)
        torch.cuda.synchronize()
        for _ in range(3):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        res[mode] = round((time.perf_counter() - t0) / 10 / B * 1e6, 4)
        o = out.cpu().numpy().copy()
        if ref is None:
            ref = o
        elif not np.array_equal(ref, o):
            res[mode] = "MISMATCH"
        plan.close()
    print("n = 2^%d %-10s L = %6d batch %5d: us per product by NTT_AMD_FUSED %s" % (lgn, name, 2 * n if name == "linear" else n, B, res), flush=True)
ctx.close()
