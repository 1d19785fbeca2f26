#!/usr/bin/env python3
"""Per-kernel time of ONE n = 2^17 product per call (the harness's use), device-resident."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ntt_amd  # noqa: E402

n, p = 131072, 998244353
ctx = ntt_amd.Context(0)
plan = ctx.plan(p, n, 1)
rng = np.random.default_rng(0)
a = torch.from_numpy(rng.integers(0, p, size=(1, n), dtype=np.int64).astype(np.int32)).cuda()
b = torch.from_numpy(rng.integers(0, p, size=(1, n), dtype=np.int64).astype(np.int32)).cuda()
out = torch.zeros((1, 2 * n), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for timing in (False, True):
    ctx.enable_kernel_timing(timing)
    for _ in range(5):
        plan.multiply(a, b, out=out, batch=1, out_stride=2 * n)
    ctx.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.multiply(a, b, out=out, batch=1, out_stride=2 * n)
        ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.multiply(a, b, out=out, batch=1, out_stride=2 * n)
    ctx.synchronize()
    dt2 = (time.perf_counter() - t0) / reps
    print("timing %s: %.1f us per product synchronised each call, %.1f us back to back" % (timing, dt * 1e6, dt2 * 1e6))
    if timing:
        for k, (ms, cnt) in ctx.kernel_times().items():
            if cnt:
                print("   %-10s %6d launches  %.1f us avg" % (k, cnt, ms * 1e3 / cnt))
