#!/usr/bin/env python3
"""Per-kernel time of the direct 64-bit path (ntt_plan64) at n = 2^17: batch 64 and single product."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ntt_amd  # noqa: E402
import orc  # noqa: E402

p64, n, B = int(os.environ.get("P64", "1337006139375617")), 131072, 64
ctx = ntt_amd.Context(0)
plan = ctx.plan64(p64, n, batch=B)
xa = torch.from_numpy((orc.splitmix64_np(5, B * n) % np.uint64(p64)).view(np.int64).reshape(B, n)).cuda()
xb = torch.from_numpy((orc.splitmix64_np(6, B * n) % np.uint64(p64)).view(np.int64).reshape(B, n)).cuda()
xo = torch.zeros((B, 2 * n), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for batch in (B, 1):
    for timing in (False, True):
        ctx.enable_kernel_timing(timing)
        plan.multiply(xa, xb, out=xo, batch=batch, out_stride=2 * n)
        ctx.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            plan.multiply(xa, xb, out=xo, batch=batch, out_stride=2 * n)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("batch %d timing %s: %.1f us per call, %.2f us per product" % (batch, timing, dt * 1e6, dt * 1e6 / batch))
        if timing:
            for k, (ms, cnt) in ctx.kernel_times().items():
                if cnt:
                    print("   %-10s %6d launches  %.1f us avg" % (k, cnt, ms * 1e3 / cnt))
want, _ = orc.poly_multiply_crt(xa[0].cpu().numpy().view(np.uint64), xb[0].cpu().numpy().view(np.uint64), p64, threads=8)
assert np.array_equal(xo[0, :2 * n - 1].cpu().numpy().view(np.uint64), want)
print("ok")
