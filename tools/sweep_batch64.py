#!/usr/bin/env python3
"""Direct 64-bit path at n = 2^17: time per product against the batch size of one call (how much of the
intermediates the 256 MiB Infinity Cache holds: 4 MiB per product)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

p64, n = 1337006139375617, 131072
ctx = ntt_amd.Context(0)
rng = np.random.default_rng(1)
Bmax = 128
xa = torch.from_numpy(rng.integers(0, p64, size=(Bmax, n), dtype=np.uint64).view(np.int64)).cuda()
xb = torch.from_numpy(rng.integers(0, p64, size=(Bmax, n), dtype=np.uint64).view(np.int64)).cuda()
xo = torch.zeros((Bmax, 2 * n), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for B in (1, 2, 4, 8, 16, 24, 32, 48, 64, 128):
    plan = ctx.plan64(p64, n, batch=B)
    calls = Bmax // B
    for _ in range(2):
        for c in range(calls):
            plan.multiply(xa[c * B:], xb[c * B:], out=xo[c * B:], batch=B, out_stride=2 * n)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        for c in range(calls):
            plan.multiply(xa[c * B:], xb[c * B:], out=xo[c * B:], batch=B, out_stride=2 * n)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps / (calls * B)
    print("batch %3d per call: %.2f us per product" % (B, dt * 1e6), flush=True)
    plan.close()
