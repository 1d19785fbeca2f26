#!/usr/bin/env python3
"""64-bit products around L = 4096 (FHE-sized rings with ~50-bit limbs): time per product, batch 2048."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

p = int(os.environ.get("P64", "1337006139375617"))
ctx = ntt_amd.Context(0)
rng = np.random.default_rng(0)
B = 2048
for lgn in (10, 11, 12, 13):
    n = 1 << lgn
    a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
    b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
    for name in ("linear", "negacyclic"):
        plan = ctx.plan64(p, n, B) if name == "linear" else ctx.negacyclic_plan64(p, n, B)
        stride = 2 * n if name == "linear" else n
        out = torch.zeros((B, stride), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5 / B
        print("lgn %d %-10s L=%6d  %.4f us per product" % (lgn, name, plan.length, dt * 1e6), flush=True)
        plan.close()
