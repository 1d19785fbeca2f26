#!/usr/bin/env python3
"""Throughput across sizes: linear products (L = 2n) and negacyclic products (L = n), p = 998244353 /
469762049, device-resident, batch sized to ~64 Mi coefficients per operand."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ntt_amd  # noqa: E402

ctx = ntt_amd.Context(0)
p = 469762049
rng = np.random.default_rng(0)
for lgn in range(8, 23):
    n = 1 << lgn
    B = max(1, min(4096, (1 << 25) // n))
    a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    res = {"lgn": lgn, "batch": B}
    for name in ("linear", "negacyclic"):
        plan = ctx.plan(p, n, B) if name == "linear" else ctx.negacyclic_plan(p, n, B)
        stride = 2 * n if name == "linear" else n
        out = torch.zeros((B, stride), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps / B
        L = plan.length
        lgL = L.bit_length() - 1
        bf = 3 * (L // 2) * lgL
        res[name] = {"us_per_product": round(dt * 1e6, 3), "products_per_s": round(1 / dt), "ps_per_butterfly": round(dt * 1e12 / bf, 1)}
        plan.close()
        del out
    print(json.dumps(res), flush=True)
ctx.close()
