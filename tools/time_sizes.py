#!/usr/bin/env python3
"""Per-kernel time split of the three-kernel path at the sizes just above the row length (one stream lane, every
kernel bracketed by events): where the steps of the size sweep come from."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

os.environ["NTT_AMD_STREAMS"] = "1"
ctx = ntt_amd.Context(0)
p = 469762049
rng = np.random.default_rng(0)
CASES = (("linear", 13), ("negacyclic", 13), ("linear", 14), ("negacyclic", 14), ("linear", 15), ("negacyclic", 15), ("linear", 16),
         ("negacyclic", 16), ("negacyclic", 17), ("linear", 17))
if len(sys.argv) > 1:
    CASES = tuple((a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:])
for name, lgn in CASES:
    n = 1 << lgn
    B = (1 << 25) // n
    a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    plan = ctx.plan(p, n, B) if name == "linear" else ctx.negacyclic_plan(p, n, B)
    stride = 2 * n if name == "linear" else n
    out = torch.zeros((B, stride), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        plan.multiply(a, b, out=out, batch=B, out_stride=stride)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plan.multiply(a, b, out=out, batch=B, out_stride=stride)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / 5 / B * 1e6
    ctx.enable_kernel_timing(True)
    for _ in range(3):
        plan.multiply(a, b, out=out, batch=B, out_stride=stride)
    ctx.synchronize()
    kt = {k: round(1e3 * v[0] / (3 * B), 4) for k, v in ctx.kernel_times().items() if v[1]}
    ctx.enable_kernel_timing(False)
    print("n = 2^%d %-10s L = %6d batch %5d: %.4f us per product; kernels (us per product): %s" % (lgn, name, plan.length, B, wall, kt), flush=True)
    plan.close()
ctx.close()
