#!/usr/bin/env python3
"""Per-kernel time of ntt_forward + ntt_inverse (natural order) at L = 2^26, p = 469762049 (config 5)."""
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

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
p, L = 469762049, 1 << lg
ctx = ntt_amd.Context(0)
x0 = (orc.splitmix64_np(7, L) % np.uint64(p)).astype(np.uint32)
x = torch.from_numpy(x0.view(np.int32)).cuda()
plan = ctx.plan(p, L // 2, 1, ntt_amd.MONTGOMERY)
for timing in (False, True):
    ctx.enable_kernel_timing(timing)
    plan.forward(x, 1)
    plan.inverse(x, 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        plan.forward(x, 1)
        plan.inverse(x, 1)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("L=2^%d timing %s: %.3f ms per forward+inverse  (%.0f GB/s algorithmic)" % (lg, timing, dt * 1e3, 16 * L / dt / 1e9))
    if timing:
        for k, (ms, cnt) in ctx.kernel_times().items():
            if cnt:
                print("   %-10s %6d launches  %.1f us avg  %.2f ms total per pair" % (k, cnt, ms * 1e3 / cnt, ms / (reps + 1)))
assert np.array_equal(x.cpu().numpy().view(np.uint32), x0)
print("round trip ok")
