#!/usr/bin/env python3
"""Batched natural-order transforms (ntt_forward / ntt_inverse): one stream lane against two (NTT_AMD_TRANSFORM_LANES,
read per call), interleaved in one process; us per transform, best of 3."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

ctx = ntt_amd.Context(0)
for lgL, p, B in ((18, 998244353, 512), (17, 998244353, 1024), (16, 469762049, 2048), (20, 998244353, 128), (18, 1224736769, 512)):
    L = 1 << lgL
    plan = ctx.plan(p, L // 2, B)
    x = torch.randint(0, p, (B, L), device="cuda", dtype=torch.int64).to(torch.int32)
    keep = x.clone()
    torch.cuda.synchronize()
    best = {1: 1e9, 2: 1e9}
    spectra = {}
    for _ in range(3):
        for lanes in (1, 2):
            os.environ["NTT_AMD_TRANSFORM_LANES"] = str(lanes)
            x.copy_(keep)
            plan.forward(x, B)
            ctx.synchronize()
            spectra.setdefault(lanes, x.clone())
            plan.inverse(x, B)
            ctx.synchronize()
            if not torch.equal(x, keep):
                print("ROUND TRIP MISMATCH", lgL, lanes)
            reps = 4
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.forward(x, B)
                plan.inverse(x, B)
            ctx.synchronize()
            best[lanes] = min(best[lanes], (time.perf_counter() - t0) / reps / (2 * B) * 1e6)
    same = torch.equal(spectra[1], spectra[2])
    print(f"L = 2^{lgL}, p = {p}, batch {B}: one lane {best[1]:.3f} us per transform, two lanes {best[2]:.3f}  (spectra identical: {same})")
    plan.close()
    del x, keep, spectra
