#!/usr/bin/env python3
"""Direct 64-bit path at n = 2^17: stream lanes (NTT_AMD_LANES64) x products per launch (NTT_AMD_CHUNK64), both read per
call, interleaved in one process; us per product, best of 3."""
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

n = 131072
ctx = ntt_amd.Context(0)
for p64 in (1337006139375617, 4611686018326724609):       # 51 bits: FP64 lanes; 62 bits: Montgomery lanes
    for B in (64, 256):
        plan = ctx.plan64(p64, n, batch=B)
        xa = torch.from_numpy((orc.splitmix64_np(5, B * n) % np.uint64(p64)).view(np.int64).reshape(B, n)).cuda()
        xb = torch.from_numpy((orc.splitmix64_np(6, B * n) % np.uint64(p64)).view(np.int64).reshape(B, n)).cuda()
        xo = torch.zeros((B, 2 * n), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        cfgs = [(1, 48), (1, 32), (2, 8), (2, 12), (2, 16), (2, 24), (2, 32), (2, 48)]
        best = {c: 1e9 for c in cfgs}
        ref = None
        for _ in range(3):
            for lanes, chunk in cfgs:
                os.environ["NTT_AMD_LANES64"], os.environ["NTT_AMD_CHUNK64"] = str(lanes), str(chunk)
                plan.multiply(xa, xb, out=xo, batch=B, out_stride=2 * n)
                ctx.synchronize()
                if ref is None:
                    ref = xo.clone()
                elif not torch.equal(xo, ref):
                    print("MISMATCH", lanes, chunk)
                reps = max(4, 1024 // B)
                t0 = time.perf_counter()
                for _ in range(reps):
                    plan.multiply(xa, xb, out=xo, batch=B, out_stride=2 * n)
                ctx.synchronize()
                best[(lanes, chunk)] = min(best[(lanes, chunk)], (time.perf_counter() - t0) / reps / B * 1e6)
        print(f"p64 = {p64} ({p64.bit_length()} bits), batch {B}: " + ", ".join(f"{l} lane{'s' if l > 1 else ''} x {c}: {best[(l, c)]:.2f}" for l, c in cfgs))
        plan.close()
        del xa, xb, xo, ref
