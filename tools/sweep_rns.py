#!/usr/bin/env python3
"""RNS shape (BASELINE.json configs[3]: negacyclic n = 2^16, 32 limbs x 256 polynomials = 8192 products, device
resident): us per product over stream lanes x products per launch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

p, n, B = 998244353, 65536, 8192
rng = np.random.default_rng(0)
a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
out = torch.zeros((B, n), dtype=torch.int32, device="cuda")
for lanes in (2, 3, 4):
    for chunk in (0, 32, 64, 96, 128, 256, 384):
        os.environ["NTT_AMD_STREAMS"] = str(lanes)
        if chunk:
            os.environ["NTT_AMD_CHUNK"] = str(chunk)
        else:
            os.environ.pop("NTT_AMD_CHUNK", None)
        ctx = ntt_amd.Context(0)
        plan = ctx.negacyclic_plan(p, n, B)
        for _ in range(2):
            plan.multiply(a, b, out=out, batch=B, out_stride=n)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.multiply(a, b, out=out, batch=B, out_stride=n)
        ctx.synchronize()
        us = (time.perf_counter() - t0) / 5 / B * 1e6
        print("lanes %d chunk %4s: %.4f us per product, %.2f M products/s" % (lanes, chunk or "dflt", us, 1 / us), flush=True)
        plan.close()
        ctx.close()
