#!/usr/bin/env python3
"""Latency of the reference's own entry point, void poly_multiply(int *a, int *b, int *ab, int n, int p)
(main-montgomery.cc:71), as the unmodified harness would call it: pageable host arrays in and out, one
product per call, nothing resident.  Also the handle API on the same host arrays and on device arrays."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ntt_amd  # noqa: E402

rng = np.random.default_rng(3)
for n, p in ((4, 7340033), (4096, 998244353), (131072, 998244353)):
    a = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
    b = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
    ab = np.zeros(2 * n - 1, dtype=np.int32)
    call = lambda: ntt_amd._lib.poly_multiply(a.ctypes.data, b.ctypes.data, ab.ctypes.data, n, p)
    for _ in range(5):
        call()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print("poly_multiply(host int arrays)  n = %6d: %.1f us per call" % (n, dt * 1e6), flush=True)
    ctx = ntt_amd.Context(0)
    plan = ctx.plan(p, n, 1)
    au, bu = a.view(np.uint32).reshape(1, n), b.view(np.uint32).reshape(1, n)
    out = np.zeros((1, 2 * n - 1), dtype=np.uint32)
    for _ in range(5):
        plan.multiply(au, bu, out=out)
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.multiply(au, bu, out=out)
    dt = (time.perf_counter() - t0) / reps
    print("plan.multiply(host arrays)      n = %6d: %.1f us per call" % (n, dt * 1e6), flush=True)
    da, db = torch.from_numpy(au.view(np.int32)).cuda(), torch.from_numpy(bu.view(np.int32)).cuda()
    do = torch.zeros((1, 2 * n - 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        plan.multiply(da, db, out=do)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.multiply(da, db, out=do)
        ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("plan.multiply(device arrays)    n = %6d: %.1f us per call (synchronised per call)" % (n, dt * 1e6), flush=True)
    plan.close()
    ctx.close()

# the same entry point on page-locked arrays (ntt_host_alloc): what registering the caller's arrays could buy
n, p = 131072, 998244353
pa, pb = ntt_amd.host_array(n, np.int32), ntt_amd.host_array(n, np.int32)
pab = ntt_amd.host_array(2 * n - 1, np.int32)
pa[:] = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
pb[:] = rng.integers(0, p, size=n, dtype=np.int64).astype(np.int32)
call = lambda: ntt_amd._lib.poly_multiply(pa.ctypes.data, pb.ctypes.data, pab.ctypes.data, n, p)
for _ in range(5):
    call()
t0 = time.perf_counter()
for _ in range(200):
    call()
print("poly_multiply(pinned int arrays) n = %6d: %.1f us per call" % (n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)

