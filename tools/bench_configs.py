#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] on ONE MI355X (bench.py stays the headline,
configs[1]).  One JSON line per config; results are kept under profiles/.

  config 3: large-modulus product through the 4-prime CRT, n = 2^17, p = 1337006139375617,
            device-resident u64 in/out (ntt_crt_multiply: reduce, 4 residue products, Garner merge)
  config 4: FHE-style RNS batch, n = 2^16, 32 primes = 1 (mod 2^17) above 2^30, 256 polynomial pairs
            per limb (8192 products), device-resident
  config 5: one forward + one inverse transform of length 2^26, p = 469762049 (natural order API)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rns_primes(count, two_adicity, above):
    out, step = [], 1 << two_adicity
    p = (above // step + 1) * step + 1
    while len(out) < count:
        if all(p % q for q in range(3, int(p ** 0.5) + 1, 2)):
            out.append(p)
        p += step
    return out


def main():
    import torch
    import ntt_amd
    import orc

    ctx = ntt_amd.Context(0)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    # ---- headline workload with HOST buffers across the boundary (PCIe-inclusive; never bench.py's value)
    n, p, B = 131072, 998244353, 256
    ha = (orc.splitmix64_np(1, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    hb = (orc.splitmix64_np(2, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    hab = np.zeros((B, 2 * n - 1), dtype=np.uint32)
    plan = ctx.plan(p, n, B, ntt_amd.MONTGOMERY)
    plan.multiply(ha, hb, out=hab)
    t0 = time.perf_counter()
    for _ in range(3):
        plan.multiply(ha, hb, out=hab)
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"config": "1 with pageable HOST buffers handed over the C ABI (H2D a, b + D2H ab inside the call)",
                      "batch": B, "poly_mults_per_s": B / dt, "GBps_over_pcie": B * (16 * n - 4) / dt / 1e9}), flush=True)
    plan.close()

    # ---- the same with the streaming entry point: slices of 64 products in a three-slot ring, copies and
    # kernels on separate streams; pageable arrays are staged through pinned bounce buffers, pinned
    # arrays (ntt_host_alloc) are read and written by DMA directly
    B = 1024
    plan = ctx.plan(p, n, 64, ntt_amd.MONTGOMERY)
    pa, pb = ntt_amd.host_array((B, n)), ntt_amd.host_array((B, n))
    pab = ntt_amd.host_array((B, 2 * n - 1))
    pa[:] = (orc.splitmix64_np(1, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    pb[:] = (orc.splitmix64_np(2, B * n) % np.uint64(p)).astype(np.uint32).reshape(B, n)
    qa, qb = np.array(pa), np.array(pb)                      # pageable copies
    qab = np.zeros((B, 2 * n - 1), dtype=np.uint32)
    for name, (xa_, xb_, xo_) in (("pinned", (pa, pb, pab)), ("pageable", (qa, qb, qab))):
        plan.multiply_stream(xa_[:128], xb_[:128], out=xo_[:128])
        dt = 1e9
        for _ in range(2):                                   # the first pass also faults in the output pages
            t0 = time.perf_counter()
            plan.multiply_stream(xa_, xb_, out=xo_)
            dt = min(dt, time.perf_counter() - t0)
        print(json.dumps({"config": "1 streamed from %s HOST arrays (ntt_poly_multiply_stream, 1024 products)" % name,
                          "poly_mults_per_s": B / dt, "GBps_each_way_over_pcie": B * (8 * n) / dt / 1e9}), flush=True)
    assert np.array_equal(pab, qab)
    assert np.array_equal(hab[:4], pab[:4])
    plan.close()
    del pa, pb, pab, qa, qb, qab

    # ---- config 3 ------------------------------------------------------------------------------
    n, p64 = 131072, 1337006139375617
    a = orc.splitmix64_np(0x20250711, n) % np.uint64(p64)
    b = orc.splitmix64_np(0x20250711 + 1, n) % np.uint64(p64)
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    dab = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ctx.crt_multiply(da, db, p64, out=dab)
    sync()
    want, _ = orc.poly_multiply_crt(a, b, p64, threads=8)
    assert np.array_equal(dab.cpu().numpy().view(np.uint64), want)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.crt_multiply(da, db, p64, out=dab)
    sync()
    dt = (time.perf_counter() - t0) / reps
    _, cpu_s = orc.poly_multiply_crt(a, b, p64, threads=min(16, os.cpu_count() or 1))
    print(json.dumps({"config": "3: 4-prime CRT product, n=2^17, p=1337006139375617, one GPU, device-resident",
                      "ms_per_product": dt * 1e3, "products_per_s": 1 / dt,
                      "cpu_oracle_ms_per_product": cpu_s * 1e3,
                      "reference_published_ms": 90.7393}), flush=True)

    B3 = 64
    ya = torch.from_numpy((orc.splitmix64_np(5, B3 * n) % np.uint64(p64)).view(np.int64).reshape(B3, n)).cuda()
    yb = torch.from_numpy((orc.splitmix64_np(6, B3 * n) % np.uint64(p64)).view(np.int64).reshape(B3, n)).cuda()
    yo = torch.zeros((B3, 2 * n - 1), dtype=torch.int64, device="cuda")
    ctx.crt_multiply_batched(ya, yb, p64, out=yo)
    sync()
    assert np.array_equal(yo[0].cpu().numpy().view(np.uint64),
                          orc.poly_multiply_crt(ya[0].cpu().numpy().view(np.uint64), yb[0].cpu().numpy().view(np.uint64), p64, threads=8)[0])
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.crt_multiply_batched(ya, yb, p64, out=yo)
    sync()
    dtb3 = (time.perf_counter() - t0) / 10 / B3
    print(json.dumps({"config": "3 batched: 64 CRT products per call (ntt_crt_multiply_batched)",
                      "ms_per_product": dtb3 * 1e3, "products_per_s": 1 / dtb3}), flush=True)
    del ya, yb, yo

    # ---- config 3 through the direct 64-bit path (one 64-bit transform per operand instead of four primes)
    plan64 = ctx.plan64(p64, n, batch=64)
    plan64.multiply(da, db, out=dab, batch=1)
    sync()
    assert np.array_equal(dab.cpu().numpy().view(np.uint64), want)
    t0 = time.perf_counter()
    for _ in range(reps):
        plan64.multiply(da, db, out=dab, batch=1)
    sync()
    dt1 = (time.perf_counter() - t0) / reps
    B64 = 64
    xa = torch.from_numpy((orc.splitmix64_np(5, B64 * n) % np.uint64(p64)).view(np.int64).reshape(B64, n)).cuda()
    xb = torch.from_numpy((orc.splitmix64_np(6, B64 * n) % np.uint64(p64)).view(np.int64).reshape(B64, n)).cuda()
    xo = torch.zeros((B64, 2 * n), dtype=torch.int64, device="cuda")
    plan64.multiply(xa, xb, out=xo, batch=B64, out_stride=2 * n)
    sync()
    t0 = time.perf_counter()
    for _ in range(10):
        plan64.multiply(xa, xb, out=xo, batch=B64, out_stride=2 * n)
    sync()
    dtb = (time.perf_counter() - t0) / 10 / B64
    print(json.dumps({"config": "3 via the direct 64-bit path (ntt_plan64), n=2^17, p=1337006139375617, device-resident",
                      "ms_per_product_single": dt1 * 1e3, "ms_per_product_batch64": dtb * 1e3,
                      "products_per_s_batch64": 1 / dtb}), flush=True)
    plan64.close()

    # ---- config 4 ------------------------------------------------------------------------------
    n, limbs, per = 65536, 32, 256
    primes = rns_primes(limbs, 17, 1 << 30)
    plans = [ctx.plan(p, n, per, ntt_amd.MONTGOMERY) for p in primes]
    rng = np.random.default_rng(1)
    xa = torch.from_numpy(rng.integers(0, primes[0], size=(per, n), dtype=np.int64).astype(np.int32)).cuda()
    xb = torch.from_numpy(rng.integers(0, primes[0], size=(per, n), dtype=np.int64).astype(np.int32)).cuda()
    out = torch.zeros((per, 2 * n), dtype=torch.int32, device="cuda")

    def rns_step():
        for pl in plans:          # same buffers for every limb: values < the smallest prime are valid everywhere
            pl.multiply(xa, xb, out=out, batch=per, out_stride=2 * n)

    rns_step()
    sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rns_step()
    sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": "4: RNS batch n=2^16, 32 limbs x 256 pairs (strict [0,p) kernels, p > 2^30), one GPU",
                      "ms_per_batch": dt * 1e3, "poly_mults_per_s": limbs * per / dt}), flush=True)
    for pl in plans:
        pl.close()

    # ---- config 4 in the ring FHE workloads use: negacyclic products mod (x^n + 1), same shape
    nplans = [ctx.negacyclic_plan(p, n, per, ntt_amd.MONTGOMERY) for p in primes]
    nout = torch.zeros((per, n), dtype=torch.int32, device="cuda")

    def neg_step():
        for pl in nplans:
            pl.multiply(xa, xb, out=nout, batch=per)

    neg_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        neg_step()
    sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": "4 negacyclic (x^n + 1): n=2^16, 32 limbs x 256 pairs, one transform of length n per operand",
                      "ms_per_batch": dt * 1e3, "poly_mults_per_s": limbs * per / dt}), flush=True)
    for pl in nplans:
        pl.close()

    # ---- the same ring with 64-bit limbs (FHE moduli of ~50 bits): 16 primes k * 2^17 + 1 just below 2^50 and,
    # for the Montgomery lanes, just below 2^62; 64 pairs per limb
    def is_prime(q):
        if q % 2 == 0:
            return q == 2
        d, s = q - 1, 0
        while d % 2 == 0:
            d //= 2
            s += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if q % a == 0:
                return q == a
            x = pow(a, d, q)
            if x in (1, q - 1):
                continue
            for _ in range(s - 1):
                x = x * x % q
                if x == q - 1:
                    break
            else:
                return False
        return True

    def primes_below(bits, count):
        out, k = [], (1 << bits) >> 17
        while len(out) < count:
            k -= 1
            if is_prime(k * (1 << 17) + 1):
                out.append(k * (1 << 17) + 1)
        return out

    per64 = 64
    rng64 = np.random.default_rng(64)
    for bits, label in ((50, "exact-integer FP64 lanes"), (62, "64-bit Montgomery lanes")):
        primes64 = primes_below(bits, 16)
        ya = torch.from_numpy(rng64.integers(0, primes64[-1], size=(per64, n), dtype=np.uint64).view(np.int64)).cuda()
        yb = torch.from_numpy(rng64.integers(0, primes64[-1], size=(per64, n), dtype=np.uint64).view(np.int64)).cuda()
        yout = torch.zeros((per64, n), dtype=torch.int64, device="cuda")
        plans64 = [ctx.negacyclic_plan64(q, n, per64) for q in primes64]

        def step64():
            for pl in plans64:
                pl.multiply(ya, yb, out=yout, batch=per64)

        step64()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            step64()
        sync()
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"config": f"4 negacyclic with 64-bit limbs: n=2^16, 16 primes below 2^{bits} x 64 pairs ({label})",
                          "ms_per_batch": dt * 1e3, "poly_mults_per_s": 16 * per64 / dt}), flush=True)
        for pl in plans64:
            pl.close()

    # ---- config 5 ------------------------------------------------------------------------------
    p, L = 469762049, 1 << 26
    x = torch.from_numpy((orc.splitmix64_np(7, L) % np.uint64(p)).astype(np.uint32).view(np.int32)).cuda()
    plan = ctx.plan(p, L // 2, 1, ntt_amd.MONTGOMERY)
    plan.forward(x, 1)
    plan.inverse(x, 1)
    sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.forward(x, 1)
        plan.inverse(x, 1)
    sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": "5: forward + inverse transform, L=2^26, p=469762049, one GPU (natural order in/out)",
                      "ms_per_roundtrip": dt * 1e3, "gbytes_algorithmic": 4 * L * 4 / 1e9,
                      "algorithmic_GBps": 4 * L * 4 / dt / 1e9}), flush=True)
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
