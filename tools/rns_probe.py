import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ntt_amd, bench
n = 65536
primes = bench.rns_primes(32, 17, 1 << 30)
ctx = ntt_amd.Context(0)
def t(fn, reps=5):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps
for B in (256, 1024, 8192):
    pl = ctx.plan(primes[0], n, B)
    a = torch.randint(0, primes[0], (B, n), device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, primes[0], (B, n), device="cuda", dtype=torch.int64).to(torch.int32)
    o = torch.zeros((B, 2 * n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    dt = t(lambda: pl.multiply(a, b, out=o, batch=B, out_stride=2 * n))
    print(f"one plan, batch {B}: {B / dt / 1e3:.0f} k products/s  (chunk {getattr(pl, 'chunk', None)})")
    pl.close(); del a, b, o
plans = [ctx.plan(p, n, 256) for p in primes]
a = torch.randint(0, min(primes), (32, 256, n), device="cuda", dtype=torch.int64).to(torch.int32)
b = torch.randint(0, min(primes), (32, 256, n), device="cuda", dtype=torch.int64).to(torch.int32)
o = torch.zeros((32, 256, 2 * n), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
def limbs():
    for i, pl in enumerate(plans):
        pl.multiply(a[i], b[i], out=o[i], batch=256, out_stride=2 * n)
dt = t(limbs)
print(f"32 plans x 256, one call each: {8192 / dt / 1e3:.0f} k products/s")
o2 = [o[i] for i in range(32)]
dt = t(lambda: ntt_amd.multiply_limbs(plans, [a[i] for i in range(32)], [b[i] for i in range(32)], o2, 256, out_stride=2 * n))
print(f"32 plans x 256, ntt_poly_multiply_limbs: {8192 / dt / 1e3:.0f} k products/s")
