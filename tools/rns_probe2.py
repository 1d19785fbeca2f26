"""32 RNS limbs x 256 products at n = 2^16 through ntt_poly_multiply_limbs, chunk size of a limb's launches swept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntt_amd, bench
n = 65536
primes = bench.rns_primes(32, 17, 1 << 30)
ctx = ntt_amd.Context(0)
plans = [ctx.plan(p, n, 256) for p in primes]
a = torch.randint(0, min(primes), (32, 256, n), device="cuda", dtype=torch.int64).to(torch.int32)
b = torch.randint(0, min(primes), (32, 256, n), device="cuda", dtype=torch.int64).to(torch.int32)
o = torch.zeros((32, 256, 2 * n), dtype=torch.int32, device="cuda")
al, bl, ol = [a[i] for i in range(32)], [b[i] for i in range(32)], [o[i] for i in range(32)]
torch.cuda.synchronize()
for chunk in (16, 32, 43, 52, 64, 86, 128):
    os.environ["NTT_AMD_LIMB_CHUNK"] = str(chunk)
    best = 0
    for _ in range(3):
        ntt_amd.multiply_limbs(plans, al, bl, ol, 256, out_stride=2 * n); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): ntt_amd.multiply_limbs(plans, al, bl, ol, 256, out_stride=2 * n)
        ctx.synchronize()
        best = max(best, 5 * 8192 / (time.perf_counter() - t0))
    print(f"limb chunk {chunk}: {best / 1e3:.0f} k products/s")
for pl in plans: pl.close()
os.environ.pop("NTT_AMD_LIMB_CHUNK", None)
# the same for the limbs of x^n + 1 (negacyclic plans, rows of n coefficients)
nplans = [ctx.negacyclic_plan(p, n, 256) for p in primes]
def per_call():
    for i, pl in enumerate(nplans): pl.multiply(al[i], bl[i], out=ol[i], batch=256, out_stride=2 * n)
def one_call():
    ntt_amd.multiply_limbs(nplans, al, bl, ol, 256, out_stride=2 * n)
for name, fn in (("one call per limb", per_call), ("ntt_poly_multiply_limbs", one_call)):
    best = 0
    for _ in range(3):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        ctx.synchronize()
        best = max(best, 5 * 8192 / (time.perf_counter() - t0))
    print(f"negacyclic 32 x 256, {name}: {best / 1e3:.0f} k products/s")
