import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, ntt_amd
p, n, B = 1337006139375617, 1 << 18, 32
ctx = ntt_amd.Context(0)
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
o = torch.zeros((B, 2 * n), dtype=torch.int64, device="cuda")
plan = ctx.plan64(p, n, B)
for _ in range(2): plan.multiply(a, b, out=o, batch=B, out_stride=2 * n)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): plan.multiply(a, b, out=o, batch=B, out_stride=2 * n)
ctx.synchronize()
print("n=2^18 64-bit: %.2f us per product (L=%d)" % ((time.perf_counter() - t0) / 5 / B * 1e6, plan.length))
