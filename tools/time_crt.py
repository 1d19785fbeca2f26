import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import numpy as np, torch, ntt_amd
p64, n, B = 1337006139375617, 131072, int(os.environ.get("CRT_B", "64"))
ctx = ntt_amd.Context(0)
rng = np.random.default_rng(1)
a = torch.from_numpy(rng.integers(0, p64, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
b = torch.from_numpy(rng.integers(0, p64, size=(B, n), dtype=np.uint64).view(np.int64)).cuda()
o = torch.zeros((B, 2 * n - 1), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for timing in (False, True):
    ctx.enable_kernel_timing(timing)
    ctx.crt_multiply_batched(a, b, p64, out=o); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.crt_multiply_batched(a, b, p64, out=o)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("timing", timing, "%.1f us per call, %.2f per product" % (dt * 1e6, dt * 1e6 / B))
    if timing:
        for k, (ms, cnt) in ctx.kernel_times().items():
            if cnt: print("  %-12s %5d launches %.1f us avg, %.1f us per call" % (k, cnt, ms * 1e3 / cnt, ms * 1e3 / 11))
