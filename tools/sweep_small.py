import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ntt_amd
ctx = ntt_amd.Context(0)
p = 469762049
rng = np.random.default_rng(0)
for lgn in (12, 13, 14, 15):
    n = 1 << lgn
    B = 4096 if lgn <= 13 else 2048
    a = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, p, size=(B, n), dtype=np.int64).astype(np.int32)).cuda()
    for name in ("linear", "negacyclic"):
        plan = ctx.plan(p, n, B) if name == "linear" else ctx.negacyclic_plan(p, n, B)
        stride = 2 * n if name == "linear" else n
        out = torch.zeros((B, stride), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.multiply(a, b, out=out, batch=B, out_stride=stride)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 5 / B
        print("lgn %d %-10s L=%6d  %.4f us per product" % (lgn, name, plan.length, dt * 1e6), flush=True)
        plan.close()
