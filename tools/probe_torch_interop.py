"""Probe: can torch.cuda and libntt_amd.so (system HIP runtime) share one process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
order = sys.argv[1]
import numpy as np
if order == "torch_first":
    import torch
    print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
    t = torch.arange(8, device="cuda")
    print(t.sum().item())
import ntt_amd, orc
ctx = ntt_amd.Context(0)
n, p = 8192, 998244353
rng = np.random.default_rng(0)
a = rng.integers(0, p, size=(2, n)).astype(np.uint32); b = rng.integers(0, p, size=(2, n)).astype(np.uint32)
plan = ctx.plan(p, n, 2)
ab = plan.multiply(a, b)
print("host ptr path ok", np.array_equal(ab[0], orc.poly_multiply(a[0], b[0], p)))
if order != "torch_first":
    import torch
    print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
ta = torch.from_numpy(a.view(np.int32)).cuda(); tb = torch.from_numpy(b.view(np.int32)).cuda()
tab = torch.zeros((2, 2*n-1), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
plan.multiply(ta, tb, out=tab, batch=2)
ctx.synchronize()
print("device ptr path ok", np.array_equal(tab.cpu().numpy().view(np.uint32)[1], orc.poly_multiply(a[1], b[1], p)))
