"""Does replaying a captured HIP graph beat launching the kernels of ONE product per call?  torch captures the library's
launches (the context runs on torch's stream; its second lane joins the capture through the fork / join events).
Cases: the 4-prime large-modulus product (14 launches on two lanes) and the 32-bit product (3 launches)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch, ntt_amd
os.environ["NTT_AMD_CRT_DIRECT"] = "0"
n = 131072
rng = np.random.default_rng(1)

def bench(name, fn, ctx):
    for _ in range(5):
        fn()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
        torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(f"{name}: {dt * 1e6:.1f} us per product, synchronised each call", flush=True)

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = ntt_amd.Context(0)
    ctx.use_torch_stream()
    p64 = 1337006139375617
    a = torch.from_numpy(rng.integers(0, p64, size=n, dtype=np.uint64).view(np.int64)).cuda()
    b = torch.from_numpy(rng.integers(0, p64, size=n, dtype=np.uint64).view(np.int64)).cuda()
    o = torch.zeros(2 * n - 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    bench("large-modulus product, launches", lambda: ctx.crt_multiply(a, b, p64, out=o), ctx)
    want = o.clone()
    o.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.crt_multiply(a, b, p64, out=o)
    bench("large-modulus product, graph replay", g.replay, ctx)
    print("  same result:", torch.equal(o, want))
    p = 998244353
    plan = ctx.plan(p, n, 1)
    a32 = torch.randint(0, p, (1, n), device="cuda", dtype=torch.int64).to(torch.int32)
    b32 = torch.randint(0, p, (1, n), device="cuda", dtype=torch.int64).to(torch.int32)
    o32 = torch.zeros((1, 2 * n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    bench("32-bit product, launches", lambda: plan.multiply(a32, b32, out=o32, batch=1, out_stride=2 * n), ctx)
    want = o32.clone()
    o32.zero_()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        plan.multiply(a32, b32, out=o32, batch=1, out_stride=2 * n)
    bench("32-bit product, graph replay", g2.replay, ctx)
    print("  same result:", torch.equal(o32, want))
