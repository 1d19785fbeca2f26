#!/usr/bin/env python3
"""The reference's recorded benchmark (test.o: stdout of main-mpi.cc on cases 0-4) on
this framework: writes the golden cases 0-3 (tests/golden) and a synthetic case 4 (n = 131072,
p = 1337006139375617, splitmix64 operands; the reference's nttdata/4.* is not shipped, so 4.out is the product
of the direct 64-bit path, a different route from the launcher's 4-prime CRT)
as text files into a scratch directory and runs `python -m ntt_amd.launch` on them, 50 epochs:
one rank (device tensors, "nccl" route) and four ranks sharing this GPU (exchanges over gloo)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
d = tempfile.mkdtemp()
data, out = os.path.join(d, "nttdata"), os.path.join(d, "files")
os.makedirs(data)
os.makedirs(out)
open(os.path.join(data, "0.in"), "wb").write(open(os.path.join(G, "nttdata_0.in"), "rb").read())
open(os.path.join(data, "0.out"), "wb").write(open(os.path.join(G, "nttdata_0.out"), "rb").read())
for cid in (1, 2, 3):
    z = np.load(os.path.join(G, f"nttdata_{cid}.npz"))
    with open(os.path.join(data, f"{cid}.in"), "w") as f:
        f.write(f"{int(z['n'])} {int(z['p'])}\n" + " ".join(map(str, z["a"])) + "\n" + " ".join(map(str, z["b"])) + "\n")
    with open(os.path.join(data, f"{cid}.out"), "w") as f:
        f.write(" ".join(map(str, z["ab"])) + "\n")
from bench import splitmix64  # noqa: E402
p64, n = 1337006139375617, 131072
a = splitmix64(0x20250711, n) % np.uint64(p64)
b = splitmix64(0x20250711 + (1 << 20), n) % np.uint64(p64)
with open(os.path.join(data, "4.in"), "w") as f:
    f.write(f"{n} {p64}\n" + " ".join(map(str, a)) + "\n" + " ".join(map(str, b)) + "\n")
import ntt_amd  # noqa: E402
ctx = ntt_amd.Context(0)
plan = ctx.plan64(p64, n)
with open(os.path.join(data, "4.out"), "w") as f:
    f.write(" ".join(map(str, plan.multiply(a, b)[0])) + "\n")
plan.close()
ctx.close()
env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
for world, backend in ((1, "nccl"), (4, "gloo")):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), "-m", "ntt_amd.launch",
           "--data", data, "--out", out, "--begin", "0", "--end", "4", "--epochs", "50", "--backend", backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    print(f"== {world} rank(s), {backend}: exit {r.returncode}")
    print(r.stdout)
    print("\n".join(l for l in r.stderr.split("\n") if "amdgpu.ids" not in l and l.strip())[-600:])
