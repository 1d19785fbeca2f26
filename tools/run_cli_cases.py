#!/usr/bin/env python3
"""Writes the golden cases (tests/golden) as the reference's text files into a scratch directory and runs
ntt_amd/ntt_cli on them (50 epochs): the harness's own latency lines, device-resident single products."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for extra in ([], ["--barrett"]):
    r = subprocess.run([os.path.join(ROOT, "ntt_amd", "ntt_cli"), "--data", data, "--out", out, "--begin", "0", "--end", "3"] + extra,
                       capture_output=True, text=True)
    print(" ".join(["ntt_cli"] + extra), "->", r.returncode)
    print(r.stdout)
    print(r.stderr[-500:])
