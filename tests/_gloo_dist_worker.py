"""Worker for tests/test_distributed_cpu.py::test_dist_transform_gloo: ntt_amd.shard.dist_transform with
its two all-to-all exchanges over gloo.  The per-rank compute (world-point column pass with 4-step
twiddles, local S-point transform) is done with Python integers / the CPU oracle here; on GPU boxes
the same driver is fed by DistPlan.colpass / DistPlan.local (tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import orc  # noqa: E402
from ntt_amd import shard  # noqa: E402


def main():
    out_path = sys.argv[1]
    g = shard.Group("gloo")
    G, rank = g.world, g.rank
    p, lgL = 998244353, 10
    L = 1 << lgL
    S, cols = L // G, L // G // G
    x = (orc.splitmix64_np(99, L) % np.uint64(p)).astype(np.uint32)
    X = orc.ntt(x, p)
    w = pow(3, (p - 1) // L, p)

    def colpass(arr, inv):
        ww = pow(w, p - 2, p) if inv else w
        wG = pow(ww, L // G, p)
        a = arr.numpy().view(np.uint32).astype(object)
        out = np.zeros((G, cols), dtype=np.uint32)
        invG = pow(G, p - 2, p)
        for j in range(cols):
            i2 = rank * cols + j
            for o in range(G):
                if not inv:
                    acc = sum(int(a[i1][j]) * pow(wG, i1 * o, p) for i1 in range(G)) % p
                    out[o][j] = acc * pow(ww, i2 * o, p) % p
                else:
                    out[o][j] = sum(int(a[k1][j]) * pow(ww, i2 * k1, p) % p * pow(wG, o * k1, p)
                                    for k1 in range(G)) % p * invG % p
        return torch.from_numpy(out.view(np.int32))

    def local(row, inv):
        return torch.from_numpy(orc.ntt(row.numpy().view(np.uint32), p, inverse=inv).view(np.int32))

    a2a = shard.torch_all_to_all(g)
    mine = torch.from_numpy(x[rank * S:(rank + 1) * S].view(np.int32).copy())
    fwd = shard.dist_transform(G, rank, a2a, colpass, local, mine, inverse=False)
    ok = bool(np.array_equal(fwd.numpy().view(np.uint32), X[rank::G]))
    back = shard.dist_transform(G, rank, a2a, colpass, local, fwd, inverse=True)
    ok = ok and bool(np.array_equal(back.numpy().view(np.uint32), x[rank * S:(rank + 1) * S]))
    flags = [None] * G
    g.dist.all_gather_object(flags, ok)
    if rank == 0:
        open(out_path, "w").write("ok" if all(flags) else "mismatch")
    g.close()


if __name__ == "__main__":
    main()
