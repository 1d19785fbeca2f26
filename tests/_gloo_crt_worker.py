"""Worker for tests/test_distributed_cpu.py::test_crt_sharded_gloo: the prime-per-rank CRT exchange of
ntt_amd.shard.crt_multiply_sharded over gloo.  Residue products and the merge come from the CPU oracle /
Python integers here (no GPU in this container); on GPU boxes the same driver is fed by
Context.crt_residue_multiply and Context.crt_merge (tests/test_gpu_parity.py::test_crt_sharded_world1)."""
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
    n, p64 = 300, 1337006139375617
    a = orc.splitmix64_np(11, n) % np.uint64(p64)
    b = orc.splitmix64_np(12, n) % np.uint64(p64)
    want, res, _ = orc.poly_multiply_crt(a, b, p64, threads=1, want_residues=True)
    calls = []

    def residue_fn(k):
        calls.append(k)
        return torch.from_numpy(res[k, :2 * n - 1].astype(np.uint32).view(np.int32).copy())

    mods = orc.CRT_MODS
    M = 1
    for m in mods:
        M *= m

    def merge_fn(rows, count):
        out = np.zeros(count, dtype=np.uint64)
        rr = [r.numpy().view(np.uint32) for r in rows]
        for i in range(count):
            x = 0
            for k, m in enumerate(mods):          # Gauss form of main-openmp.cc:473-484 in Python integers
                Mi = M // m
                x += int(rr[k][i]) * pow(Mi % m, m - 2, m) % m * Mi
            out[i] = (x % M) % p64
        return torch.from_numpy(out.view(np.int64).copy())

    got = shard.crt_multiply_sharded(g, residue_fn, merge_fn, n).numpy().view(np.uint64)
    ok = bool(np.array_equal(got, want))
    owned = shard.partition(4, g.world, g.rank)
    assert calls == list(range(owned[0], owned[0] + owned[1])), (calls, owned)
    # the same exchange with preallocated buffers (shard.CrtExchange, what ntt_amd.launch runs per epoch): the
    # context's two halves are stood in for by the oracle's residues and the Python merge above; two epochs on
    # the same buffers
    class OracleHalves:
        def crt_residue_multiply(self, a_, b_, k, out=None):
            out[:2 * n - 1] = res[k, :2 * n - 1].astype(np.uint32)

        def crt_merge(self, rows, p, out=None, count=None):
            out[:count] = merge_fn([torch.from_numpy(r.view(np.int32)) for r in rows], count).numpy().view(np.uint64)

    xchg = shard.CrtExchange(g, OracleHalves(), n, p64)
    ta, tb = torch.from_numpy(a.view(np.int64).copy()), torch.from_numpy(b.view(np.int64).copy())
    for _ in range(2):
        got2 = xchg.multiply(ta, tb).numpy().view(np.uint64)
        ok = ok and bool(np.array_equal(got2, want))
    flags = [None] * g.world
    g.dist.all_gather_object(flags, ok)
    if g.rank == 0:
        open(out_path, "w").write("ok" if all(flags) else "mismatch")
    g.close()


if __name__ == "__main__":
    main()
