"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with backend gloo.
Exercises the N>1 host path of bench.py (ntt_amd.shard): rank/world from the environment, unit
partition, per-rank seeds, barrier, max/sum over ranks.  The per-unit compute is done by the CPU
oracle here (no GPU in this container); on a GPU box the same code paths drive libntt_amd.so."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import orc  # noqa: E402
from ntt_amd import shard  # noqa: E402


def main():
    out_path = sys.argv[1]
    g = shard.Group("gloo")
    total, n, p = 7, 300, 998244353
    start, count = shard.partition(total, g.world, g.rank)
    sums = {}
    for u in range(start, start + count):
        a = (orc.splitmix64_np(0x20250711 + u, n) % np.uint64(p)).astype(np.uint32)
        b = (orc.splitmix64_np(0x20250711 + (1 << 20) + u, n) % np.uint64(p)).astype(np.uint32)
        sums[u] = int(orc.poly_multiply(a, b, p).astype(np.uint64).sum())
    g.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (g.rank + 1))          # rank 1 is slower: the max must report its time
    g.barrier()
    elapsed = time.perf_counter() - t0
    tmax = g.max_over_ranks(0.05 * (g.rank + 1))
    nunits = g.sum_over_ranks(count)
    # host-side wait on rank 0 (bench.py: the other ranks idle while rank 0 runs its child process): no collective
    t1 = time.perf_counter()
    if g.rank == 0:
        time.sleep(0.2)
        g.host_signal("rank0_done")
    else:
        g.host_wait("rank0_done", 30.0)
    waited = g.max_over_ranks(time.perf_counter() - t1)
    gathered = [None] * g.world
    g.dist.all_gather_object(gathered, sums)
    if g.rank == 0:
        merged = {}
        for d in gathered:
            merged.update(d)
        json.dump({"world": g.world, "tmax": tmax, "nunits": nunits, "elapsed": elapsed, "waited": waited,
                   "sums": {str(k): v for k, v in merged.items()}}, open(out_path, "w"))
    g.close()


if __name__ == "__main__":
    main()
