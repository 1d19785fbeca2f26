"""world_size-2 gloo test of the multi-process host path (ntt_amd.shard) on CPU."""
import json
import os
import subprocess
import sys

import numpy as np

import orc
from ntt_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_matches_reference_mpi_rule():
    # main-mpi.cc:657-660 with 4 primes
    assert [shard.partition(4, 4, r) for r in range(4)] == [(0, 1), (1, 1), (2, 1), (3, 1)]
    assert [shard.partition(4, 2, r) for r in range(2)] == [(0, 2), (2, 2)]
    assert [shard.partition(4, 3, r) for r in range(3)] == [(0, 2), (2, 2), (4, 0)]
    assert [shard.partition(4, 8, r) for r in range(8)] == [(0, 1), (1, 1), (2, 1), (3, 1)] + [(4, 0)] * 4
    for total in (1, 7, 32, 1024):
        for world in (1, 2, 3, 8):
            parts = [shard.partition(total, world, r) for r in range(world)]
            assert sum(c for _, c in parts) == total
            covered = [u for s, c in parts for u in range(s, s + c)]
            assert covered == list(range(total))
    assert shard.rank_seed(100, 3, 1024) == 100 + 3 * 1024


def test_library_partition_is_the_same_rule():
    """The multi-device contexts shard inside the C++ library (multidev.cpp): its rule, exported as ntt_partition,
    against the Python one and the reference's arithmetic (main-mpi.cc:657-660) for the four primes."""
    import ntt_amd
    for total in (0, 1, 4, 7, 32, 256, 1024, 8192):
        for world in (1, 2, 3, 4, 8, 16, 64):
            for r in range(world):
                assert ntt_amd.partition(total, world, r) == shard.partition(total, world, r), (total, world, r)
    for size in (1, 2, 3, 4, 5, 8):
        mpp = (4 + size - 1) // size                       # mods_per_process
        for rank in range(size):
            start, end = min(rank * mpp, 4), min(rank * mpp + mpp, 4)
            assert ntt_amd.partition(4, size, rank) == (start, max(0, end - start))
    with pytest.raises(ntt_amd.NttError):
        ntt_amd.partition(4, 2, 2)


def test_multi_device_context_needs_gpus():
    """ntt_ctx_create with ndev > 1 on a box without a GPU: a clean NTT_ERR_HIP, no crash, no CPU fallback."""
    import torch
    import ntt_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ntt_amd.NttError) as e:
        ntt_amd.Context(devices=[0, 0])
    assert e.value.code == ntt_amd.ERR_HIP


def test_two_ranks_gloo(tmp_path):
    out = tmp_path / "result.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(HERE, "_gloo_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["world"] == 2 and res["nunits"] == 7
    assert abs(res["tmax"] - 0.10) < 1e-9            # max over ranks, not rank 0's own 0.05
    assert res["elapsed"] >= 0.10                      # the closing barrier waits for the slow rank
    assert 0.19 <= res["waited"] < 20.0                # host_wait returned once rank 0 had signalled, not before
    n, p = 300, 998244353
    for u in range(7):
        a = (orc.splitmix64_np(0x20250711 + u, n) % np.uint64(p)).astype(np.uint32)
        b = (orc.splitmix64_np(0x20250711 + (1 << 20) + u, n) % np.uint64(p)).astype(np.uint32)
        assert res["sums"][str(u)] == int(orc.poly_multiply(a, b, p).astype(np.uint64).sum())


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_crt_sharded_gloo(tmp_path, world):
    """Prime-per-rank CRT with the all-to-all exchange, 2 and 3 ranks (3: one rank owns
    no prime, as in main-mpi.cc:657-660 with size = 3)."""
    out = tmp_path / "crt.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29540 + world),
           os.path.join(HERE, "_gloo_crt_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok"


@pytest.mark.parametrize("world", [2, 4])
def test_dist_transform_gloo(tmp_path, world):
    """One transform split over 2 / 4 ranks with two all-to-all exchanges, forward
    and inverse, against the oracle's full-length transform."""
    out = tmp_path / "dist.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29550 + world),
           os.path.join(HERE, "_gloo_dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok"


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no torchrun around it (the way the driver calls it): bench.py starts
    the two ranks itself as a child job (main-mpi.cc:642-665 brings up its ranks the same way) and rank 0 prints
    ONE line with n_gpus == 2.  No GPU here, so --dry-run stops after the rendezvous, barrier and max-over-ranks."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 2
    # a world that differs from --gpus is an error, not a silent one-rank run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "started 3 ranks" in r.stderr
