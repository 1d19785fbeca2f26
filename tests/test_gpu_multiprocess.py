"""One process per rank around the real kernels: 2 and 4 ranks launched by torch.distributed.run, every
rank computing through the C ABI on the GPU (a one-GPU box: all ranks share cuda:0), exchanges over gloo.
Covers the prime-per-rank CRT with its all-to-all, the split transform with its two all-to-alls, and the
batch partition of bench.py.  (RCCL between GPUs is exercised by the driver's multi-GPU bench only.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu(tmp_path, world):
    out = tmp_path / "mp.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29570 + world),
           os.path.join(HERE, "_gpu_mp_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok"
