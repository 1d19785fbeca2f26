"""ntt_comm (include/ntt_amd.h): the exchanges of the sharded workloads inside the library for a host program with one
process per GPU -- the reference's MPI shape (main-mpi.cc:645-797).  Host mode (the program's own all-to-all on host
memory) runs here with 2 and 4 ranks sharing the one GPU, torch.distributed over gloo standing in for MPI_Alltoall;
RCCL mode can only run at world = 1 on a one-GPU box (RCCL refuses two ranks on one device): its grouped send / recv
calls execute, against this rank itself -- between GPUs it is unexercised ("parity unpinned" on real links)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
P64 = 1337006139375617


@pytest.fixture(scope="module")
def nt():
    import ntt_amd
    return ntt_amd


@pytest.mark.parametrize("world", [2, 4])
def test_comm_host_mode_ranks_on_one_gpu(tmp_path, world):
    out = tmp_path / "comm.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29610 + world),
           os.path.join(HERE, "_comm_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok"


def test_comm_host_mode_one_rank(nt):
    """world = 1: the all-to-all is a copy; the product must equal the oracle's and the single-context one."""
    ctx = nt.Context(0)
    seen = []

    def alltoall(send, recv):
        seen.append(send.shape)
        recv[...] = send
    comm = nt.Comm.host(ctx, alltoall, 1, 0)
    n = 2500
    a = orc.splitmix64_np(11, n) % np.uint64(P64)
    b = orc.splitmix64_np(12, n) % np.uint64(P64)
    want, _ = orc.poly_multiply_crt(a, b, P64, threads=2)
    assert np.array_equal(comm.crt_multiply(a, b, P64), want)
    sl = 2 * n - 1
    assert seen == [(1, 4 * 4 * sl), (1, 8 * sl)]          # four residue slices out, the merged u64 slice gathered
    with pytest.raises(nt.NttError) as e:                   # a split transform needs world >= 2 (ntt_dist_plan_create)
        nt.DistPlan(ctx, 469762049, 16, 1, 0)
    assert e.value.code == nt.ERR_INVALID
    with pytest.raises(nt.NttError):
        nt.Comm.host(ctx, alltoall, 2, 2)                  # rank out of range
    grp = nt.Context(devices=[0, 0])
    with pytest.raises(nt.NttError) as e:
        nt.Comm.host(grp, alltoall, 2, 0)                  # a multi-device context exchanges by itself
    assert e.value.code == nt.ERR_INVALID
    grp.close()
    comm.close()
    ctx.close()


def test_comm_rccl_mode_one_rank(nt):
    """RCCL mode at world = 1: id, ncclCommInitRank, the grouped send / recv of both CRT exchanges (to this rank
    itself), device and host results."""
    import torch
    ctx = nt.Context(0)
    cid = nt.Comm.rccl_id()
    assert len(cid) == nt.COMM_ID_BYTES and any(cid)
    comm = nt.Comm.rccl(ctx, cid, 1, 0)
    assert (comm.world, comm.rank, comm.kind) == (1, 0, "rccl")
    for n in (2500, 131072):
        a = orc.splitmix64_np(21 + n, n) % np.uint64(P64)
        b = orc.splitmix64_np(22 + n, n) % np.uint64(P64)
        want = ctx.crt_multiply(a, b, P64)
        if n <= 3000:
            assert np.array_equal(want, orc.poly_multiply_crt(a, b, P64, threads=2)[0])
        assert np.array_equal(comm.crt_multiply(a, b, P64), want)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dout = torch.full((2 * n - 1,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            comm.crt_multiply(da, db, P64, out=dout)       # queued
        ctx.synchronize()
        assert np.array_equal(dout.cpu().numpy().view(np.uint64), want)
    comm.close()
    ctx.close()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_comm_ranks_program(world):
    """ntt_amd/comm_ranks (csrc/comm_ranks.cpp): the same in plain C++ -- N forked processes, a context and a host-mode
    communicator each, the all-to-all through shared memory where an MPI program passes MPI_Alltoall; every rank
    compares the large-modulus product at n = 2^17, a split transform of 2^20 points (forward, round trip) and a split
    product with the results of its own single context."""
    exe = os.path.join(os.path.dirname(HERE), "ntt_amd", "comm_ranks")
    assert os.path.exists(exe), "build it with make -C ntt_amd/csrc"
    r = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert f"comm_ranks: {world} ranks, ok" in r.stdout
