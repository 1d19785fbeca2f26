"""One process per rank around the real kernels: 2 and 4 ranks launched by torch.distributed.run, every
rank computing through the C ABI on the GPU (a one-GPU box: all ranks share cuda:0), exchanges over gloo.
Covers the prime-per-rank CRT with its all-to-all, the split transform with its two all-to-alls, and the
batch partition of bench.py.  (RCCL between GPUs is exercised by the driver's multi-GPU bench only.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

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


def _write_case(data, cid, n, p, a, b, ab):
    with open(data / f"{cid}.in", "w") as f:              # the reference's layout
        f.write(f"{n} {p}\n" + "".join(f"{v} " for v in a) + "\n" + "".join(f"{v} " for v in b))
    with open(data / f"{cid}.out", "w") as f:
        f.write("".join(f"{v} " for v in ab))


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (1, "gloo"), (2, "gloo"), (4, "gloo")])
def test_launcher_rank_layout(tmp_path, nttdata, world, backend):
    """python -m ntt_amd.launch = main() of main-mpi.cc:642-819 with one process per rank: rank 0 reads and
    broadcasts, 32-bit cases run on rank 0, the 51-bit case shards its four primes over the ranks.  Cases 0
    and 1 are the reference's own (golden outputs; "nccl" at one rank = the device-tensor route without a peer); case 2 is a 33-bit prime (64-bit lanes on rank 0), case 4
    the reference's 51-bit modulus on synthetic operands with the CRT oracle's product as <id>.out."""
    data, out = tmp_path / "nttdata", tmp_path / "files"
    data.mkdir()
    out.mkdir()
    for cid in (0, 1):
        _write_case(data, cid, *nttdata[cid])
    rng = np.random.default_rng(40 + world)
    expected = {}
    for cid, p, n in ((2, 8588886017, 700), (4, 1337006139375617, 3001)):
        a = rng.integers(0, p, size=n, dtype=np.uint64)
        b = rng.integers(0, p, size=n, dtype=np.uint64)
        ab, _ = orc.poly_multiply_crt(a, b, p, threads=2)
        _write_case(data, cid, n, p, a, b, ab)
        expected[cid] = ab
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2",
               PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29580 + world),
           "-m", "ntt_amd.launch", "--data", str(data), "--out", str(out), "--begin", "0", "--end", "4",
           "--epochs", "3", "--backend", backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 1, r.stderr[-3000:]              # case 3 is missing: reported in the exit status
    assert "case 3 skipped" in r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines.count("多项式乘法结果正确") == 4 and "多项式乘法结果错误" not in r.stdout
    assert any(l.startswith("average latency for n = 3001 p = 1337006139375617 : ") and l.endswith(" (us)")
               for l in lines)
    assert np.array_equal(np.loadtxt(out / "1.out", dtype=np.uint64).astype(np.uint32), nttdata[1][4])
    for cid, ab in expected.items():
        assert np.array_equal(np.loadtxt(out / f"{cid}.out", dtype=np.uint64), ab)
    # packed cases through the same launcher (one configuration is enough: the loader runs on rank 0 only)
    if (world, backend) == (2, "gloo"):
        import ntt_amd
        for cid in (0, 1, 2, 4):
            ntt_amd.pack_case(data, cid)
        r = subprocess.run(cmd + ["--bin"], env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 1 and "case 3 skipped" in r.stderr, r.stderr[-2000:]
        assert r.stdout.count("多项式乘法结果正确") == 4 and "多项式乘法结果错误" not in r.stdout
        n4, p4, _, _ = ntt_amd.read_case_bin(data, 4)
        assert ntt_amd.check_case_bin(data, 4, expected[4]) and (n4, p4) == (3001, 1337006139375617)
        assert (out / "4.outb").exists() and (out / "1.outb").exists()
    # a wrong expected output flips the verdict and the exit status
    (data / "3.in").write_text((data / "0.in").read_text())
    (data / "3.out").write_text("1 2 3 4 5 6 7 ")
    r = subprocess.run(cmd[:-8] + ["--begin", "3", "--end", "3", "--epochs", "1", "--backend", backend],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 1 and "多项式乘法结果错误" in r.stdout


@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_flag_on_one_gpu(world):
    """bench.py called the way the driver calls it for N > 1 (no torchrun wrapper): it starts its ranks itself;
    over gloo they share this box's GPU.  One JSON line from rank 0 with n_gpus == world; every rank has checked rows of
    its own output before timing, rank 0 one row against the oracle."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--batch", "96", "--sustained-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["batch_per_gpu"] == 96 and j["roofline"]["kernel"] in ("row_mul", "col_fwd", "col_inv")
    # the other sharded workloads behind the headline loop: prime-per-rank CRT with its exchange, RNS limbs
    # per rank, one 2^26-point transform split over the two ranks -- each timed, each checked on every rank
    also = j["also"]
    assert also["checks"] == "ok", also
    for key in ("crt4_n2p17_ms", "crt4_GBps", "rns_32x256_n2p16_ms", "rns_products_per_s", "ntt2p26_fwd_inv_ms", "ntt2p26_GBps"):
        assert also[key] > 0, key
    # the rank-spanning CRT product runs at n = 2^17 here (4 primes over 2 or 4 ranks), against the oracle on rank 0
    assert also["crt4_primes_per_rank"] == 4 // world and also["rns_limbs_per_rank"] == 32 // world
    sl = -(-(2 * 131072 - 1) // world)
    assert also["crt4_sent_bytes_per_rank"] == (world - 1) * sl * (4 * (4 // world) + 8)
    assert also["ntt2p26_sent_bytes_per_rank"] == 4 * (world - 1) * ((1 << 26) // world // world) * 4
    assert j["roofline"]["binding"] == "valu-int" and 0 < j["roofline"]["valu_frac_of_coupled_peak"] < 1.2
    # ... and the in-library multi-device path (one process over all slots), run by rank 0 as a child process
    assert also.get("mdev_checks") == "ok", also
    assert also["mdev_devices"] == ",".join(["0"] * world) and also["mdev_crt4_n2p17_ms"] > 0
    assert also["mdev_ntt2p26_p2p_fwd_inv_ms"] > 0 and also["mdev_ntt2p26_copy_fwd_inv_ms"] > 0
    # ... and the process-per-GPU communicator (ntt_comm), run as a child job of `world` ranks (host mode here: gloo)
    assert also.get("comm_checks") == "ok" and also["comm_kind"] == "host" and also["comm_world"] == world, also
    assert also["comm_crt4_n2p17_ms"] > 0 and also["comm_ntt2p26_fwd_inv_ms"] > 0
    assert also["mdev_batch_products_per_s"] > 0 and also["mdev_batch_per_slot"] == 512      # arrays per slot, one process


def test_bench_also_block_one_rank():
    """The default (one GPU) bench line carries the same block: CRT through CrtExchange at world 1, 8192 RNS products,
    the single-GPU 2^26 transform."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--batch", "96", "--sustained-seconds", "0", "--no-one-lane"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    also = j["also"]
    assert also["checks"] == "ok" and also["ntt2p26_sent_bytes_per_rank"] == 0 and also["crt4_sent_bytes_per_rank"] == 0, also
    assert also["rns_products_per_s"] > 1e5 and also["crt4_n2p17_ms"] < 5 and also["ntt2p26_fwd_inv_ms"] < 50


def test_bench_watchdog_keeps_the_headline_line():
    """A hung `also` block must not cost the line: with a watchdog far too short for the block, bench.py still prints
    the complete headline line (roofline included) with the entries that had finished and the reason."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--batch", "96", "--sustained-seconds", "0", "--no-one-lane", "--also-timeout", "0.05"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["value"] > 0 and j["n_gpus"] == 1 and j["roofline"]["frac"] > 0 and j["config"]["batch_per_gpu"] == 96
    assert "did not finish" in j["also"]["error"] and "checks" not in j["also"]
