"""Host-side sharding helpers for multi-GPU runs: one process per GPU, launched by
``torch.distributed.run``; the NTT hot path has no data-path collective when the units are
independent products (polynomial batch, RNS limbs, CRT primes), so ranks only meet at the
barrier and at the max-over-ranks of the measured time.

``partition`` mirrors the reference's MPI prime sharding (main-mpi.cc:657-660:
``mods_per_process = ceil(4 / size)``, rank r owns ``[r * mpp, min((r + 1) * mpp, 4))``),
generalised to any unit count.
"""
import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def partition(total_units, world, rank):
    """(start, count) of the contiguous block of units owned by `rank`: ceil(total/world) units per
    rank, the last ranks may own fewer or none (exactly the reference's rule for 4 primes)."""
    per = -(-total_units // world)
    start = min(rank * per, total_units)
    return start, max(0, min(per, total_units - start))


def rank_seed(base_seed, rank, units_per_rank):
    """Unit u of rank r uses seed base + r * units_per_rank + u: every rank multiplies different
    polynomials, and a 1-rank run of world*units units reproduces the same inputs."""
    return base_seed + rank * units_per_rank


class Group:
    """Thin wrapper over torch.distributed (backend "nccl" = RCCL on GPU boxes, "gloo" on CPU)."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local_rank = env_rank_world()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
