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


def crt_multiply_sharded(group, residue_fn, merge_fn, n, device="cpu"):
    """Large-modulus product with the four CRT primes sharded over the ranks of `group`
    (main-mpi.cc:657-660: rank r owns primes partition(4, world, r)) and the ONE exchange step the
    path has: instead of the reference's gather-to-root (main-mpi.cc:772-791, after which rank 0
    alone merges, :793-796) every rank receives coefficient slice r of all four residue polynomials
    (all-to-all over RCCL/xGMI, 4 B per coefficient and prime), merges its slice, and the u64 slices
    are all-gathered.

    residue_fn(k) -> int32 torch tensor [2n-1] (u32 bit pattern) with the product mod prime k;
    merge_fn(rows, count) -> int64 torch tensor [count] from four int32 tensors [count].
    Returns an int64 torch tensor [2n-1] (u64 bit pattern), identical on every rank.
    The callbacks must hand back tensors whose producing work is complete with respect to torch's
    current stream (the library runs on its own stream: synchronise the context, or give it torch's
    stream with Context.set_stream).
    """
    import torch
    world, rank = group.world, group.rank
    out_len = 2 * n - 1
    sl = -(-out_len // world)                       # coefficients per rank (last slices zero padded)
    maxk = -(-4 // world)                           # primes per rank, main-mpi.cc:657
    start, count = partition(4, world, rank)
    send = torch.zeros((world, maxk, sl), dtype=torch.int32, device=device)
    for j in range(count):
        r = residue_fn(start + j)
        padded = torch.zeros(world * sl, dtype=torch.int32, device=device)
        padded[:out_len] = r
        send[:, j, :] = padded.view(world, sl)
    if world > 1:
        recv = torch.empty_like(send)
        group.dist.all_to_all_single(recv.view(-1), send.view(-1))
    else:
        recv = send
    rows = []
    for k in range(4):                              # prime k came from rank k // maxk, slot k % maxk
        rows.append(recv[k // maxk, k % maxk, :].contiguous())
    mine = merge_fn(rows, sl)
    if world > 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        group.dist.all_gather(parts, mine)
        full = torch.cat(parts)
    else:
        full = mine
    return full[:out_len]
