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
    The tensors handed to merge_fn come out of all_to_all_single / .contiguous() on torch's CURRENT
    stream, and what residue_fn returns is sent from that stream: with device tensors the library must
    run on the same stream (Context.set_stream(torch.cuda.current_stream().cuda_stream), as
    ntt_amd.launch does) or the callbacks must synchronise both ways themselves -- the context's own
    stream is not ordered against torch's.
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


def dist_transform(world, rank, all_to_all, colpass_fn, local_fn, x_local, inverse=False):
    """One transform of length L = world * S split over `world` ranks (BASELINE.json configs[4]).

    Forward: rank r holds x[r*S:(r+1)*S] (natural order) and ends with X[r + world*k2], k2 < S.
    Inverse: the reverse (input in that transposed layout, output natural blocks, scaled by 1/L).
    Two all-to-all exchanges, each moving (world-1)/world of the data, around the per-rank compute:
      colpass_fn(arr2d [world][S/world], inverse) -> same shape   (ntt_dist_colpass on GPU boxes)
      local_fn(row [S], inverse) -> [S]                           (ntt_dist_local)
    all_to_all(send [world][S/world]) -> recv [world][S/world] with recv[s] = what rank s sent to us
    (torch.distributed.all_to_all_single under the hood; injected so that tests can simulate ranks).
    """
    S = x_local.shape[0]
    cols = S // world
    if not inverse:
        held = all_to_all(x_local.reshape(world, cols))      # held[i1][j] = x[i1*S + rank*cols + j]
        held = colpass_fn(held, False)                       # held[k1][j] = Y[k1][rank*cols + j]
        row = all_to_all(held).reshape(S)                    # row[h*cols + j] = Y[rank][h*cols + j]
        return local_fn(row, False)
    row = local_fn(x_local, True)                            # D[rank][i2], i2 natural
    held = all_to_all(row.reshape(world, cols))              # held[k1][j] = D[k1][rank*cols + j]
    held = colpass_fn(held, True)                            # held[i1][j] = x[i1*S + rank*cols + j]
    return all_to_all(held).reshape(S)                       # x[rank*S + h*cols + j]


def torch_all_to_all(group):
    """all_to_all callable for dist_transform over a Group (RCCL on GPU tensors, gloo on CPU tensors)."""
    def a2a(send):
        import torch
        send = send.contiguous()
        if group.dist is None:
            return send
        recv = torch.empty_like(send)
        group.dist.all_to_all_single(recv.view(-1), send.view(-1))
        return recv
    return a2a
