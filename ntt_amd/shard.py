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

    def host_signal(self, key):
        """Rank 0 announces `key` through the rendezvous store; see host_wait."""
        if self.dist is not None:
            self.dist.distributed_c10d._get_default_store().set(key, "1")

    def host_wait(self, key, seconds=600.0):
        """Block on the HOST until rank 0 has announced `key` (no collective, so no kernel spins on this rank's GPU
        meanwhile -- an RCCL barrier would keep one there for as long as rank 0 takes)."""
        if self.dist is not None:
            import datetime
            self.dist.distributed_c10d._get_default_store().wait([key], datetime.timedelta(seconds=seconds))

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
    run on the same stream (Context.use_torch_stream(), as
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


class CrtExchange:
    """The sharded large-modulus product of crt_multiply_sharded for REPEATED calls at one size, with every
    buffer preallocated and the whole epoch queued on one stream (reference: the epoch loop of
    main-mpi.cc:706-802 with its gather-to-root :772-791 and merge-on-root :793-796).

    * rank r owns primes partition(4, world, r) (main-mpi.cc:657-660); ``ntt_crt_residue_multiply`` writes the
      residue product of owned prime slot j straight into ``send[j]`` -- a [world][sl] buffer whose row h is
      the coefficient slice rank h merges, i.e. the product lands in exchange order (its zero tail padding is
      written once, here);
    * one ``all_to_all_single`` per prime slot (RCCL on device tensors) brings slice `rank` of every prime;
      ``ntt_crt_merge`` turns the four slices into this rank's u64 slice; an all-gather into views of
      ``full`` assembles the product on every rank.

    With device tensors the context is switched to torch's current stream, so compute and collectives are
    ordered by the stream alone: no host synchronisation inside an epoch, no allocation after construction
    (``torch.cuda.memory_stats`` is flat across calls; tests/test_gpu_multiprocess.py).  The returned tensor
    is a view of ``full`` and is overwritten by the next call.  With host tensors (gloo rehearsal on a
    one-GPU box) the library stages through its own buffers and returns after its copies.
    """

    def __init__(self, group, ctx, n, p64, device="cpu"):
        import torch
        self.group, self.ctx, self.n, self.p64 = group, ctx, int(n), int(p64)
        self.device = torch.device(device)
        world = group.world
        self.out_len = 2 * self.n - 1
        self.sl = -(-self.out_len // world)
        self.maxk = -(-4 // world)
        self.start, self.count = partition(4, world, group.rank)
        kw = dict(dtype=torch.int32, device=self.device)
        self.send = [torch.zeros(world * self.sl, **kw) for _ in range(self.maxk)]
        self.recv = [torch.zeros((world, self.sl), **kw) for _ in range(self.maxk)] if world > 1 else \
                    [s.view(world, self.sl) for s in self.send]
        self.mine = torch.zeros(self.sl, dtype=torch.int64, device=self.device)
        self.full = torch.zeros(world * self.sl, dtype=torch.int64, device=self.device)
        self.parts = list(self.full.view(world, self.sl).unbind(0))
        if self.device.type == "cuda":
            ctx.use_torch_stream()

    def _arg(self, t):                                  # host tensors go in as numpy views (the library stages them)
        return t if self.device.type == "cuda" else t.numpy()

    def multiply(self, a, b):
        """a, b: int64 tensors [n] (u64 bit patterns) on `device`; returns int64 [2n-1], identical on every rank."""
        import numpy as np
        g, dev = self.group, self.device.type == "cuda"
        for j in range(self.count):
            out = self.send[j] if dev else self.send[j].numpy().view(np.uint32)
            self.ctx.crt_residue_multiply(a if dev else a.numpy().view(np.uint64), b if dev else b.numpy().view(np.uint64),
                                          self.start + j, out=out)
        if g.world > 1:
            for j in range(self.maxk):
                g.dist.all_to_all_single(self.recv[j].view(-1), self.send[j])
        rows = [self.recv[k % self.maxk][k // self.maxk] for k in range(4)]     # prime k: rank k // maxk, slot k % maxk
        self.ctx.crt_merge(rows if dev else [r.numpy().view(np.uint32) for r in rows], self.p64,
                           out=self.mine if dev else self.mine.numpy().view(np.uint64), count=self.sl)
        if g.world > 1:
            g.dist.all_gather(self.parts, self.mine)
            return self.full[:self.out_len]
        return self.mine[:self.out_len]


class DistExchange:
    """dist_transform for repeated calls: the two all-to-alls of one split transform through two preallocated
    ping-pong buffers (device tensors: RCCL on torch's current stream, which the context is switched to, so
    the column pass and the exchanges are ordered by the stream alone; host exchange over gloo: two pinned-size
    host mirrors).  ``forward`` leaves its input untouched; the result is a view of an internal buffer and is
    overwritten by the next call."""

    def __init__(self, group, ctx, dplan, on_device_exchange=True):
        import torch
        self.group, self.ctx, self.dp = group, ctx, dplan
        self.world, self.S = dplan.world, dplan.local_length
        self.buf = [torch.zeros(self.S, dtype=torch.int32, device="cuda") for _ in range(2)]
        self.host = None
        if not on_device_exchange and group.dist is not None:
            self.host = [torch.zeros(self.S, dtype=torch.int32) for _ in range(2)]
        ctx.use_torch_stream()

    def _a2a(self, src, dst):
        if self.group.dist is None:
            dst.copy_(src)
        elif self.host is None:
            self.group.dist.all_to_all_single(dst, src)
        else:
            self.host[0].copy_(src)                    # D2H on torch's stream, complete on return
            self.group.dist.all_to_all_single(self.host[1], self.host[0])
            dst.copy_(self.host[1])

    def forward(self, x_local):
        a, b = self.buf
        self._a2a(x_local, a)
        self.dp.colpass(a, inverse=False)
        self._a2a(a, b)
        self.dp.local(b, inverse=False)
        return b

    def inverse(self, X_local):
        a, b = self.buf
        a.copy_(X_local)
        self.dp.local(a, inverse=True)
        self._a2a(a, b)
        self.dp.colpass(b, inverse=True)
        self._a2a(b, a)
        return a
