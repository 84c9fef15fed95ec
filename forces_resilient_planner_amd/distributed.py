"""Multi-GPU sharding of a batch of independent NMPC problems (one process per GPU).

The path partitions by problem: contiguous batch ranges, no exchange during the solve.  The only
communication is the trivial split / gather around it:

  * ``scatter_batch`` / ``gather_batch``: rank 0 owns the full batch; every other rank receives / returns its contiguous
    shard by point-to-point transfers issued as ONE group (``batch_isend_irecv`` = ncclGroupStart / ncclSend /
    ncclRecv / ncclGroupEnd on the GPU box, so the seven xGMI links of the root carry the seven shards in parallel);
  * ``broadcast_nominal`` + ``monte_carlo_fext``: the Monte-Carlo workload (BASELINE configs[4]) ships ONE nominal
    problem and every rank draws the external-force samples of its own shard on its device (counter-based, so a sample
    depends on its global index only, not on the sharding);
  * ``summary_stats``: a 3-scalar reduction for reporting.

The tensors are torch tensors on the rank's device: HBM with backend "nccl" (= RCCL over xGMI) on the GPU box, host
memory with "gloo" in the CPU tests -- the same code path.
"""
from __future__ import annotations

import numpy as np


def shard_range(B: int, rank: int, world: int):
    """Contiguous range [lo, hi) of problems owned by `rank` (ceil(B / world) per rank)."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def shard_workload(w: dict, rank: int, world: int) -> dict:
    B = int(w["xinit"].shape[0])
    lo, hi = shard_range(B, rank, world)
    out = {}
    for k, v in w.items():
        out[k] = v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v
    out["B"] = hi - lo
    return out


def scatter_batch(full, shard, B: int, dist, src: int = 0):
    """full: list of tensors [B, ...] (meaningful on `src` only, may be None elsewhere); shard: list of tensors
    [hi - lo, ...] of this rank (allocated by the caller, filled here).  One grouped P2P exchange for all tensors."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ops = []
    if rank == src:
        for r in range(world):
            lo, hi = shard_range(B, r, world)
            for f, s in zip(full, shard):
                if r == src:
                    s.copy_(f[lo:hi])
                elif hi > lo:
                    ops.append(dist.P2POp(dist.isend, f[lo:hi].contiguous(), r))
    else:
        lo, hi = shard_range(B, rank, world)
        if hi > lo:
            for s in shard:
                ops.append(dist.P2POp(dist.irecv, s, src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def gather_batch(shard, full, B: int, dist, dst: int = 0):
    """The reverse of scatter_batch: every rank's shard tensors end up in full[lo:hi] on `dst`."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ops = []
    if rank == dst:
        for r in range(world):
            lo, hi = shard_range(B, r, world)
            for f, s in zip(full, shard):
                if r == dst:
                    f[lo:hi].copy_(s)
                elif hi > lo:
                    ops.append(dist.P2POp(dist.irecv, f[lo:hi], r))  # a leading-dimension slice is contiguous
    else:
        lo, hi = shard_range(B, rank, world)
        if hi > lo:
            for s in shard:
                ops.append(dist.P2POp(dist.isend, s, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def chunk_bounds(n: int, chunks: int):
    """Cut a shard of n problems into `chunks` pieces for the overlapped strong-scaling step: a small first piece (n / 8, so the
    first solve starts early) and equal ones behind it.  Returns the boundaries [0, b1, ..., n] (empty pieces dropped)."""
    if chunks <= 1 or n <= 1:
        return [0, n]
    first = max(1, n // 8)
    rest = n - first
    k = chunks - 1
    b = [0, first] + [first + (rest * (i + 1)) // k for i in range(k)]
    out = [0]
    for x in b[1:]:
        if x > out[-1]:
            out.append(x)
    return out


def _exchange(full, part, B: int, dist, root: int, to_root: bool, piece, chunks: int):
    """One grouped P2P exchange of piece `piece` of every rank's shard: root -> ranks (scatter) or ranks -> root (gather).
    full: tensors [B, ...] on the root; part: this rank's shard tensors [hi - lo, ...]."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ops = []
    if rank == root:
        for r in range(world):
            lo, hi = shard_range(B, r, world)
            cb = chunk_bounds(hi - lo, chunks)
            if piece + 1 >= len(cb):
                continue
            a, b = cb[piece], cb[piece + 1]
            if b <= a:
                continue
            for f, s in zip(full, part):
                if r == root:
                    if to_root: f[lo + a:lo + b].copy_(s[a:b])
                    else: s[a:b].copy_(f[lo + a:lo + b])
                elif to_root:
                    ops.append(dist.P2POp(dist.irecv, f[lo + a:lo + b], r))  # a leading-dimension slice is contiguous
                else:
                    ops.append(dist.P2POp(dist.isend, f[lo + a:lo + b], r))
    else:
        lo, hi = shard_range(B, rank, world)
        cb = chunk_bounds(hi - lo, chunks)
        if piece + 1 < len(cb):
            a, b = cb[piece], cb[piece + 1]
            for s in part:
                if b > a:
                    ops.append(dist.P2POp(dist.isend if to_root else dist.irecv, s[a:b], root))
    return dist.batch_isend_irecv(ops) if ops else []


def strong_step_overlapped(full_in, shard_in, shard_out, full_out, B: int, dist, solve_range, chunks: int = 4,
                           comm_stream=None, compute_stream=None, root: int = 0):
    """SURVEY 8e's scatter -> solve -> gather with the transfers streamed UNDER the solve: every rank's shard
    is cut into pieces (chunk_bounds); piece c + 1 is in flight from the root while piece c is being solved, and the plans of
    piece c travel back while piece c + 1 is being solved -- the way HostPipe overlaps PCIe for the host-buffer API.
    solve_range(a, b, c) solves the local problems [a, b) = piece c (asynchronous on the current stream on a GPU, synchronous in the gloo tests).
    On a GPU pass two torch streams: the exchanges are enqueued on comm_stream, the solves on compute_stream, ordered by events; the
    caller's current stream waits for both at the end.  Without streams (gloo / CPU tensors) the same schedule runs in order, so the
    results are those of the one-piece step by construction -- and by test (tests/test_distributed_cpu.py).
    Returns the number of pieces this rank solved."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(B, rank, world)
    n_max = max(shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world))
    pieces = len(chunk_bounds(n_max, chunks)) - 1   # every rank walks the same number of exchanges (grouped P2P is collective-like)
    cb = chunk_bounds(hi - lo, chunks)
    gpu = comm_stream is not None and compute_stream is not None
    # compute_stream may be a list: piece c is solved on stream c mod len, so the few long solves that end one piece's launch overlap
    # the head of the next (solve_range then gets the piece number and must give concurrent pieces their own queue workspace)
    comp = list(compute_stream) if isinstance(compute_stream, (list, tuple)) else [compute_stream]
    if gpu:
        import torch
        cur = torch.cuda.current_stream()
        comm_stream.wait_stream(cur)
        for cs in comp:
            cs.wait_stream(cur)
        arrived = [torch.cuda.Event() for _ in range(pieces)]
        solved = [torch.cuda.Event() for _ in range(pieces)]

    def scatter(c):
        if gpu:
            import torch
            with torch.cuda.stream(comm_stream):
                for req in _exchange(full_in, shard_in, B, dist, root, False, c, chunks):
                    req.wait()
                arrived[c].record(comm_stream)
        else:
            for req in _exchange(full_in, shard_in, B, dist, root, False, c, chunks):
                req.wait()

    def gather(c):
        if gpu:
            import torch
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(solved[c])
                for req in _exchange(full_out, shard_out, B, dist, root, True, c, chunks):
                    req.wait()
        else:
            for req in _exchange(full_out, shard_out, B, dist, root, True, c, chunks):
                req.wait()

    done = 0
    scatter(0)
    for c in range(pieces):
        if c + 1 < pieces:
            scatter(c + 1)             # in flight while piece c is solved
        if c + 1 < len(cb):
            if gpu:
                import torch
                cs = comp[c % len(comp)]
                with torch.cuda.stream(cs):
                    cs.wait_event(arrived[c])
                    solve_range(cb[c], cb[c + 1], c)
                    solved[c].record(cs)
            else:
                solve_range(cb[c], cb[c + 1], c)
            done += 1
        elif gpu:
            solved[c].record(comp[c % len(comp)])
        gather(c)                      # enqueued behind the solve of piece c; overlaps the solve of piece c + 1
    if gpu:
        cur.wait_stream(comm_stream)
        for cs in comp:
            cur.wait_stream(cs)
    return done


def broadcast_nominal(tensors, dist, src: int = 0):
    """One nominal problem (a handful of small tensors) to every rank."""
    for t in tensors:
        dist.broadcast(t, src)


def _splitmix64(x):
    """Counter-based 64-bit mixer (Steele, Lea & Flood's SplitMix64 finaliser) on int64 tensors; wrap-around arithmetic."""
    import torch
    x = x + torch.tensor(-7046029254386353131, dtype=torch.int64, device=x.device)             # 0x9E3779B97F4A7C15
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * torch.tensor(-4658895280553007687, dtype=torch.int64, device=x.device)  # 0xBF58476D1CE4E5B9
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * torch.tensor(-7723592293110705685, dtype=torch.int64, device=x.device)  # 0x94D049BB133111EB
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def monte_carlo_fext(fbar, sigma: float, lo: int, hi: int, seed: int, device):
    """f_ext samples [hi - lo, 3] ~ N(fbar, sigma^2 I) of the problems lo..hi-1 of a Monte-Carlo batch, generated on
    `device`.  Sample i depends on (seed, i) only: Box-Muller on two SplitMix64 streams indexed by the global problem
    number, so any sharding of the batch reproduces the single-process batch bit for bit."""
    import torch
    idx = torch.arange(lo, hi, dtype=torch.int64, device=device)[:, None] * 8 + torch.arange(3, dtype=torch.int64, device=device)[None, :]
    base = torch.tensor(int(seed) * 2654435761 % (1 << 62), dtype=torch.int64, device=device)
    a = _splitmix64(idx * 2 + base)
    b = _splitmix64(idx * 2 + 1 + base)
    u1 = (((a >> 11) & ((1 << 53) - 1)).to(torch.float64) + 1.0) / 9007199254740993.0   # (0, 1)
    u2 = ((b >> 11) & ((1 << 53) - 1)).to(torch.float64) / 9007199254740992.0           # [0, 1)
    n = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)
    return torch.as_tensor(fbar, dtype=torch.float64, device=device)[None, :] + sigma * n


def gather_solutions(z_local, flag_local, iters_local, B: int, dist, device="cpu"):
    """Gather per-rank outputs to every rank (all_gather of equal-size padded shards)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (B + world - 1) // world
    N = z_local.shape[1]
    zt = torch.zeros((per, N, 17), dtype=torch.float64, device=device)
    ft = torch.full((per,), -999, dtype=torch.int32, device=device)
    it = torch.zeros((per,), dtype=torch.int32, device=device)
    n = z_local.shape[0]
    zt[:n] = torch.as_tensor(z_local, device=device)
    ft[:n] = torch.as_tensor(flag_local, device=device)
    it[:n] = torch.as_tensor(iters_local, device=device)
    zs = [torch.empty_like(zt) for _ in range(world)]
    fs = [torch.empty_like(ft) for _ in range(world)]
    its = [torch.empty_like(it) for _ in range(world)]
    dist.all_gather(zs, zt); dist.all_gather(fs, ft); dist.all_gather(its, it)
    z = torch.cat(zs)[:B].cpu().numpy(); f = torch.cat(fs)[:B].cpu().numpy(); i = torch.cat(its)[:B].cpu().numpy()
    return z, f, i


def summary_stats(flag_local, iters_local, dist, device="cpu"):
    """(converged count, iteration sum, problem count) summed over ranks: the only reduction there is."""
    import torch
    t = torch.tensor([float((flag_local == 1).sum()), float(iters_local.sum()), float(len(flag_local))],
                     dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
