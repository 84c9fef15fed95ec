"""Multi-GPU sharding of a batch of independent NMPC problems (one process per GPU).

The path partitions by problem: contiguous batch ranges, no exchange during the solve.
The only communication is the trivial split/gather around it -- and a 3-scalar reduction for summary
statistics.  `backend` is "nccl" (= RCCL over xGMI) on the GPU box and "gloo" in the CPU tests.
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
