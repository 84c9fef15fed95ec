"""World-size-2 gloo test of the multi-GPU path's host logic (sharding, gather, summary reduction).
The per-rank solve is the CPU oracle here (test infrastructure); on the GPU box each rank runs the HIP
solver on its own device instead -- the partition/gather code is the same."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from forces_resilient_planner_amd import distributed as D
from forces_resilient_planner_amd import workloads

from . import oracle_lib as OL


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workloads.config2(B)
    ws = D.shard_workload(w, rank, world)
    z, fl, info = OL.solve_batch(ws, nthreads=2)
    it = np.array([i.it for i in info], dtype=np.int32)
    zg, fg, ig = D.gather_solutions(z, fl, it, B, dist)
    stats = D.summary_stats(fl, it, dist)
    if rank == 0:
        outq.put((zg, fg, ig, stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 7])
def test_two_rank_shard_solve_gather_matches_single_process(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    zg, fg, ig, stats = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = workloads.config2(B)
    z, fl, info = OL.solve_batch(w)
    assert np.array_equal(fg, fl)
    assert np.array_equal(zg, z)          # same code, same inputs per problem -> identical
    assert stats[2] == B and stats[0] == (fl == 1).sum()


def test_shard_ranges_cover_batch_exactly():
    for B in (1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi = D.shard_range(B, r, world)
                got += list(range(lo, hi))
            assert got == list(range(B))
