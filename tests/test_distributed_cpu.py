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


def _sg_worker(rank, world, port, B, outq):
    """rank 0 owns the full batch; scatter the shards (grouped P2P), 'solve' with the oracle, gather back to rank 0:
    the exact code path bench.py --scaling strong runs on HBM tensors over RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workloads.config2(B) if rank == 0 else None
    N, npar = 20, 130
    lo, hi = D.shard_range(B, rank, world)
    f64 = torch.float64
    full = [torch.from_numpy(w["xinit"]), torch.from_numpy(w["x0"]), torch.from_numpy(w["params"]),
            torch.from_numpy(w["nfaces"].astype(np.int32))] if rank == 0 else [None] * 4
    shard = [torch.zeros((hi - lo, 9), dtype=f64), torch.zeros((hi - lo, N, 17), dtype=f64),
             torch.zeros((hi - lo, N, npar), dtype=f64), torch.zeros((hi - lo, N), dtype=torch.int32)]
    D.scatter_batch(full, shard, B, dist)
    ws = dict(xinit=shard[0].numpy(), x0=shard[1].numpy(), params=shard[2].numpy(), nfaces=shard[3].numpy(), N=N, M=30, model=0)
    if hi > lo:
        z, fl, info = OL.solve_batch(ws, nthreads=2)
        it = np.array([i.it for i in info], dtype=np.int32)
    else:
        z = np.zeros((0, N, 17)); fl = np.zeros(0, np.int32); it = np.zeros(0, np.int32)
    out_full = [torch.zeros((B, N, 17), dtype=f64), torch.zeros((B,), dtype=torch.int32), torch.zeros((B,), dtype=torch.int32)] if rank == 0 else [None] * 3
    D.gather_batch([torch.from_numpy(np.ascontiguousarray(z)), torch.from_numpy(fl.astype(np.int32)), torch.from_numpy(it)], out_full, B, dist)
    if rank == 0:
        outq.put(tuple(t.numpy() for t in out_full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,world", [(9, 2), (4, 3), (2, 3)])
def test_scatter_solve_gather_from_rank0_matches_single_process(B, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sg_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    zg, fg, ig = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = workloads.config2(B)
    z, fl, info = OL.solve_batch(w)
    assert np.array_equal(fg, fl) and np.array_equal(zg, z)
    assert np.array_equal(ig, np.array([i.it for i in info]))


def _overlap_worker(rank, world, port, B, chunks, outq):
    """The overlapped strong-scaling step (pieces of every shard scattered / solved / gathered in turn) with the oracle as the
    per-piece solve: what bench.py --scaling strong runs on HBM tensors over RCCL with two streams."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = workloads.config2(B) if rank == 0 else None
    N, npar = 20, 130
    lo, hi = D.shard_range(B, rank, world)
    f64 = torch.float64
    full = [torch.from_numpy(w["xinit"]), torch.from_numpy(w["x0"]), torch.from_numpy(w["params"]),
            torch.from_numpy(w["nfaces"].astype(np.int32))] if rank == 0 else [None] * 4
    shard = [torch.zeros((hi - lo, 9), dtype=f64), torch.zeros((hi - lo, N, 17), dtype=f64),
             torch.zeros((hi - lo, N, npar), dtype=f64), torch.zeros((hi - lo, N), dtype=torch.int32)]
    out = [torch.full((hi - lo, N, 17), -7.0, dtype=f64), torch.full((hi - lo,), -99, dtype=torch.int32), torch.full((hi - lo,), -99, dtype=torch.int32)]
    out_full = [torch.zeros((B, N, 17), dtype=f64), torch.zeros((B,), dtype=torch.int32), torch.zeros((B,), dtype=torch.int32)] if rank == 0 else [None] * 3
    ranges = []

    def solve_range(a, b, c):
        ranges.append((a, b))
        ws = dict(xinit=shard[0][a:b].numpy(), x0=shard[1][a:b].numpy(), params=shard[2][a:b].numpy(), nfaces=shard[3][a:b].numpy(), N=N, M=30, model=0)
        z, fl, info = OL.solve_batch(ws, nthreads=2)
        out[0][a:b] = torch.from_numpy(z); out[1][a:b] = torch.from_numpy(fl.astype(np.int32))
        out[2][a:b] = torch.from_numpy(np.array([i.it for i in info], dtype=np.int32))

    D.strong_step_overlapped(full, shard, out, out_full, B, dist, solve_range, chunks=chunks)
    assert ranges == list(zip(D.chunk_bounds(hi - lo, chunks)[:-1], D.chunk_bounds(hi - lo, chunks)[1:])) or hi == lo
    if rank == 0:
        outq.put(tuple(t.numpy() for t in out_full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,world,chunks", [(21, 2, 4), (10, 3, 3), (2, 3, 4), (17, 2, 1)])
def test_overlapped_strong_step_equals_the_unchunked_one(B, world, chunks):
    """Chunked == unchunked: the pieces of every shard cover it exactly once, in order, and the gathered
    plans / flags / iteration counts are those of the single-process solve of the whole batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, B, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    zg, fg, ig = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = workloads.config2(B)
    z, fl, info = OL.solve_batch(w)
    assert np.array_equal(fg, fl) and np.array_equal(zg, z)
    assert np.array_equal(ig, np.array([i.it for i in info]))


def test_chunk_bounds_partition_a_shard():
    for n in (0, 1, 2, 7, 511, 512, 2048, 16384):
        for c in (1, 2, 4, 8):
            b = D.chunk_bounds(n, c)
            assert b[0] == 0 and b[-1] == n and all(x < y for x, y in zip(b[:-1], b[1:])) or n == 0
            assert len(b) - 1 <= max(1, c)


def test_monte_carlo_samples_do_not_depend_on_the_sharding():
    fbar = np.array([0.3, -1.2, 0.8])
    full = D.monte_carlo_fext(fbar, 0.5, 0, 4099, 7, "cpu").numpy()
    for world in (2, 8):
        parts = [D.monte_carlo_fext(fbar, 0.5, *D.shard_range(4099, r, world), 7, "cpu").numpy() for r in range(world)]
        assert np.array_equal(np.concatenate(parts), full)
    assert np.all(np.isfinite(full))
    big = D.monte_carlo_fext(fbar, 0.5, 0, 200000, 11, "cpu").numpy()
    assert np.max(np.abs(big.mean(0) - fbar)) < 0.01 and np.max(np.abs(big.std(0) - 0.5)) < 0.01
    assert abs(np.corrcoef(big[:, 0], big[:, 1])[0, 1]) < 0.01 and abs(np.corrcoef(big[:-1, 0], big[1:, 0])[0, 1]) < 0.01
    assert not np.array_equal(D.monte_carlo_fext(fbar, 0.5, 0, 16, 8, "cpu").numpy(), full[:16])


def test_shard_ranges_cover_batch_exactly():
    for B in (1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                lo, hi = D.shard_range(B, r, world)
                got += list(range(lo, hi))
            assert got == list(range(B))


def test_bench_gpus_n_relaunches_itself_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher around it re-executes under torch.distributed.run with N ranks on
    127.0.0.1, keeping its own arguments (both --scaling modes); with WORLD_SIZE set (the driver's torchrun form) it does not."""
    import sys
    import bench
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "1", "--scaling", "strong"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1", "--scaling", "strong"]
    # main() takes that path exactly when --gpus > 1 and no launcher set WORLD_SIZE
    calls = []
    monkeypatch.setattr(os, "execv", lambda exe, argv: (calls.append(argv), (_ for _ in ()).throw(SystemExit(0))))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--scaling", "weak"])
    with pytest.raises(SystemExit):
        bench.main()
    assert len(calls) == 1 and "--nproc-per-node=4" in calls[0] and calls[0][-4:] == ["--gpus", "4", "--scaling", "weak"]
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")  # under a launcher: no re-exec (here it then stops at "needs an MI355X")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not calls and "MI355X" in str(e.value)


def test_bench_dry_run_plans_eight_ranks_without_a_device():
    """what `bench.py --gpus 8` WOULD do, asserted without a GPU: the shards tile the batch, every rank gets the same
    share (weak) or a contiguous shard (strong), the piece count follows the logged policy, and the pieces tile their shard."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def plan(*a):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", *a], capture_output=True, text=True, check=True)
        return json.loads(r.stdout.strip().splitlines()[-1])
    p = plan("--gpus", "8")
    assert p["world_size"] == 8 and p["batch_total"] == 8 * 4096 and [r["problems"] for r in p["ranks"]] == [[4096 * i, 4096 * (i + 1)] for i in range(8)]
    assert all(r["pieces"] == 1 for r in p["ranks"])
    for cfg, total in (("2", 4096), ("3", 16384)):
        p = plan("--gpus", "8", "--scaling", "strong", "--config", cfg)
        assert p["batch_total"] == total
        edges = [r["problems"] for r in p["ranks"]]
        assert edges[0][0] == 0 and edges[-1][1] == total and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        for r in p["ranks"]:
            pr = r["piece_ranges"]
            assert pr[0][0] == r["problems"][0] and pr[-1][1] == r["problems"][1] and all(a[1] == b[0] for a, b in zip(pr, pr[1:])) and len(pr) <= r["pieces"]
            assert ("two rounds" in r["why"]) and r["pieces"] in (1, 2)
    # configs[2] strong at 8 ranks: shards of 512 are below two rounds of resident workgroups -> one piece; configs[3]: 2048 >= 2 * 512 -> two
    assert plan("--gpus", "8", "--scaling", "strong", "--config", "2")["ranks"][0]["pieces"] == 1
    assert plan("--gpus", "8", "--scaling", "strong", "--config", "3")["ranks"][0]["pieces"] == 2
