#!/usr/bin/env python3
"""Randomised agreement of the f-2 / f-3 / f-4 kernels with their oracles over many worlds (GPU box).
   python tests/tools/soak_corridor.py [seconds=120]
Per world: random cloud (300..20000 points, sometimes snapped to a voxel grid), random tunnel width, B = 8 planners;
references from a random kinodynamic path through frp_nmpc_reference_batch, tube from frp_nmpc_tube_batch on random
plans, corridor from frp_nmpc_corridor_batch -- each compared with its oracle on the same inputs."""
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver
from oracle import corridor_oracle as C, tube_oracle as T, reference_oracle as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
stats = dict(worlds=0, planners=0, decompositions=0, index_mismatch=0, rowcount_mismatch=0, row_order_only=0, row_value_mismatch=0,
             tube_max_rel=0.0, ref_max_abs=0.0, shrink_worlds=0, grid_worlds=0, row_order_only_in_shrink=0, row_order_only_in_grid=0)
seed = 0
lb, ub = L.bounds()
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    B, N, K = 8, 20, 90
    P = int(rng.choice([300, 2000, 6000, 20000]))
    cloud = np.c_[rng.uniform(-3, 9, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
    grid = rng.random() < 0.3
    if grid:
        cloud = np.round(cloud / 0.1) * 0.1; stats["grid_worlds"] += 1
    speed = rng.uniform(0.5, 2.5)
    tt = np.arange(K) * 0.05
    path = np.c_[speed * tt, rng.uniform(0.1, 0.6) * np.sin(rng.uniform(0.5, 1.5) * tt), 1.0 + 0.15 * np.cos(tt)]
    tunnel = rng.uniform(0.2, 0.9)
    cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > tunnel]
    z = lb + (ub - lb) * rng.random((B, N + 1, 17))
    z[..., 11:14] = rng.uniform(-5, 5, (B, N + 1, 3)); z[..., 16] = rng.uniform(-3.1, 3.1, (B, N + 1))
    z[:, 1, 8:11] = path[0] + rng.normal(0, 0.5, (B, 3))
    off = rng.uniform(0, 0.05 * 40, B)
    rp, ry, fl = solver.reference_batch_host(path, off, z)
    E = solver.tube_batch_host(z[:, :N])
    consts = dict(solver.CORRIDOR_DEFAULTS)
    shrink = rng.random() < 0.25
    if shrink:
        consts["seed_len"] = float(rng.uniform(0.5, 1.5)); stats["shrink_worlds"] += 1
    pi, A, b, nf, cnt = solver.corridor_batch_host(cloud, rp, ry, E, consts=consts)
    for p in range(B):
        po, yo, fo = R.references_one(path, K, off[p], z[p], N)
        stats["ref_max_abs"] = max(stats["ref_max_abs"], float(np.abs(rp[p] - po).max()), float(np.abs(ry[p] - yo).max()))
        Eo = T.tube_one(z[p, :N])
        tr = float(np.max(np.abs(E[p] - Eo) / (1e-3 + np.abs(Eo))))
        if tr > stats["tube_max_rel"]:
            stats["tube_max_rel"] = tr; stats["tube_max_rel_at"] = [seed, p]  # (world seed, planner: to re-run the worst case)
        idx, polys = C.corridor_one(rp[p], ry[p], E[p], cloud, bbox=consts['bbox'], seed_len=consts['seed_len'], inflation=consts['inflation'])
        stats["planners"] += 1; stats["decompositions"] += len(polys)
        if not np.array_equal(pi[p], idx):
            stats["index_mismatch"] += 1
            continue
        for k, (Ao, bo) in enumerate(polys):
            if nf[p, k] != len(bo):
                stats["rowcount_mismatch"] += 1
                continue
            G = np.c_[A[p, k, :len(bo)], b[p, k, :len(bo)]]; O = np.c_[Ao, bo]
            if np.max(np.abs(G - O)) < 1e-9:
                continue
            D = np.abs(G[:, None, :] - O[None, :, :]).max(axis=2)
            m = D.argmin(axis=0)
            if sorted(m) == list(range(len(bo))) and np.max(np.abs(G[m] - O)) < 1e-9:
                stats["row_order_only"] += 1
                stats["row_order_only_in_shrink"] += int(shrink); stats["row_order_only_in_grid"] += int(grid)
            else:
                stats["row_value_mismatch"] += 1
    stats["worlds"] += 1
print(json.dumps(stats))
