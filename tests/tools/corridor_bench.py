#!/usr/bin/env python3
"""Measurement of the f-3 kernel (corridor generation / selection).  Memory-side work: every decomposition scans the
whole cloud once (24 B per point; later scans only touch surviving 64-point words), so the algorithmic bytes are
decompositions x P x 24 and the bound is L2/HBM bandwidth (the cloud is shared by the fleet and L2-resident).
The numpy oracle is timed beside it on a bounded sample.     python tests/tools/corridor_bench.py [B=4096] [P=20000] [grid_cell=0]"""
import json
import sys
import time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from forces_resilient_planner_amd import solver
from oracle import corridor_oracle as C, tube_oracle as T   # CPU-baseline leg only

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
N, F = 20, 64
rng = np.random.default_rng(0)
cloud = np.c_[rng.uniform(-3, 9, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
s = np.linspace(0, 5, N)
centre = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
cx = np.interp(cloud[:, 0], centre[:, 0], centre[:, 1]); cz = np.interp(cloud[:, 0], centre[:, 0], centre[:, 2])
cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.6]
ref = centre[None] + rng.normal(0, 0.03, (B, N, 3))
yaw = np.arctan2(np.gradient(centre[:, 1]), np.gradient(centre[:, 0]))[None] + rng.normal(0, 0.05, (B, N))
solver.lib()
dev = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dtype=dt)
z = np.zeros((B, N + 1, 17)); z[..., 3] = 7.3; z[:, :N, 8:11] = ref; z[:, :N, 16] = yaw
E = torch.empty((B, N, 3, 3), dtype=torch.float64, device="cuda:0")
solver.tube_batch_device(dev(z), E)
d_cloud, d_ref, d_yaw = dev(cloud), dev(ref), dev(yaw)
A = torch.zeros((B, N, F, 3), dtype=torch.float64, device="cuda:0"); b = torch.zeros((B, N, F), dtype=torch.float64, device="cuda:0")
nf = torch.zeros((B, N), dtype=torch.int32, device="cuda:0"); pi = torch.zeros((B, N), dtype=torch.int32, device="cuda:0")
cnt = torch.zeros((B,), dtype=torch.int32, device="cuda:0")
GRID = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # cell size of the uniform grid, 0 = scan the whole cloud
grid = solver.CloudGrid(d_cloud, GRID) if GRID > 0 else None
fn = lambda: solver.corridor_batch_device(d_cloud, d_ref, d_yaw, E, A, b, nf, pi, cnt, grid=grid)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
ndec = int(cnt.abs().sum().item()); rows = nf.cpu().numpy()
ns = min(B, 16)
Eh = E[:ns].cpu().numpy()
t0 = time.time()
for p in range(ns):
    C.corridor_one(ref[p], yaw[p], Eh[p], cloud)
tc = time.time() - t0
by = ndec * len(cloud) * 24
print(json.dumps({"B": B, "N": N, "cloud_points": len(cloud), "grid_cell": GRID, "seconds": t, "planners_per_s": B / t,
                  "decompositions": ndec, "decompositions_per_planner": ndec / B, "mean_rows": float(rows[rows > 0].mean()),
                  "max_rows": int(rows.max()), "overflowed_planners": int((cnt < 0).sum().item()),
                  "algorithmic_bytes_first_scan": by, "GBps_first_scan": by / t / 1e9,
                  "cpu_oracle": {"planners_per_s": ns / tc, "sample": f"{ns} planners, numpy, 1 core"}}))
