#!/usr/bin/env python3
"""GPU box: BASELINE configs[4] tick by tick -- solver-kernel time of every tick (HIP events around DeviceSolver.solve on its stream)
beside THAT tick's iteration statistics and Gauss-Newton redos (bench.py reports the statistics of the last tick only, the kernel time
averaged over all).  Also: the same batch size of configs[2] (every problem different) and of ONE configs[2] problem repeated B times
(every slot in lockstep, as the Monte-Carlo fleet nearly is).

    python tools/r06/config4_ticks.py [B] [ticks] > gpurun_out/r06/config4_ticks.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import distributed as D, layout as L, solver, workloads  # noqa: E402
from forces_resilient_planner_amd.workloads import _bbox_faces, _weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0"); f64 = dict(dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev)


def timed(fn):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(stream); fn(); b.record(stream); b.synchronize()
    return a.elapsed_time(b)


def line(tag, ms, ds):
    it = ds.iters[:B].cpu().numpy(); redo = ds.info[:B, 7].cpu().numpy(); fl = ds.exitflag[:B].cpu().numpy()
    h = np.bincount(it, minlength=8)[:8]
    print(f"{tag:28s} solve {ms:7.3f} ms  its mean {it.mean():.3f} max {it.max():3d} hist(0..7) {h.tolist()}  redos/solve {redo.mean():.4f}  flag==1 {np.mean(fl == 1):.4f}"
          f"  per solve per slot {ms * 1e3 / (B / 1024):6.1f} us = {ms * 1e3 / (B / 1024) / max(it.mean(), 1e-9):5.1f} us/iteration", flush=True)


# configs[4]
w1 = workloads.config4_nominal(1, ticks=T + 2)
N, M, model = w1["N"], w1["M"], w1["model"]
fleet = solver.DeviceFleet(B, N, M, 6, model, _weights(model), "cuda:0")
ds = fleet.solver
fleet.mpc_output.copy_(torch.from_numpy(np.ascontiguousarray(w1["mpc_output"])).to(dev).expand(B, N + 1, L.NZ))
fleet.ellipsoid.copy_(torch.from_numpy(np.ascontiguousarray(w1["E"])).to(dev).expand(B, N, 3, 3))
fleet.poly_nfaces.fill_(6)
fbar = w1["f_ext"].mean(0)
f_ext = D.monte_carlo_fext(fbar, 0.5, 0, B, workloads.SEED0 + 5, dev)
yaw = float(w1["heading"][0]); ref_yaw = torch.full((B, N), yaw, **f64)
for t in range(T):
    r1 = w1["ref_long"][:, t:t + N]
    A1, b1 = _bbox_faces(r1, np.full((1, N), yaw))
    fleet.poly_A.copy_(torch.from_numpy(A1).to(dev).expand(B, N, 6, 3)); fleet.poly_b.copy_(torch.from_numpy(b1).to(dev).expand(B, N, 6))
    if t > 0:
        fleet.coldstart(thrust=7.3)
    fleet.pack(f_ext, torch.from_numpy(r1).to(dev).expand(B, N, 3).contiguous(), ref_yaw)
    ms = timed(lambda: ds.solve(stream))
    line(f"configs[4] tick {t}", ms, ds)
    fleet.update()
del fleet, ds
torch.cuda.empty_cache()

# configs[2] at the same batch: all different, then one problem B times
w = workloads.config2(B, seed=workloads.SEED0 + 3)
for tag, pick in (("configs[2] all different", None), ("configs[2] problem 0 x B", 0), ("configs[2] problem 1 x B", 1)):
    ww = dict(w)
    if pick is not None:
        for k, v in w.items():
            if isinstance(v, np.ndarray) and v.shape[:1] == (B,):
                ww[k] = np.ascontiguousarray(np.broadcast_to(v[pick], v.shape))
    ds = solver.DeviceSolver(B, w["N"], w["M"], 6, w["model"], "cuda:0")
    ds.upload(ww)
    for r in range(3):
        ms = timed(lambda: ds.solve(stream))
    line(tag, ms, ds)
    del ds
