#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one GPU (device-resident, 2 streams like bench.py): configs[1] (no corridor),
configs[3] (N = 30, <= 15 faces, per-stage f_ext; infeasible problems exit with flag -7)."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from forces_resilient_planner_amd import solver, workloads

def run(name, w, steps=10):
    B = w["B"]
    MF = int(w["nfaces"].max()) if w["nfaces"].size else 0
    lanes = []
    for i in range(2):
        ds = solver.DeviceSolver(B, w["N"], w["M"], max(MF, 1), w["model"]); ds.upload(w)
        lanes.append((ds, torch.cuda.Stream() if i else torch.cuda.current_stream()))
    for i in range(2): lanes[i][0].solve(lanes[i][1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): lanes[i % 2][0].solve(lanes[i % 2][1])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = lanes[0][0].exitflag.cpu().numpy(); it = lanes[0][0].iters.cpu().numpy()
    print(json.dumps({"config": name, "B": B, "N": w["N"], "solves_per_s": B * steps / dt, "ms_per_batch": dt / steps * 1e3,
                      "converged_frac": float((fl == 1).mean()), "mean_iters": float(it.mean()), "max_iters": int(it.max())}))

run("configs[1] no corridor", workloads.config1(4096))
run("configs[3] N=30 <=15 faces", workloads.config3(16384))
run("configs[2] final model", workloads.config2(4096, model=1))
