#!/usr/bin/env python3
"""BASELINE configs[3] alone (B = 16384, N = 30, <= 15 faces): throughput for A/B runs of kernel variants / FRP_RESIDENT_SLOTS."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forces_resilient_planner_amd import solver, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
w = workloads.config3(B)
MF = int(w["nfaces"].max())
lanes = []
for i in range(2):
    ds = solver.DeviceSolver(B, w["N"], w["M"], MF, w["model"]); ds.upload(w)
    lanes.append((ds, torch.cuda.Stream() if i else torch.cuda.current_stream()))
for i in range(2): lanes[i][0].solve(lanes[i][1])
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 6
for i in range(steps): lanes[i % 2][0].solve(lanes[i % 2][1])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl = lanes[0][0].exitflag.cpu().numpy(); it = lanes[0][0].iters.cpu().numpy()
print(json.dumps({"B": B, "solves_per_s": B * steps / dt, "ms_per_batch": dt / steps * 1e3, "converged_frac": float((fl == 1).mean()),
                  "mean_iters": float(it.mean()), "slots_env": os.environ.get("FRP_RESIDENT_SLOTS")}))
