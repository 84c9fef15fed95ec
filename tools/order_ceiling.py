#!/usr/bin/env python3
"""How much launch time is left in the queue order: configs[2] at B = 4096 with the shipped key, with the order a caller with
perfect knowledge would give (order_hint = the iteration counts of the same solve), and in index order."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forces_resilient_planner_amd import solver, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workloads.config2(B)
ds = solver.DeviceSolver(B, w["N"], w["M"], 6, w["model"])
ds.upload(w)
ds.solve(); torch.cuda.synchronize()
out = {"B": B, "shipped_key_ms": ds.time_solve(20)}
ds.order_by_last_iters = True
ds.solve(); torch.cuda.synchronize()
out["perfect_knowledge_ms"] = ds.time_solve(20)
ds.iters.fill_(7)
# (time_solve rewrites iters every launch: constant hint only for the first; so time single launches)
ts = []
for _ in range(10):
    ds.iters.fill_(7); torch.cuda.synchronize()
    ts.append(ds.time_solve(1))
out["constant_hint_ms"] = sorted(ts)[len(ts) // 2]
print(json.dumps(out))
