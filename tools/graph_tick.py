#!/usr/bin/env python3
"""Small fleets are launch-bound: one tick = 9 kernel launches.  Every entry point of the C-ABI is asynchronous on the
caller's stream and allocates nothing, so the whole tick can be captured once into a hipGraph and replayed.
The FIRST replay after capture is the measured one's predecessor on purpose: it is compared too (a 4-byte
hipMemsetAsync captured as a memset node did not reset the solver's queue counter on ROCm 7.2 -- the solve kernel then
exits at once and leaves the previous outputs in place, which looks converged; launch_ipm resets the counter with a
one-thread kernel instead, and this tool poisons z / exitflag before every run so that a dead solve cannot hide).
   python tools/graph_tick.py [B=64] [ticks=200] [P=5000]"""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
P = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
N, M, F, K = 20, 30, 64, 400
rng = np.random.default_rng(0)
s = np.arange(K) * 0.05 * 0.4
path = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
cloud = np.c_[rng.uniform(-3, 12, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.9]
plan = np.zeros((B, N + 1, 17)); plan[..., 3] = 7.3; plan[..., 7] = 7.3
plan[..., 8:11] = path[0] + rng.normal(0, 0.02, (B, 1, 3)); plan[..., 16] = 0.2
fleet = solver.DeviceFleet(B, N, M, F, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0))
d_path, d_cloud = fleet.to_device(path), fleet.to_device(cloud)
d_f = fleet.to_device(rng.normal(0, 0.5, (B, 3)))
rp = torch.zeros((B, N, 3), dtype=torch.float64, device="cuda:0"); ry = torch.zeros((B, N), dtype=torch.float64, device="cuda:0")
toff = torch.zeros((B,), dtype=torch.float64, device="cuda:0")
fleet.poly_index = torch.zeros((B, N), dtype=torch.int32, device="cuda:0")

def run(ticks, tick_fn):
    fleet.mpc_output.copy_(fleet.to_device(plan)); toff.zero_()
    fleet.solver.z.fill_(float('nan')); fleet.solver.exitflag.fill_(-99)   # a solve that silently does nothing must show
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ticks):
        tick_fn()
        toff.add_(0.05)          # the next tick starts one sample later (device-side, so the graph sees it)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ticks, fleet.mpc_output.clone(), int((fleet.solver.exitflag == 1).sum())

side = torch.cuda.Stream()
def eager():
    fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry)
eager(); torch.cuda.synchronize()
ms_eager, plan_eager, ok_eager = run(TICKS, eager)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry, stream=side)   # warm-up on the capture stream
side.synchronize()
with torch.cuda.graph(g, stream=side):
    fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry, stream=torch.cuda.current_stream())
ms_graph, plan_graph, ok_graph = run(TICKS, g.replay)
print(json.dumps({"B": B, "ticks": TICKS, "cloud_points": len(cloud), "ms_per_tick_eager": ms_eager, "ms_per_tick_graph": ms_graph,
                  "speedup": ms_eager / ms_graph, "converged_last_tick": [ok_eager, ok_graph],
                  "max_plan_difference_graph_vs_eager": float((plan_eager - plan_graph).abs().max())}))
