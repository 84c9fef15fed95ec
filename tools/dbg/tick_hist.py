import sys, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import full_tick_bench as F
from forces_resilient_planner_amd import solver
# monkeypatch: capture iteration counts of the last tick
orig = solver.DeviceSolver.solve
its = []
def solve(self, *a, **k):
    r = orig(self, *a, **k); its.append(self.iters.clone()); return r
solver.DeviceSolver.solve = solve
F.run(B=4096, TICKS=6, P=20000, GRID=0.5, SPLIT=0)
torch.cuda.synchronize()
for t in its[-3:]:
    v = t.cpu().numpy(); print("mean %.2f  p50 %d  p90 %d  p99 %d  max %d  hist %s" % (v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), np.bincount(v)[:16].tolist()))
a, b = its[-2].cpu().numpy(), its[-1].cpu().numpy()
print("correlation of consecutive ticks' iteration counts", np.corrcoef(a, b)[0, 1])
