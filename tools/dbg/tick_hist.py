import sys, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import full_tick_bench as F
from forces_resilient_planner_amd import solver
# monkeypatch: capture iteration counts of the last tick
orig = solver.DeviceSolver.solve
its = []; rows = []
def solve(self, *a, **k):
    rows.append(self.nfaces.clone())  # live corridor rows per (planner, stage) of THIS solve, as packed
    r = orig(self, *a, **k); its.append(self.iters.clone()); return r
solver.DeviceSolver.solve = solve
F.run(B=4096, TICKS=6, P=20000, GRID=0.5, SPLIT=0)
torch.cuda.synchronize()
for t in its[-3:]:
    v = t.cpu().numpy(); print("mean %.2f  p50 %d  p90 %d  p99 %d  max %d  hist %s" % (v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), np.bincount(v)[:16].tolist()))
a, b = its[-2].cpu().numpy(), its[-1].cpu().numpy()
print("correlation of consecutive ticks' iteration counts", np.corrcoef(a, b)[0, 1])
# how many corridor rows does a stage of the tick's problems have live?  (the (20, 10) solver variant holds 30)
nf = rows[-1].cpu().numpy()
print("live corridor rows per stage, last tick: mean %.1f  p10 %d  p50 %d  p90 %d  max %d" % (nf.mean(), np.percentile(nf, 10), np.percentile(nf, 50), np.percentile(nf, 90), nf.max()))
print("  histogram over (planner, stage) by rows 0..30:", np.bincount(nf.ravel(), minlength=31).tolist())
pm = nf.max(axis=1)
print("  per planner, max over its stages: share with <= 6 rows %.3f, <= 12 %.3f, <= 15 %.3f, <= 20 %.3f, <= 24 %.3f; histogram %s" % (
      (pm <= 6).mean(), (pm <= 12).mean(), (pm <= 15).mean(), (pm <= 20).mean(), (pm <= 24).mean(), np.bincount(pm, minlength=31).tolist()))
