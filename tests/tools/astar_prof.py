import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from forces_resilient_planner_amd import solver, workloads
w = workloads.astar_world(1, "pillars", allocate_num=40000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
q = workloads.astar_queries(B, 1)
pl = solver.AstarPlanner(w, B, K=2048, want_path_nodes=True)
pl.upload(q["start_pt"], q["start_v"], q["start_a"], q["end_pt"], q["end_v"], q["f_ext"])
pl.plan(); torch.cuda.synchronize()
t0 = time.time(); pl.plan(); torch.cuda.synchronize(); print("gpu ms", (time.time() - t0) * 1e3)
stats = pl.stats.cpu().numpy(); pn = pl.path_nodes.cpu().numpy()
it = stats[:, 1].astype(float)
prof = pn[:, -1, :11]
names = ["top", "window", "pop", "transit+hash", "collision", "heuristic", "leader", "commit: node writes", "commit: set-up loads", "commit: walk", "commit: wait for the hash lane"]
tot = prof[:, :11].sum(0); its = it.sum()
print("expansions", its, "cycles per expansion:", {n: round(tot[i] / its) for i, n in enumerate(names)}, "total/exp", round(tot.sum() / its))
b = int(np.argmax(it)); print("longest: it", it[b], {n: round(prof[b, i] / it[b]) for i, n in enumerate(names)})
