"""Per-phase cycle counts (needs a library built with -DFRP_PROFILE; run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import solver, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workloads.config2(B)
import time
for rep in range(2):
    t = time.time(); z, fl, it, info = solver.solve_batch_host(w); dt = time.time() - t
names = ["eval", "factor", "fwd affine", "affine+step", "backvec", "fwd corr+y"]
per_it = info[:, :6] / np.maximum(it[:, None], 1)
print("B", B, "mean it", it.mean(), "host wall", dt)
for i, n in enumerate(names):
    print(f"{n:12s} mean cycles/iter {per_it[:, i].mean():10.0f}   (problems with it>=20: {per_it[it >= 20, i].mean() if (it>=20).any() else 0:10.0f})")
print("total cycles/iter", per_it.sum(1).mean(), " total cycles/solve", info[:, :6].sum(1).mean())
