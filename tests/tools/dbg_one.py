"""Debug helper: one problem of a config2 batch, GPU vs oracle for increasing maxit (run on the GPU box).
   python tests/tools/dbg_one.py <batch> <index> [maxit_hi]"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
B, b = int(sys.argv[1]), int(sys.argv[2]); hi = int(sys.argv[3]) if len(sys.argv) > 3 else 12
wf = workloads.config2(B, seed=workloads.SEED0 + 3)
w = {k: (v[b:b + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in wf.items()}
for mi in range(0, hi + 1):
    z, fl, it, info = solver.solve_batch_host(w, solver.default_options(maxit=mi))
    zo, flo, io = OL.solve_batch(w, OL.default_options(maxit=mi))
    i = io[0]
    print(f"maxit {mi:2d}: gpu fl {fl[0]:2d} it {it[0]:2d} eq {info[0,0]:.3e} st {info[0,2]:.3e} comp {info[0,3]:.3e} obj {info[0,4]:.5f} mu {info[0,5]:.3e} a {info[0,6]:.3f} fb {info[0,7]:.0f}"
          f" | orc fl {flo[0]:2d} it {i.it:2d} eq {i.res_eq:.3e} st {i.rsnorm:.3e} obj {i.pobj:.5f} mu {i.mu:.3e} a {i.step_cc:.3f} fb {i.nfallback} |dz| {np.max(np.abs(z-zo)):.2e}")
