"""Debug helper: compare GPU and oracle iterates for increasing maxit (run on the GPU box)."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = workloads.CONFIGS[cfg]() if cfg == 0 else workloads.CONFIGS[cfg](B)
for mi in list(range(0, 10)):
    z, fl, it, info = solver.solve_batch_host(w, solver.default_options(maxit=mi))
    zo, flo, io = OL.solve_batch(w, OL.default_options(maxit=mi))
    b = 0
    print(f"maxit {mi}: gpu flag {fl[b]} it {it[b]} eq {info[b,0]:.3e} in {info[b,1]:.3e} st {info[b,2]:.3e} comp {info[b,3]:.3e} obj {info[b,4]:.6f} mu {info[b,5]:.3e} a {info[b,6]:.3f} sig {info[b,7]:.3e}")
    i = io[b]
    print(f"          orc flag {flo[b]} it {i.it} eq {i.res_eq:.3e} in {i.res_ineq:.3e} st {i.rsnorm:.3e} comp {i.rcompnorm:.3e} obj {i.pobj:.6f} mu {i.mu:.3e} a {i.step_cc:.3f} sig {i.sigma:.3e}  |dz| {np.max(np.abs(z-zo)):.3e}")
