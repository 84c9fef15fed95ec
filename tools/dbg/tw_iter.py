#!/usr/bin/env python3
"""GPU box: twisted solve (hybrid end game) against the oracle's on hard instances at 1e-8 tolerances: flags, and for the first instance that
differs the iterate after m iterations.   python tools/dbg/tw_iter.py"""
import sys, numpy as np
sys.path.insert(0, ".")
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
w = workloads.config_hard(512, seed=303, model=0)
kw = dict(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8, twist=-1)
z, fl, it, info = solver.solve_batch_host(w, solver.default_options(**kw))
zo, flo, io = OL.solve_batch(w, OL.default_options(**kw))
ito = np.array([i.it for i in io])
print("flags equal", (fl == flo).mean(), "gpu converged", (fl == 1).sum(), "oracle", (flo == 1).sum(), "its equal", (it == ito).mean())
bad = np.nonzero(fl != flo)[0]
print("differ:", bad[:10], fl[bad[:10]], flo[bad[:10]], it[bad[:10]], ito[bad[:10]])
if len(bad):
    b = int(bad[0])
    from tests.tools.twist_certify import sub_batch
    w1 = sub_batch(w, np.array([b]))
    for m in range(1, 14):
        k2 = dict(kw, maxit=m)
        z1, f1, i1, inf1 = solver.solve_batch_host(w1, solver.default_options(**k2))
        zo1, fo1, io1 = OL.solve_batch(w1, OL.default_options(**k2))
        print(f"maxit {m:2d}: |dz| {np.abs(z1 - zo1).max():.2e}  gpu res eq {inf1[0,0]:.1e} in {inf1[0,1]:.1e} rs {inf1[0,2]:.1e} rc {inf1[0,3]:.1e} | oracle eq {io1[0].res_eq:.1e} rs {io1[0].rsnorm:.1e} rc {io1[0].rcompnorm:.1e}  flags {f1[0]} {fo1[0]}")
