import sys, os, numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import solver
g = np.load("tests/golden/solutions_config3.npz")
N, M = int(g["N"]), int(g["M"])
good = g["status"] == 0
for model in np.unique(g["model"]):
    sel = np.where((g["model"] == model) & good)[0]
    print("model", model, "B", len(sel), "N", N, "M", M, "nfaces max", g["nfaces"][sel].max(), flush=True)
    w = dict(xinit=g["xinit"][sel], x0=g["x0"][sel], params=g["params"][sel], nfaces=g["nfaces"][sel], N=N, M=M, model=int(model))
    z, fl, it, info = solver.solve_batch_host(w)
    print("  default tol: flags", fl, "its", it, flush=True)
    opt = solver.default_options(); opt.tol_stat = opt.tol_eq = opt.tol_ineq = opt.tol_comp = 1e-8
    zt, flt, itt, _ = solver.solve_batch_host(w, opt)
    print("  tight tol: flags", flt, "its", itt, flush=True)
