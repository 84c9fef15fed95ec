#!/usr/bin/env python3
"""GPU box: latency of the drop-in call (BASELINE configs[0] through FORCESNLPsolver_normal_solve), median of 200 warm calls, for the library
FRP_LIB selects; with FRP_NMPC_TWIST in the environment the latency option.   python tools/r06/dropin_lat.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
w0 = workloads.config0()
p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel(); p.num_of_threads = 1
rows = int(os.environ.get("DROPIN_ROWS", "0"))  # > 6: the stage's block padded with far-away (inactive, but live) corridor rows up to this count -- a planner whose polyhedra have ~30 faces
if rows > 6:
    M = (w0["params"].shape[2] - 10) // 4
    pr = w0["params"][0].copy()
    for j in range(6, min(rows, M)):
        ang = 0.7 * j
        pr[:, 10 + 3 * j:10 + 3 * j + 3] = [np.cos(ang), np.sin(ang), 0.3 * np.cos(2.1 * j)]
        pr[:, 10 + 3 * M + j] = 60.0 + j
    p.all_parameters[:] = pr.ravel()
lat = []
for i in range(230):
    t1 = time.perf_counter()
    flag = solver.lib().FORCESNLPsolver_normal_solve(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
    lat.append(time.perf_counter() - t1)
print("drop-in call: median %.4f ms  p10 %.4f  p90 %.4f  flag %d its %d pobj %.10f" % (np.median(lat[30:]) * 1e3, np.percentile(lat[30:], 10) * 1e3, np.percentile(lat[30:], 90) * 1e3, flag, info.it, info.pobj))
