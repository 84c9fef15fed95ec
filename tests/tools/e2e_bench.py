"""End-to-end rate of frp_nmpc_solve_batch_host (host buffers in and out): the reference's dense 30-row parameter layout and the\ncompact 6-row layout of the same problems.  python tests/tools/e2e_bench.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
w = workloads.config2(4096, seed=workloads.SEED0 + 3)
out = solver.solve_batch_host(w)
ts = []
for _ in range(9):
    t = time.perf_counter(); z, fl, it, info = solver.solve_batch_host(w, out=out); ts.append(time.perf_counter() - t)
print("dense M=30: ms", np.median(ts) * 1e3, "solves/s", 4096 / np.median(ts), "conv", (fl == 1).mean())
zo, flo, io = OL.solve_batch(w, nthreads=16)
print("max |z - oracle|", np.abs(z - zo).max(), (fl == flo).all())
# compact layout (M = 6 rows per stage instead of the reference's 30 padded rows)
p = w["params"]; B, N = p.shape[0], p.shape[1]
pc = np.concatenate([p[:, :, :10], p[:, :, 10:10 + 18], p[:, :, 100:106]], axis=2).copy()
wc = dict(w); wc["params"] = pc; wc["M"] = 6
outc = solver.solve_batch_host(wc)
ts = []
for _ in range(9):
    t = time.perf_counter(); zc, flc, itc, infoc = solver.solve_batch_host(wc, out=outc); ts.append(time.perf_counter() - t)
print("compact M=6: ms", np.median(ts) * 1e3, "solves/s", 4096 / np.median(ts), "same plans", np.abs(zc - z).max())
