"""Replay tests/tools/soak.py's random launch sequence up to a given (kind, B, seed) and report the mismatching problems."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
target = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
rng = np.random.default_rng(2026)
while True:
    kind = int(rng.integers(0, 4)); B = int(rng.integers(1, 5000)); seed = int(rng.integers(0, 1 << 30))
    N = M = model = None
    if kind == 1: model = int(rng.integers(0, 2))
    elif kind == 2: N = int(rng.integers(2, 41)); M = 15
    elif kind == 3: N = int(rng.integers(41, 65)); M = int(rng.integers(15, 31))
    if (kind, B, seed) == target: break
print("kind", kind, "B", B, "seed", seed, "N", N, "M", M)
w = workloads.config3(min(B, 800 if kind == 3 else 1500), seed=seed, N=N, M=M)
z, fl, it, info = solver.solve_batch_host(w)
zo, flo, io = OL.solve_batch(w, nthreads=16)
ito = np.array([i.it for i in io])
for b in np.where(fl != flo)[0]:
    print("problem", b, "gpu", fl[b], it[b], info[b, :4], "orc", flo[b], ito[b], io[b].res_eq, io[b].res_ineq, io[b].rsnorm, io[b].rcompnorm, "fb gpu", info[b, 7], "orc", io[b].nfallback)
