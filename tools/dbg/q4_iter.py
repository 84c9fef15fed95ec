#!/usr/bin/env python3
"""GPU box: the HIP path against the oracle after exactly m iterations (maxit = m), m = 1..5 -- where does a variant first leave the oracle?
   python tools/dbg/q4_iter.py [cfg] [B]"""
import sys, numpy as np
sys.path.insert(0, ".")
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = workloads.CONFIGS[cfg](B)
for m in (1, 2, 3, 4, 5, 200):
    og = solver.default_options(maxit=m); oo = OL.default_options(maxit=m)
    z, fl, it, info = solver.solve_batch_host(w, og)
    zo, flo, io = OL.solve_batch(w, oo)
    ito = np.array([i.it for i in io])
    d = np.abs(z - zo).reshape(B, -1).max(1)
    print(f"maxit {m:3d}: max|dz| {d.max():.3e} (median {np.median(d):.1e})  flags equal {(fl == flo).mean():.3f}  its equal {(it == ito).mean():.3f}  mean its {it.mean():.2f} / {ito.mean():.2f}", flush=True)
    if m == 1:
        k = int(np.argmax(d)); dz = np.abs(z[k] - zo[k])
        print("   worst problem", k, "per-stage max:", np.array2string(dz.max(1), precision=1), "\n   per-variable max:", np.array2string(dz.max(0), precision=1))
