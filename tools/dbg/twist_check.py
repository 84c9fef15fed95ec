"""GPU twisted solve vs the oracle's twisted solve (and the plain GPU solve), iterate by iterate (maxit = 1, 2, ...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from forces_resilient_planner_amd import solver, workloads
from tests import oracle_lib as OL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tw = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for cfg in (1, 2):
    w = workloads.CONFIGS[cfg](B)
    for maxit in (1, 2, 3, 200):
        og = solver.default_options(); og.maxit = maxit
        ot = solver.default_options(); ot.maxit = maxit; ot.twist = tw
        z0, f0, i0, _ = solver.solve_batch_host(w, og)
        z1, f1, i1, _ = solver.solve_batch_host(w, ot)
        zo, fo, io = OL.solve_batch(w, OL.default_options(maxit=maxit))
        d = np.abs(z1 - z0); dd = np.abs(z0 - zo)
        print(f"{cfg} tw={tw} maxit={maxit:3d}  |tw-plain|={d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)} |plain-orc|={dd.max():.2e} flags eq {np.mean(f1 == f0):.3f}"
              f"  it eq {np.mean(i1 == i0):.3f}  nan={np.isnan(z1).sum()}", flush=True)
        if maxit == 1:
            print("   per-stage max:", " ".join(f"{x:.1e}" for x in d.max(axis=(0, 2))))
            print("   per-comp  max:", " ".join(f"{x:.1e}" for x in d.max(axis=(0, 1))))
