import sys, numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import solver, workloads
w = workloads.config2(256)
z, fl, it, info = solver.solve_batch_host(w)
import os
if os.environ.get("SEG")=="factor":
    print("last factor sweep segments (cycles/sweep): stage+sync %.0f Ctile %.0f X,G mfma %.0f gather %.0f inverse %.0f Rlds %.0f T,S,P %.0f" % tuple(info[:, :7].mean(0)))
else:
    print("last costate sweep segments (cycles per sweep, mean): stage+sync %.0f  Ctile %.0f  mfma1 %.0f  M+mfma2 %.0f  tail %.0f | wait %.0f" % tuple(info[:, [0,1,2,3,4,6]].mean(0)))
