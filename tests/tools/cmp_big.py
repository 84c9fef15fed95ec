"""GPU vs oracle on a big batch: report problems whose exit flag / iteration count differ."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
w = workloads.config2(B, seed=workloads.SEED0 + 3)
z, fl, it, info = solver.solve_batch_host(w)
zo, flo, io = OL.solve_batch(w, nthreads=16)
ito = np.array([i.it for i in io])
print("gpu: conv", (fl == 1).mean(), "max it", it.max(), "flags", np.unique(fl, return_counts=True))
print("orc: conv", (flo == 1).mean(), "max it", ito.max())
bad = np.where((fl != flo) | (np.abs(it - ito) > 1))[0]
print("differing problems:", len(bad))
for b in bad[:10]:
    print(b, "gpu", fl[b], it[b], "orc", flo[b], ito[b], "gpu info", info[b, :4], "orc", io[b].res_eq, io[b].rsnorm)
ok = (fl == 1) & (flo == 1)
print("max |dz| on converged:", np.max(np.abs(z[ok] - zo[ok])))
