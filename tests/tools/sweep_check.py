"""Large GPU-vs-oracle sweep over the workload families (run on the GPU box): exit flags, iteration counts, solutions."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL

def check(name, w):
    t = time.time(); z, fl, it, info = solver.solve_batch_host(w); tg = time.time() - t
    t = time.time(); zo, flo, io = OL.solve_batch(w, nthreads=16); to = time.time() - t
    ito = np.array([i.it for i in io])
    ok = (fl == 1) & (flo == 1)
    same = ok & (it == ito)
    print(f"{name}: B {len(fl)} gpu flags {dict(zip(*np.unique(fl, return_counts=True)))} orc flags {dict(zip(*np.unique(flo, return_counts=True)))} "
          f"flag mismatches {(fl != flo).sum()} it equal {100 * (it[ok] == ito[ok]).mean():.2f}% max it gpu {it.max()} orc {ito.max()} "
          f"max|dz| same-it {np.max(np.abs(z[same] - zo[same])):.2e} all {np.max(np.abs(z[ok] - zo[ok])):.2e}  (gpu {tg:.2f}s orc {to:.2f}s)", flush=True)
    bad = np.where(fl != flo)[0]
    for b in bad[:5]:
        print("   mismatch", b, "gpu", fl[b], it[b], "orc", flo[b], ito[b])

for seed in (1, 2, 3):
    check(f"config2 seed {seed}", workloads.config2(32768, seed=seed))
check("config1", workloads.config1(32768, seed=11))
check("config3", workloads.config3(8192, seed=12))
check("config3 N=20 M=30", workloads.config3(8192, seed=13, N=20, M=30))
check("config2 final model", workloads.config2(8192, seed=14, model=1))
