"""First contact of a solver kernel with the hardware: per-maxit agreement with the oracle on a few problems, then
batch agreement on configs[1..3].  python tests/tools/lds_first_contact.py [B]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
import tests.oracle_lib as OL

def sub(wf, B, b0, n):
    return {k: (v[b0:b0 + n] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in wf.items()}

def per_maxit(name, wf, B, b, hi):
    w = sub(wf, B, b, 1)
    for mi in range(0, hi + 1):
        z, fl, it, info = solver.solve_batch_host(w, solver.default_options(maxit=mi))
        zo, flo, io = OL.solve_batch(w, OL.default_options(maxit=mi))
        i = io[0]
        print(f"{name} #{b} maxit {mi:2d}: gpu fl {fl[0]:2d} it {it[0]:2d} eq {info[0,0]:.3e} in {info[0,1]:.3e} st {info[0,2]:.3e} comp {info[0,3]:.3e} obj {info[0,4]:.5f} mu {info[0,5]:.3e} a {info[0,6]:.3f}"
              f" | orc fl {flo[0]:2d} it {i.it:2d} eq {i.res_eq:.3e} in {i.res_ineq:.3e} st {i.rsnorm:.3e} comp {i.rcompnorm:.3e} obj {i.pobj:.5f} mu {i.mu:.3e} a {i.step_cc:.3f} |dz| {np.max(np.abs(z-zo)):.2e}", flush=True)

def batch(name, w):
    t0 = time.time()
    z, fl, it, info = solver.solve_batch_host(w)
    t1 = time.time()
    zo, flo, io = OL.solve_batch(w, nthreads=16)
    ito = np.array([i.it for i in io])
    ok = (fl == 1) & (flo == 1)
    same = ok & (it == ito)
    print(f"{name}: B {len(fl)} gpu conv {np.mean(fl == 1):.4f} mean it {it.mean():.2f} max {it.max()} | orc conv {np.mean(flo == 1):.4f} mean it {ito.mean():.2f} | flags differ {np.sum(fl != flo)}"
          f" | it differ {np.sum(ok & (it != ito))} | max|dz| same-it {np.max(np.abs(z[same] - zo[same])) if same.any() else -1:.2e} all-conv {np.max(np.abs(z[ok] - zo[ok])) if ok.any() else -1:.2e} | host call {t1 - t0:.2f}s", flush=True)
    bad = np.where(fl != flo)[0]
    for b in bad[:5]:
        print("   ", b, "gpu", fl[b], it[b], info[b, :4], "orc", flo[b], ito[b], io[b].res_eq, io[b].rsnorm)

print(solver.lib().frp_nmpc_version())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w2 = workloads.config2(B)
per_maxit("cfg2", w2, B, 0, 7)
per_maxit("cfg2", w2, B, 1, 3)
batch("cfg0", workloads.config0())
batch("cfg1", workloads.config1(B))
batch("cfg2", w2)
batch("cfg3", workloads.config3(B))
batch("cfg2-final", workloads.config2(B, model=1))
batch("cfg2-B4096", workloads.config2(4096))
