"""frp_nmpc_options.twist: launch time of the solver kernel against the batch size, plain vs twisted, and the single-problem
drop-in call (FORCESNLPsolver_normal_solve, BASELINE configs[0]) with and without FRP_NMPC_TWIST.  GPU box; writes one text table."""
import sys, os, subprocess, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def dropin():
    from forces_resilient_planner_amd import solver, workloads
    w0 = workloads.config0()
    p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
    p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel(); p.num_of_threads = 1
    lat = []
    for i in range(105):
        t1 = time.perf_counter()
        flag = solver.lib().FORCESNLPsolver_normal_solve(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
        lat.append(time.perf_counter() - t1)
    print(f"dropin FRP_NMPC_TWIST={os.environ.get('FRP_NMPC_TWIST', '(unset)')}: median {np.median(lat[5:]) * 1e6:.1f} us, min {np.min(lat[5:]) * 1e6:.1f} us, "
          f"exitflag {flag}, iterations {info.it}, pobj {info.pobj:.9f}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "dropin":
    dropin()
    sys.exit(0)

import torch
from forces_resilient_planner_amd import solver, workloads
TW = (0, -1, 6, 8, 10, 12)
for cfg in (1, 2):
    for B in (1, 16, 64, 256, 512, 768, 1024, 2048, 4096):
        w = workloads.CONFIGS[cfg](B)
        row = []
        for tw in TW:
            ds = solver.DeviceSolver(B, w["N"], w["M"], max(1, int(w["nfaces"].max())), w["model"])
            ds.upload(w); ds.opt.twist = tw
            ds.time_solve(3)
            row.append(min(ds.time_solve(10) for _ in range(3)))
            torch.cuda.synchronize()
        print(f"configs[{cfg}] B {B:5d}  kernel us: " + "  ".join(f"twist {t:2d}: {m * 1e3:7.1f}" for t, m in zip(TW, row)) +
              f"   twist -1 / plain = {row[1] / row[0]:.3f}", flush=True)
for env in ({}, {"FRP_NMPC_TWIST": "-1"}):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "dropin"], env=e)
