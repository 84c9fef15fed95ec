"""Launch time of the solver kernel against the batch size, plain vs twisted (the latency regime: fewer problems than resident slots)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from forces_resilient_planner_amd import solver, workloads
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for B in (1, 16, 64, 256, 512, 768, 1024, 1536, 2048, 4096):
    w = workloads.CONFIGS[cfg](B)
    row = []
    for tw in (0, 4, 6, 8, 10, 12):
        ds = solver.DeviceSolver(B, w["N"], w["M"], max(1, int(w["nfaces"].max())), w["model"])
        ds.upload(w); ds.opt.twist = tw
        ds.time_solve(3)
        ms = min(ds.time_solve(10) for _ in range(3))
        torch.cuda.synchronize()
        row.append(ms)
    print(f"cfg {cfg} B {B:5d}  " + "  ".join(f"tw{t}: {m * 1e3:7.1f} us" for t, m in zip((0, 4, 6, 8, 10, 12), row)), flush=True)
