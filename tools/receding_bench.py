#!/usr/bin/env python3
"""BASELINE.json configs[4] at full scale on one GPU: B Monte-Carlo planners (f_ext ~ N(fbar, 0.5^2 I) around one
nominal problem, N = 20), T warm-started receding-horizon ticks, every tick on the device (pack -> solve -> update,
SURVEY 8f row f-1).  Prints solves/s over all B*T solves and the iteration-count drop against the cold first tick.
With a third argument `tube` the tube matrices are also recomputed from the plans every tick on the device (row f-2),
which is the reference's complete per-tick computation short of corridor generation.
   python tools/receding_bench.py [B=65536] [ticks=20] [tube]"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import receding, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
TUBE = len(sys.argv) > 3 and sys.argv[3] == "tube"
t0 = time.time()
w0 = workloads.config4_nominal(B=B, ticks=T)
t_gen = time.time() - t0
receding.run_device(dict(w0, B=min(B, 256), **{k: w0[k][:min(B, 256)] for k in ("mpc_output", "E", "f_ext")}), 2)  # warm-up
flags, iters, plan, secs = receding.run_device(w0, T, propagate_tube=TUBE)
ok = flags == 1
print(json.dumps({
    "workload": f"configs[4]: B={B} planners x {T} receding-horizon ticks, N=20, device-side " + ("tube/" if TUBE else "") + "pack/solve/update",
    "solves": int(B * T), "gpu_seconds": secs, "solves_per_s": B * T / secs, "ms_per_tick": secs / T * 1e3,
    "converged_frac": float(ok.mean()), "mean_iters_tick0_cold": float(iters[0].mean()),
    "mean_iters_warm_ticks": float(iters[1:].mean()), "max_iters": int(iters.max()),
    "host_generation_seconds": t_gen}))
