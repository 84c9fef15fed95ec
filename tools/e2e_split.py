"""End-to-end time of frp_nmpc_solve_batch_host for explicit chunk splits (FRP_HOST_SPLIT) next to the default policy.
   python tools/e2e_split.py"""
import os, sys, time, subprocess, numpy as np
code = '''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from forces_resilient_planner_amd import solver, workloads
w = workloads.config2(4096)
out = solver.solve_batch_host(w)
ts = []
for _ in range(11):
    t = time.perf_counter(); solver.solve_batch_host(w, out=out); ts.append(time.perf_counter() - t)
print(os.environ.get("FRP_HOST_SPLIT", "default"), "ms", round(np.median(ts) * 1e3, 3), flush=True)
'''
for sp in ("", "4096", "1366,1365,1365", "256,1024,2816", "256,1280,2560", "512,1024,2560"):
    env = dict(os.environ)
    if sp: env["FRP_HOST_SPLIT"] = sp
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:])
