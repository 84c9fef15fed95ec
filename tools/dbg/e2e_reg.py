"""GPU box: frp_nmpc_solve_batch_host with registered buffers, 10 calls (for rocprofv3 --kernel-trace --stats)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from forces_resilient_planner_amd import solver, workloads
w = workloads.config2(4096)
w = {k: (np.ascontiguousarray(v, dtype=(np.int32 if k == "nfaces" else np.float64)) if isinstance(v, np.ndarray) else v) for k, v in w.items()}
ref = solver.solve_batch_host(w)
out = tuple(np.zeros_like(a) for a in ref)
reg = [w["xinit"], w["x0"], w["params"], w["nfaces"]] + list(out)
solver.host_register(*reg)
ts = []
for _ in range(12):
    t = time.perf_counter(); solver.solve_batch_host(w, out=out); ts.append(time.perf_counter() - t)
print("registered ms", np.median(ts) * 1e3, "min", min(ts) * 1e3)
solver.host_unregister(*reg)
