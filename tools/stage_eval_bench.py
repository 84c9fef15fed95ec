"""HBM roofline of the batched model callback (frp_nmpc_stage_eval = the reference's extfunc for B*N stage points,
FORCESNLPsolver_normal_casadi2forces.c:42-245): device buffers in, device buffers out, HIP-event timing.
   python tools/stage_eval_bench.py [B] [N]   ->  JSON (profiles/r02_stage_eval_roofline.json)"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forces_resilient_planner_amd import solver, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = 30
w = workloads.config2(min(B, 4096))
rep = (B + w["x0"].shape[0] - 1) // w["x0"].shape[0]
dev = torch.device("cuda:0")
z = torch.from_numpy(np.tile(w["x0"], (rep, 1, 1))[:B]).to(dev).contiguous()
p = torch.from_numpy(np.tile(w["params"], (rep, 1, 1))[:B]).to(dev).contiguous()
f64 = dict(dtype=torch.float64, device=dev)
f = torch.zeros((B, N), **f64); gf = torch.zeros((B, N, 17), **f64); c = torch.zeros((B, N, 13), **f64)
Jc = torch.zeros((B, N, 221), **f64); h = torch.zeros((B, N, M), **f64)
lib = solver.lib()
lib.frp_nmpc_stage_eval.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 8
s = torch.cuda.current_stream(dev)
def run(want="f gf c Jc h"):
    w_ = want.split()
    ptr = lambda name, t: t.data_ptr() if name in w_ else None
    rc = lib.frp_nmpc_stage_eval(B, N, M, 0, z.data_ptr(), p.data_ptr(), ptr("f", f), ptr("gf", gf), ptr("c", c), ptr("Jc", Jc), ptr("h", h), s.cuda_stream)
    assert rc == 0
def timed(want, reps=20):
    run(want); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run(want)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_in = (17 + 10 + 4 * M) * 8.0; bytes_out = (1 + 17 + 13 + 221 + M) * 8.0
tot = B * N * (bytes_in + bytes_out)
out = {"kernel": "stage_eval_kernel", "stage_points": B * N, "ms": ms, "bytes_in_per_point": bytes_in, "bytes_out_per_point": bytes_out,
       "algorithmic_GB": tot / 1e9, "achieved_GBps": tot / (ms * 1e-3) / 1e9, "peak_GBps": 8000.0, "frac": tot / (ms * 1e-3) / 1e9 / 8000.0,
       "stage_points_per_s": B * N / (ms * 1e-3),
       "ms_by_output": {k: timed(k) for k in ("f gf", "c", "Jc", "h")},
       "note": "one thread per (problem, stage) computes; 147 doubles in, 282 doubles out per point (dense 13 x 17 Jacobian, column-major ld 13, as the reference's sparse2fullcopy writes it); the Jacobian and the corridor rows leave / enter through LDS so that the wavefront moves runs of consecutive doubles"}
print(json.dumps(out, indent=1))
