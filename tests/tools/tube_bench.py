#!/usr/bin/env python3
"""Measurement of the f-2 kernel (tube propagation): FP64-VALU-bound, so achieved = algorithmic flops / kernel time
against the 78.6 TFLOP/s FP64 vector peak; the numpy/scipy oracle is timed beside it on a bounded sample.
    python tests/tools/tube_bench.py [B=4096] [N=20]"""
import json
import sys
import time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver
from oracle import tube_oracle as T   # CPU-baseline leg only

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(0)
lb, ub = L.bounds()
z = lb + (ub - lb) * rng.random((B, N + 1, 17))
z[..., 11:14] = rng.uniform(-6, 6, (B, N + 1, 3))
solver.lib()
mo = torch.from_numpy(z).to("cuda:0")
E = torch.empty((B, N, 3, 3), dtype=torch.float64, device="cuda:0")
fn = lambda: solver.tube_batch_device(mo, E)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
# algorithmic flops per (stage, channel) thread: a product with Phi or Phi' is 36 multiply-adds, a Taylor term adds
# 9 scalings + 9 accumulations; 8 quadrature nodes x (14 terms + 9 + 45 outer-product multiply-adds) and
# 8 x 14 terms for the row of exp(Phi t); the 9x9 recursion and the 3x3 root are < 2 % and not counted
term = 2 * 36 + 18
per_thread = 8 * (14 * term + 9 + 2 * 45) + 8 * 14 * term
flops = B * N * 3 * per_thread
ns = min(B, 16)
t0 = time.time(); Eo = T.tube_batch(z[:ns, :N]); tc = time.time() - t0
err = float(np.max(np.abs(E[:ns].cpu().numpy() - Eo) / (1e-3 + np.abs(Eo))))
print(json.dumps({"B": B, "N": N, "seconds": t, "planners_per_s": B / t, "algorithmic_flops": flops,
                  "TFLOPs": flops / t / 1e12, "frac_of_78.6_TFLOPs_fp64_vector": flops / t / 78.6e12,
                  "cpu_oracle": {"planners_per_s": ns / tc, "sample": f"{ns} planners x {N} stages, numpy/scipy, 1 core"},
                  "max_rel_err_vs_oracle": err}))
