#!/usr/bin/env python3
"""frp_nmpc_tube_batch against oracle/tube_oracle.py on the random plans of tests/tools/soak_corridor.py (states drawn over the whole state
box): the worst relative difference of an E entry, where it is, and the distribution.   python tests/tools/tube_accuracy.py [worlds=4000]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver
from oracle import tube_oracle as T
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lb, ub = L.bounds()
B, N = 8, 20
zs = []
for seed in range(1, W + 1):
    rng = np.random.default_rng(seed)
    # (the draws of soak_corridor.py up to z: cloud size, grid flag, speed, path, tunnel -- consumed so that z is the same array)
    P = int(rng.choice([300, 2000, 6000, 20000])); rng.uniform(-3, 9, P); rng.uniform(-4, 4, P); rng.uniform(-0.5, 3, P)
    rng.random(); rng.uniform(0.5, 2.5); rng.uniform(0.1, 0.6); rng.uniform(0.5, 1.5); rng.uniform(0.2, 0.9)
    z = lb + (ub - lb) * rng.random((B, N + 1, 17))
    z[..., 11:14] = rng.uniform(-5, 5, (B, N + 1, 3)); z[..., 16] = rng.uniform(-3.1, 3.1, (B, N + 1))
    zs.append(z[:, :N])
z = np.concatenate(zs)
E = solver.tube_batch_host(z)
rel = np.empty(len(z))
for p in range(len(z)):
    Eo = T.tube_one(z[p])
    rel[p] = float(np.max(np.abs(E[p] - Eo) / (1e-3 + np.abs(Eo))))
w = int(np.argmax(rel))
print(json.dumps({"plans": len(z), "max_rel": float(rel.max()), "at_world_planner": [w // B + 1, w % B], "p50": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)),
                  "p999": float(np.quantile(rel, 0.999)), "above_1e-9": int((rel > 1e-9).sum())}))
