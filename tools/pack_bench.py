#!/usr/bin/env python3
"""Roofline of the f-1 kernels (device-side adapter packing / result bookkeeping): HBM-bound, so
achieved = algorithmic bytes / kernel time against 8 TB/s.   python tools/pack_bench.py [B=65536]"""
import json
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from forces_resilient_planner_amd import solver, workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = workloads.config2(min(B, 4096))
rep = (B + w["B"] - 1) // w["B"]
tile = lambda a: np.concatenate([a] * rep, 0)[:B]
N, M, F = w["N"], w["M"], w["poly_A"].shape[2]
fleet = solver.DeviceFleet(B, N, M, F, w["model"], workloads._weights(w["model"]))
fleet.mpc_output.copy_(fleet.to_device(tile(w["mpc_output"])))
fleet.ellipsoid.copy_(fleet.to_device(tile(w["E"])))
fleet.poly_A.copy_(fleet.to_device(tile(w["poly_A"]))); fleet.poly_b.copy_(fleet.to_device(tile(w["poly_b"])))
fleet.poly_nfaces.copy_(fleet.to_device(tile(w["nfaces"]), dtype=torch.int32))
fext, ref, yaw = fleet.to_device(tile(w["f_ext"])), fleet.to_device(tile(w["ref_pos"])), fleet.to_device(tile(w["ref_yaw"]))
fleet.solver.z.copy_(fleet.solver.x0); fleet.solver.exitflag.fill_(1)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t_pack = timeit(lambda: fleet.pack(fext, ref, yaw))
t_upd = timeit(lambda: fleet.update())
np_ = 10 + 4 * M
by_pack = B * (8 * ((N + 1) * 17 + 3 + N * 3 + N + N * 9 + N * F * 4) + 4 * N + 8 * (9 + N * 17 + N * np_) + 4 * N)
by_upd = B * 8 * (N * 17 + (N + 1) * 17) + 4 * B
print(json.dumps({"B": B, "N": N, "M": M, "faces_stored": F,
                  "pack": {"seconds": t_pack, "algorithmic_bytes": by_pack, "GBps": by_pack / t_pack / 1e9, "frac_of_8TBps": by_pack / t_pack / 8e12},
                  "update": {"seconds": t_upd, "algorithmic_bytes": by_upd, "GBps": by_upd / t_upd / 1e9, "frac_of_8TBps": by_upd / t_upd / 8e12}}))
