#!/usr/bin/env python3
"""One timing per kernel variant of the LDS-resident solver (stage stride x corridor rows): serial launches on one stream."""
import json, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from forces_resilient_planner_amd import solver, workloads

def run(name, w, steps=6):
    B = w["B"]
    MF = int(w["nfaces"].max()) if w["nfaces"].size else 0
    ds = solver.DeviceSolver(B, w["N"], w["M"], max(MF, 1), w["model"]); ds.upload(w)
    s = torch.cuda.current_stream()
    ds.solve(s); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): ds.solve(s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s MF %2d  %8.3f ms per batch of %d" % (name, MF, dt / steps * 1e3, B), flush=True)

run("(20, 2) configs[2]", workloads.config2(4096))
run("(20, 5) N=20, <= 15 faces", workloads.config3(4096, N=20, M=15))
run("(20,10) N=20, <= 30 faces", workloads.config3(4096, N=20, M=30))
run("(32, 8) N=30, <= 15 faces", workloads.config3(4096, N=30, M=15))
run("(32,15) N=30, <= 30 faces", workloads.config3(4096, N=30, M=30))
run("(64, 8) N=48, <= 8 faces", workloads.config3(2048, N=48, M=8))
run("(64,30) N=48, <= 30 faces", workloads.config3(2048, N=48, M=30))
