#!/usr/bin/env python3
"""The reference's whole per-tick computation downstream of the A* for a fleet, on one GPU, nothing on the host:
stage references (f-4) -> tube (f-2) -> corridor (f-3) -> packing (f-1) -> NLP solve -> bookkeeping
(DeviceFleet.full_tick).  Prints ms per step of the chain (HIP events on the launch stream) and planner-ticks/s.
   python tools/full_tick_bench.py [B=4096] [ticks=10] [P=20000] [grid_cell=0.5] [sub_fleets=2] [mu0]
bench.py's default run calls run() for its `full_tick` block."""
import json
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver

def run(B=4096, TICKS=10, P=20000, GRID=0.5, SPLIT=2, mu0=None):
    """One fleet of B planners sharing a P-point cloud and a kinodynamic path: ms per step of the tick (HIP events on the launch
    stream, mean over TICKS ticks), the whole tick, and the same fleet as SPLIT sub-fleets on their own streams (0 = skip).
    mu0: frp_nmpc_options.mu0 of the fleet's solves (None: the default, 1).  A tick's problem is the previous one shifted by a stage and
    starts from its solution; the interior-point iteration then spends its first iterations bringing the barrier parameter down from
    mu0 -- a smaller one saves them (4.14 -> 3.70 iterations per solve at 0.2 on this workload, the optimum of the sweep in
    profiles/r05_tick_mu0.txt; below 0.1 the count rises again).  The last tick is solved again with the default and both plans are
    compared (`plans_vs_default_mu0`): same converged points to the solver's tolerances."""
    N, M, F, K = 20, 30, 64, 120
    rng = np.random.default_rng(0)
    s = np.arange(K) * 0.05 * 1.6
    path = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
    cloud = np.c_[rng.uniform(-3, 12, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
    cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.9]
    plan = np.zeros((B, N + 1, 17)); plan[..., 3] = 7.3; plan[..., 7] = 7.3
    plan[..., 8:11] = path[0] + rng.normal(0, 0.02, (B, 1, 3)); plan[..., 16] = 0.2
    fleet = solver.DeviceFleet(B, N, M, F, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0))
    if mu0 is None and os.environ.get("FRP_TICK_MU0"):
        mu0 = float(os.environ["FRP_TICK_MU0"])
    if mu0 is not None:
        fleet.solver.opt.mu0 = float(mu0)
    for k in ("hessian", "ftb"):  # (experiments: other frp_nmpc_options fields from the environment, FRP_TICK_HESSIAN / FRP_TICK_FTB)
        if os.environ.get("FRP_TICK_" + k.upper()):
            setattr(fleet.solver.opt, k, type(getattr(fleet.solver.opt, k))(float(os.environ["FRP_TICK_" + k.upper()])))
    fleet.mpc_output.copy_(fleet.to_device(plan))
    d_path, d_cloud = fleet.to_device(path), fleet.to_device(cloud)
    d_f = fleet.to_device(rng.normal(0, 0.5, (B, 3)))
    grid = solver.CloudGrid(d_cloud, GRID) if GRID > 0 else None   # built once per cloud, not per tick
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device="cuda:0"); ry = torch.zeros((B, N), dtype=torch.float64, device="cuda:0")
    offs = [fleet.to_device(np.full(B, 0.05 * t) + rng.uniform(0, 0.01, B)) for t in range(TICKS + 1)]
    steps = [("reference", lambda t: fleet.references(d_path, offs[t], rp, ry)), ("tube", lambda t: fleet.tube()),
             ("corridor", lambda t: fleet.corridor(d_cloud, rp, ry, grid=grid)), ("pack", lambda t: fleet.pack(d_f, rp, ry)),
             ("solve", lambda t: fleet.solver.solve()), ("update", lambda t: fleet.update())]
    for _, fn in steps:  # warm-up tick
        fn(0)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(steps) + 1)] for _ in range(TICKS)]
    flags, iters = [], []
    for t in range(TICKS):
        ev[t][0].record()
        for k, (_, fn) in enumerate(steps):
            fn(t + 1)
            ev[t][k + 1].record()
        flags.append(fleet.solver.exitflag.clone()); iters.append(fleet.solver.iters.clone())
    torch.cuda.synchronize()
    ms = {name: float(np.mean([ev[t][k].elapsed_time(ev[t][k + 1]) for t in range(TICKS)])) for k, (name, _) in enumerate(steps)}
    total = float(np.mean([ev[t][0].elapsed_time(ev[t][-1]) for t in range(TICKS)]))
    fl = torch.stack(flags).cpu().numpy(); it = torch.stack(iters).cpu().numpy()
    versus_default = None
    if mu0 is not None:  # the last tick's problems once more with the default barrier start: the same points?
        z_opt, f_opt = fleet.solver.z.clone(), fleet.solver.exitflag.clone()
        fleet.solver.opt.mu0 = 1.0
        fleet.solver.solve(); torch.cuda.synchronize()
        both = (f_opt == 1) & (fleet.solver.exitflag == 1)
        dz = (fleet.solver.z - z_opt).abs().amax(dim=(1, 2))
        versus_default = {"converged_with_option": int((f_opt == 1).sum().item()), "converged_with_default": int((fleet.solver.exitflag == 1).sum().item()),
                          "max_abs_dz_where_both_converged": float(dz[both].max().item()) if bool(both.any()) else None,
                          "mean_iters_default_on_the_same_problems": float(fleet.solver.iters.double().mean().item())}
        fleet.solver.z.copy_(z_opt); fleet.solver.exitflag.copy_(f_opt); fleet.solver.opt.mu0 = float(mu0)
    split_ms = None
    if SPLIT > 0:
        # the same fleet as two half-fleets on two streams: the solver's few long problems at the end of one half overlap the
        # corridor / tube work of the other
        half = B // SPLIT
        fl2 = [solver.DeviceFleet(half, N, M, F, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0)) for _ in range(SPLIT)]
        st = [torch.cuda.Stream(device="cuda:0") for _ in range(SPLIT)]
        bufs = []
        for k, f2 in enumerate(fl2):
            f2.mpc_output.copy_(fleet.to_device(plan[k * half:(k + 1) * half]))
            bufs.append((torch.zeros((half, N, 3), dtype=torch.float64, device="cuda:0"), torch.zeros((half, N), dtype=torch.float64, device="cuda:0"),
                         d_f[k * half:(k + 1) * half].contiguous(), [o[k * half:(k + 1) * half].contiguous() for o in offs]))
        torch.cuda.synchronize()
        def split_tick(t):
            for k, f2 in enumerate(fl2):
                r2, y2, fe, of = bufs[k]
                with torch.cuda.stream(st[k]):
                    f2.full_tick(fe, d_path, of[t], d_cloud, r2, y2, stream=st[k], grid=grid)
        split_tick(0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(TICKS):
            split_tick(t + 1)
        for k in range(SPLIT):
            torch.cuda.current_stream().wait_stream(st[k])
        e1.record(); torch.cuda.synchronize()
        split_ms = e0.elapsed_time(e1) / TICKS
    return ({"workload": f"{B} planners x {TICKS} ticks, N=20, shared cloud of {len(cloud)} points, shared kinodynamic path",
                      "grid_cell": GRID, "ms_per_tick": total, "planner_ticks_per_s": B / total * 1e3, "ms_per_step": ms,
                      "sub_fleets_on_own_streams": ({"parts": SPLIT, "ms_per_tick": split_ms, "planner_ticks_per_s": B / split_ms * 1e3} if split_ms else None),
                      "converged_frac": float((fl == 1).mean()), "mean_iters": float(it.mean()),
                      "options": {"mu0": float(fleet.solver.opt.mu0)}, "plans_vs_default_mu0": versus_default,
                      "polytopes_per_planner": float((fleet.poly_nfaces > 0).sum().item() / B),
                      "mean_rows": float(fleet.poly_nfaces[fleet.poly_nfaces > 0].double().mean().item()),
                      # work terms of tools/tick_rooflines.py: points in a local box (4.1 x 4 x 2 m) and in the grid cells under its hull, from the cloud's density
                      "in_box_points_per_decomposition": float(len(cloud)) / (15.0 * 8.0 * 3.5) * 32.8 * 0.7,
                      "grid_candidates_per_decomposition": float(len(cloud)) / (15.0 * 8.0 * 3.5) * (4.6 * 4.5 * 2.5)})


if __name__ == "__main__":
    a = sys.argv[1:]
    print(json.dumps(run(int(a[0]) if len(a) > 0 else 4096, int(a[1]) if len(a) > 1 else 10, int(a[2]) if len(a) > 2 else 20000,
                         float(a[3]) if len(a) > 3 else 0.5, int(a[4]) if len(a) > 4 else 2, float(a[5]) if len(a) > 5 else None)))
