"""Receding-horizon loop around the batched solver (BASELINE.json configs[4]).

Mirrors one NMPC tick of the reference for B planners at once:
  NMPCSolver::solveNMPC (nmpc_solver.cpp:351-482):  [cold start if the last exit flag != 1 (:363-364)]
    -> FORCESNormal::solveNormal packing (forces_normal.cpp:62-136, shift warm start)
    -> solve -> updateNormal (:142-168) -> updateFORCESResults (nmpc_solver.cpp:524-543).
The reference advances time through odometry; here the plant is the model itself: the state fed to the
next tick is the plan's own stage-2 state (exactly what forces_normal.cpp:62-72 uses as xinit).
"""
from __future__ import annotations

import numpy as np

from . import layout as L
from .adapter import ForcesAdapter, init_mpc_output, update_forces_results


def run(w0, ticks, solve_fn, tube_fn=None):
    """w0: workloads.config4_nominal(...) dict; solve_fn(workload_dict) -> (z [B,N,17], exitflag [B], iters [B]).
    tube_fn(plans [B,N,17]) -> E [B,N,3,3], when given, recomputes the tube from the current plans every tick as
    NMPCSolver::setFORCESParams does (nmpc_solver.cpp:484-521); otherwise the generator's fixed E is used.
    Returns per-tick exit flags / iteration counts and the final plan deque."""
    B, N, M, model = w0["B"], w0["N"], w0["M"], w0["model"]
    ad = ForcesAdapter(B, model, N, M)
    ad.all_parameters[:] = w0["params"]                    # weights set by the generator
    mpc = w0["mpc_output"].copy()
    ref_long = w0["ref_long"]                              # [1, N + ticks, 3]
    yaw = float(w0["heading"][0])
    flags, iters = [], []
    last_ok = np.ones(B, dtype=bool)
    for t in range(ticks):
        ref_pos = np.repeat(ref_long[:, t:t + N], B, 0)
        ref_yaw = np.full((B, N), yaw)
        A, b = w0["poly_A"], w0["poly_b"]
        if t > 0:
            # corridor boxes follow the reference (same construction as the generator)
            from .workloads import _bbox_faces
            A, b = _bbox_faces(ref_pos, ref_yaw)
        # cold start for planners whose last solve failed (nmpc_solver.cpp:363-364)
        if not last_ok.all():
            bad = ~last_ok
            mpc[bad] = init_mpc_output(mpc[bad][:, 1, 8:17], N)
        E = w0["E"] if tube_fn is None else tube_fn(mpc[:, :N])
        xinit, x0, params, nf = ad.pack(mpc, w0["f_ext"], ref_pos, ref_yaw, E, A, b, np.full((B, N), 6, np.int32))
        w = dict(xinit=xinit, x0=x0, params=params, nfaces=nf, N=N, M=M, model=model, B=B)
        z, fl, it = solve_fn(w)
        ok = fl == 1
        ad.output[:] = z
        upd = mpc.copy()
        ad.update(upd)
        upd = update_forces_results(upd)
        mpc[ok] = upd[ok]                                   # results are used only on exit flag 1 (:398-429)
        last_ok = ok
        flags.append(fl.copy()); iters.append(it.copy())
    return np.array(flags), np.array(iters), mpc


def run_device(w0, ticks, device="cuda:0", collect=True, propagate_tube=False):
    """The same loop with every per-tick step on the GPU: device-side packing, solve, device-side
    result bookkeeping, and with propagate_tube also the tube propagation (row f-2); only the (shared) nominal reference / corridor of configs[4] is produced on the host, one
    problem's worth per tick, and broadcast on the device.  Returns (flags [ticks,B], iters [ticks,B], final plan,
    seconds of GPU time for all ticks)."""
    import torch
    from . import solver
    from .workloads import _bbox_faces, _weights
    B, N, M, model = w0["B"], w0["N"], w0["M"], w0["model"]
    fleet = solver.DeviceFleet(B, N, M, 6, model, _weights(model), device)
    dev = fleet.solver.device
    fleet.mpc_output.copy_(fleet.to_device(w0["mpc_output"]))
    fleet.ellipsoid.copy_(fleet.to_device(w0["E"]))
    fleet.poly_nfaces.fill_(6)
    f_ext = fleet.to_device(w0["f_ext"])
    ref_long = w0["ref_long"]; yaw = float(w0["heading"][0])
    ref_yaw = torch.full((B, N), yaw, dtype=torch.float64, device=dev)
    flags = torch.zeros((ticks, B), dtype=torch.int32, device=dev)
    iters = torch.zeros((ticks, B), dtype=torch.int32, device=dev)
    # per-tick shared reference and corridor (one problem's worth), prepared up front
    refs, As, bs = [], [], []
    for t in range(ticks):
        r1 = ref_long[:, t:t + N]
        A1, b1 = _bbox_faces(r1, np.full((1, N), yaw))
        refs.append(fleet.to_device(r1)); As.append(fleet.to_device(A1)); bs.append(fleet.to_device(b1))
    cold_thrust = 7.3
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for t in range(ticks):
        ref_pos = refs[t].expand(B, N, 3).contiguous()
        fleet.poly_A.copy_(As[t].expand(B, N, 6, 3))
        fleet.poly_b.copy_(bs[t].expand(B, N, 6))
        if t > 0:  # cold start for planners whose last solve failed (nmpc_solver.cpp:363-364), on the device
            fleet.coldstart(thrust=cold_thrust)
        fleet.tick(f_ext, ref_pos, ref_yaw, propagate_tube=propagate_tube)
        if collect:
            flags[t] = fleet.solver.exitflag; iters[t] = fleet.solver.iters
    ev1.record()
    torch.cuda.synchronize(dev)
    return flags.cpu().numpy(), iters.cpu().numpy(), fleet.mpc_output.cpu().numpy(), ev0.elapsed_time(ev1) * 1e-3
