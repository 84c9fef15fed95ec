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


def run(w0, ticks, solve_fn):
    """w0: workloads.config4_nominal(...) dict; solve_fn(workload_dict) -> (z [B,N,17], exitflag [B], iters [B]).
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
        xinit, x0, params, nf = ad.pack(mpc, w0["f_ext"], ref_pos, ref_yaw, w0["E"], A, b, np.full((B, N), 6, np.int32))
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
