"""Batched host-side mirror of the reference's solver adapters.

The reference wraps its solver in two C++ classes, ``FORCESNormal`` / ``FORCESFinal``
(plan_manage/include/plan_manage/nmpc_utils.h:61-106) with three methods each:

  setParasNormal / setParasFinal   forces_normal.cpp:36-52   weights -> all_parameters
  solveNormal    / solveFinal      forces_normal.cpp:55-140  pack xinit / x0 / all_parameters, solve
  updateNormal   / updateFinal     forces_normal.cpp:142-168 output.x01..x20 -> mpc_output

plus the receding-horizon bookkeeping of ``NMPCSolver`` around them
(nmpc_solver.cpp:265-286 cold start, :524-543 yaw wrap + terminal duplication).

This module restates that packing for a batch of B independent problems (numpy, FP64) with the
same argument meaning: ``mpc_output`` is the (N+1)-row plan deque, ``ellipsoid_matrices`` are the
per-stage tube matrices E_i, polytopes are given as (A, b, nfaces).  The arithmetic on the hot
path is NOT here: the packed arrays go to the HIP solver through the C-ABI
(forces_resilient_planner_amd.solver).  The C++ twin of this file for reference-side callers is
csrc/frp_adapter.hpp.
"""
from __future__ import annotations

import math

import numpy as np

from . import layout as L


def init_mpc_output(state: np.ndarray, n: int = L.N_REF, thrust: float = 7.3) -> np.ndarray:
    """Cold-start plan: every row = [0,0,0,T, 0,0,0,T, state] (nmpc_solver.cpp:265-286; the hover
    guess 7.3 N is ``real_thrust_c_``, nmpc_utils.h:191).  state: [B, 9] -> [B, n+1, 17]."""
    state = np.asarray(state, dtype=np.float64)
    b = state.shape[0]
    row = np.zeros((b, L.NZ))
    row[:, 3] = thrust
    row[:, 7] = thrust
    row[:, 8:] = state
    return np.repeat(row[:, None, :], n + 1, axis=1)


class ForcesAdapter:
    """Batched FORCESNormal (model=MODEL_NORMAL) / FORCESFinal (model=MODEL_FINAL)."""

    def __init__(self, batch: int, model: int = L.MODEL_NORMAL, horizon: int = L.N_REF,
                 num_const: int = L.NH_REF):
        self.B, self.N, self.M, self.model = batch, horizon, num_const, model
        self.np_ = L.npar(num_const)
        self.xinit = np.zeros((batch, L.NX))
        self.x0 = np.zeros((batch, horizon, L.NZ))
        self.all_parameters = np.zeros((batch, horizon, self.np_))
        self.nfaces = np.zeros((batch, horizon), dtype=np.int32)
        self.output = np.zeros((batch, horizon, L.NZ))
        self.exitflag = np.zeros(batch, dtype=np.int32)

    # forces_normal.cpp:36-52 / forces_final.cpp:36-51
    def set_paras(self, w_stage_wp, w_stage_input, w_input_rate, w_terminal_wp, w_terminal_input):
        p = self.all_parameters
        p[:, :, 6] = w_stage_wp
        p[:, :, 7] = w_stage_input
        p[:, :, 8] = w_input_rate
        p[:, -1, 6] = w_terminal_wp
        p[:, -1, 7] = w_terminal_input

    # forces_normal.cpp:55-136 (everything before the solver call)
    def pack(self, mpc_output, external_acc, ref_total_pos, ref_total_yaw, ellipsoid_matrices,
             poly_A, poly_b, poly_nfaces):
        """mpc_output [B,N+1,17]; external_acc [B,3] (or [B,N,3]: the per-stage parameter layout
        allows it, setup.m:62); ref_total_pos [B,N,3]; ref_total_yaw [B,N];
        ellipsoid_matrices [B,N,3,3]; poly_A [B,N,F,3]; poly_b [B,N,F]; poly_nfaces [B,N]."""
        B, N, M = self.B, self.N, self.M
        mpc_output = np.asarray(mpc_output, dtype=np.float64)
        self.xinit[:] = mpc_output[:, 1, 8:17]                       # :62-72
        self.x0[:] = mpc_output[:, 1:N + 1, :]                       # :74-97 (shifted warm start)
        p = self.all_parameters
        p[:, :, 0:3] = ref_total_pos                                 # :99-102
        ext = np.asarray(external_acc, dtype=np.float64)
        p[:, :, 3:6] = ext[:, None, :] if ext.ndim == 2 else ext     # :103-106
        p[:, :, 9] = ref_total_yaw                                   # :107-108
        F = poly_A.shape[2]
        nf = np.minimum(np.asarray(poly_nfaces, dtype=np.int32), M)  # faces beyond M dropped (:114)
        keep = min(F, M)
        live = (np.arange(keep)[None, None, :] < nf[:, :, None])
        A = np.where(live[..., None], poly_A[:, :, :keep, :], 0.0)
        # robust tightening b_j - ||E a_j||_2 (:124-125)
        Ea = np.einsum('bnij,bnfj->bnfi', ellipsoid_matrices, A)
        bt = np.where(live, poly_b[:, :, :keep] - np.linalg.norm(Ea, axis=-1), 0.0)
        p[:, :, L.NPRE:] = 0.0                                       # padded rows (:127-135)
        p[:, :, L.NPRE:L.NPRE + 3 * keep] = A.reshape(B, N, 3 * keep)
        p[:, :, L.NPRE + 3 * M:L.NPRE + 3 * M + keep] = bt
        self.nfaces[:] = nf
        return self.xinit, self.x0, self.all_parameters, self.nfaces

    # forces_normal.cpp:142-168
    def update(self, mpc_output):
        mpc_output[:, :self.N, :] = self.output
        return mpc_output


def update_forces_results(mpc_output: np.ndarray) -> np.ndarray:
    """nmpc_solver.cpp:524-543: wrap yaw of rows 0..N-1 into [-pi, pi], duplicate the last row."""
    n = mpc_output.shape[1] - 1
    yaw = mpc_output[:, :n, 16]
    yaw = np.where(yaw < -math.pi, yaw + 2 * math.pi, np.where(yaw > math.pi, yaw - 2 * math.pi, yaw))
    mpc_output[:, :n, 16] = yaw
    mpc_output[:, n, :] = mpc_output[:, n - 1, :]
    return mpc_output
