"""ORACLE for SURVEY 8f row f-4, first half (stage references) -- TEST INFRASTRUCTURE ONLY.

Only tests/ (including tests/tools/) and __graft_entry__.smoke() may import this file; the product path
(forces_resilient_planner_amd/csrc/frp_reference.hip) never does.

Plain-Python restatement of NMPCSolver::getCurTraj (plan_manage/src/nmpc_solver.cpp:109-142) and
NMPCSolver::calculate_yaw (:834-862) as called by setFORCESParams (:486, :493-495), statement by statement.
PARITY UNPINNED: nmpc_solver.cpp needs ROS and Eigen and the reference holds no vectors for these functions; they
are two dozen lines of scalar arithmetic, and the GPU kernel is tested against this file at 1e-12.
"""
import math

import numpy as np

PI = 3.1415926  # nmpc_solver.cpp:3


def references_one(kino_path, kino_size, time_offset, mpc_output, N, Ts=0.05):
    """kino_path [K,3]; mpc_output [N+1,17].  Returns (ref_pos [N,3], ref_yaw [N], replan flag)."""
    ref_pos = np.zeros((N, 3)); ref_yaw = np.zeros(N)
    last_yaw = float(mpc_output[1, 16])  # :486
    replan = False
    for index in range(N):
        index_time = index * Ts + time_offset
        kino_index = int(index_time / Ts) & 0xFFFFFFFF  # (unsigned int)(int)(...)
        if kino_index + 1 < kino_size:
            pos = kino_path[kino_index] + math.fmod(index_time, Ts) / Ts * (kino_path[kino_index + 1] - kino_path[kino_index])
        else:
            pos = kino_path[kino_size - 1]
        fwd = kino_path[kino_index + 5] if kino_index + 5 < kino_size else kino_path[kino_size - 1]
        # calculate_yaw
        d = fwd - pos
        yaw_temp = math.atan2(d[1], d[0]) if np.linalg.norm(d) > 0.1 else last_yaw
        if abs(yaw_temp - last_yaw) > PI:
            yaw = yaw_temp - 2 * PI if yaw_temp > 0 else yaw_temp + 2 * PI
        else:
            yaw = yaw_temp
        yaw = 0.2 * last_yaw + 0.8 * yaw
        last_yaw = yaw
        ref_pos[index] = pos; ref_yaw[index] = yaw
        if index == 0 and np.linalg.norm(pos - mpc_output[1, 8:11]) > 1.0:
            replan = True
    return ref_pos, ref_yaw, replan
