"""Index maps and constants of the NMPC stage problem.

Mirrors the reference's problem definition (all paths under
src/resilient_planner/plan_manage/ of the reference):
  matlab_code/setup.m:17-66            dimensions, index maps, physical constants
  matlab_code/mpc/normal/mpc_generator_normal.m:33-46   variable bounds
  solver/normal/FORCESNLPsolver_normal/include/FORCESNLPsolver_normal.h:153-236  ABI sizes
"""
import math

import numpy as np

NU = 4          # rollrate, pitchrate, yawrate, thrust           (setup.m:48)
NW = 4          # previous-input copy                             (setup.m:38: nin = 8)
NX = 9          # pos(3) vel(3) euler(3)                          (setup.m:39)
NZ = 17         # stage vector                                    (setup.m:40)
NEQ = 13        # equality rows per stage                         (setup.m:41)
NPRE = 10       # ref(3) f_ext(3) weights(3) yaw_ref              (setup.m:60-64)
NH_REF = 30     # corridor rows in the reference ABI              (setup.m:42)
N_REF = 20      # horizon of the generated reference solver       (setup.m:36)
DT = 0.05       # setup.m:37
MASS = 0.745319  # setup.m:17
GRAV = 9.81     # setup.m:18
HU = 1e-5       # corridor upper bound (mpc_generator_normal.m:14)

MODEL_NORMAL = 0
MODEL_FINAL = 1

# z index ranges (0-based)
Z_U = slice(0, 4)
Z_W = slice(4, 8)
Z_POS = slice(8, 11)
Z_VEL = slice(11, 14)
Z_EUL = slice(14, 17)


def npar(m: int) -> int:
    """Parameters per stage with m corridor rows (setup.m:43 with nh = m)."""
    return NPRE + 4 * m


def bounds():
    """lb, ub of the stage vector (mpc_generator_normal.m:33-46, setup.m:21-31)."""
    r = math.pi / 2
    tmax, tmin = 2.0 * GRAV * MASS, 0.5 * GRAV * MASS
    ub = np.array([r, r, r, tmax, r, r, r, tmax, 20, 20, 5, 2, 2, 2,
                   0.4 * math.pi, 0.4 * math.pi, 2 * math.pi], dtype=np.float64)
    lb = np.array([-r, -r, -r, tmin, -r, -r, -r, tmin, -20, -20, 0, -2, -2, -2,
                   -0.4 * math.pi, -0.4 * math.pi, -2 * math.pi], dtype=np.float64)
    return lb, ub


# exit flags (FORCESNLPsolver_normal.h:110-139)
OPTIMAL = 1
MAXIT = 0
FACTORIZATION_ERROR = -5
BADFUNCEVAL = -6
NOPROGRESS = -7
PARAM_VALUE_ERROR = -11
