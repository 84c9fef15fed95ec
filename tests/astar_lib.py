"""ctypes loader for the kinodynamic-A* oracle (oracle/astar_oracle.c in liboracle.so) -- test infrastructure only."""
import ctypes

import numpy as np

from . import oracle_lib as OL

D = ctypes.POINTER(ctypes.c_double)
IP = ctypes.POINTER(ctypes.c_int)
MAX_PATH = 256


class AstarParams(ctypes.Structure):
    _fields_ = [("occ", ctypes.c_void_p), ("grid", ctypes.c_int * 3), ("origin", ctypes.c_double * 3), ("map_size", ctypes.c_double * 3),
                ("resolution", ctypes.c_double), ("use_local", ctypes.c_int), ("local_min", ctypes.c_int * 3), ("local_max", ctypes.c_int * 3),
                ("ego_r", ctypes.c_double), ("ego_h", ctypes.c_double),
                ("max_tau", ctypes.c_double), ("init_max_tau", ctypes.c_double), ("max_vel", ctypes.c_double), ("max_acc", ctypes.c_double),
                ("w_time", ctypes.c_double), ("horizon", ctypes.c_double), ("lambda_heu", ctypes.c_double),
                ("allocate_num", ctypes.c_int), ("check_num", ctypes.c_int), ("tie_breaker", ctypes.c_double), ("max_expand", ctypes.c_int)]


class AstarResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int), ("use_node_num", ctypes.c_int), ("iter_num", ctypes.c_int), ("is_shot_succ", ctypes.c_int),
                ("n_path", ctypes.c_int), ("coef_shot", ctypes.c_double * 12), ("t_shot", ctypes.c_double),
                ("path_state", (ctypes.c_double * 6) * MAX_PATH), ("path_input", (ctypes.c_double * 3) * MAX_PATH),
                ("path_duration", ctypes.c_double * MAX_PATH), ("path_node", ctypes.c_int * MAX_PATH)]


def lib():
    l = OL.lib()
    for n in ("orc_det_cbrt", "orc_det_acos", "orc_det_cos"):
        getattr(l, n).restype = ctypes.c_double
        getattr(l, n).argtypes = [ctypes.c_double]
    return l


def make_params(world):
    """world: dict from forces_resilient_planner_amd.workloads.astar_world (occ uint8 [nx,ny,nz] + map / search constants)."""
    p = AstarParams()
    occ = np.ascontiguousarray(world["occ"], dtype=np.uint8)
    p._occ_keepalive = occ
    p.occ = occ.ctypes.data
    p.grid[:] = occ.shape; p.origin[:] = world["origin"]; p.map_size[:] = world["map_size"]; p.resolution = world["resolution"]
    p.use_local = 0
    p.ego_r = world["ego_r"]; p.ego_h = world["ego_h"]
    for k in ("max_tau", "init_max_tau", "max_vel", "max_acc", "w_time", "horizon", "lambda_heu", "allocate_num", "check_num", "tie_breaker"):
        setattr(p, k, world[k])
    p.max_expand = 4096
    return p


def plan_batch(world, start_pt, start_v, start_a, end_pt, end_v, f_ext, init=True, Ts=0.05, cap=2048, nthreads=0, retry_pt=None, retry_v=None):
    """NMPCSolver::getKinoPath's search + getKinoTraj for B planners on the CPU oracle.
    Returns dict(status [B], kino_path [B,cap,3], kino_size [B], retried [B], results [B] AstarResult)."""
    B = start_pt.shape[0]
    p = make_params(world)
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    sp, sv, sa, ep, ev, fe = map(c, (start_pt, start_v, start_a, end_pt, end_v, f_ext))
    path = np.zeros((B, cap, 3)); size = np.zeros(B, dtype=np.int32); status = np.zeros(B, dtype=np.int32); retried = np.zeros(B, dtype=np.int32)
    res = (AstarResult * B)()
    lib().orc_astar_batch(B, ctypes.byref(p), OL.P(sp), OL.P(sv), OL.P(sa), OL.P(ep), OL.P(ev), 1 if init else 0, OL.P(fe), ctypes.c_double(Ts),
                          OL.P(path), cap, size.ctypes.data_as(IP), status.ctypes.data_as(IP), res, retried.ctypes.data_as(IP), nthreads,
                          OL.P(c(retry_pt)) if retry_pt is not None else None, OL.P(c(retry_v)) if retry_v is not None else None)
    return dict(status=status, kino_path=path, kino_size=size, retried=retried, results=res)
