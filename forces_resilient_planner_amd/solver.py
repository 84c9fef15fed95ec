"""ctypes binding of the C-ABI in include/frp_nmpc.h (libfrp_nmpc_amd.so).

The product path: every function here runs HIP kernels on the MI355X; there is no CPU fallback --
if the library is missing or no device is visible the calls raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import layout as L

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRP_LIB") or os.path.join(_PKG, "libfrp_nmpc_amd.so")  # FRP_LIB: an experiment build (tools/build_variant.sh)
INFO_STRIDE = 12
ABI_VERSION = 6  # FRP_NMPC_ABI_VERSION of include/frp_nmpc.h

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)


class Options(ctypes.Structure):
    _fields_ = [("maxit", ctypes.c_int), ("tol_stat", ctypes.c_double), ("tol_eq", ctypes.c_double),
                ("tol_ineq", ctypes.c_double), ("tol_comp", ctypes.c_double), ("mu0", ctypes.c_double),
                ("ftb", ctypes.c_double), ("hessian", ctypes.c_int), ("diverge_mu", ctypes.c_double), ("twist", ctypes.c_int)]


class Batch(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("N", ctypes.c_int), ("M", ctypes.c_int), ("MF", ctypes.c_int),
                ("model", ctypes.c_int),
                ("xinit", ctypes.c_void_p), ("x0", ctypes.c_void_p), ("params", ctypes.c_void_p),
                ("nfaces", ctypes.c_void_p), ("z", ctypes.c_void_p), ("exitflag", ctypes.c_void_p),
                ("iters", ctypes.c_void_p), ("info", ctypes.c_void_p), ("model_per_problem", ctypes.c_void_p),
                ("order_hint", ctypes.c_void_p)]


class Pack(ctypes.Structure):  # frp_nmpc_pack (include/frp_nmpc.h)
    _fields_ = [("B", ctypes.c_int), ("N", ctypes.c_int), ("M", ctypes.c_int), ("NPOLY", ctypes.c_int), ("F", ctypes.c_int),
                ("external_acc_per_stage", ctypes.c_int), ("mpc_output", ctypes.c_void_p), ("external_acc", ctypes.c_void_p), ("ref_pos", ctypes.c_void_p),
                ("ref_yaw", ctypes.c_void_p), ("ellipsoid", ctypes.c_void_p), ("poly_A", ctypes.c_void_p),
                ("poly_b", ctypes.c_void_p), ("poly_nfaces", ctypes.c_void_p), ("poly_index", ctypes.c_void_p),
                ("w_stage_wp", ctypes.c_double), ("w_stage_input", ctypes.c_double), ("w_input_rate", ctypes.c_double),
                ("w_terminal_wp", ctypes.c_double), ("w_terminal_input", ctypes.c_double),
                ("xinit", ctypes.c_void_p), ("x0", ctypes.c_void_p), ("params", ctypes.c_void_p), ("nfaces", ctypes.c_void_p),
                ("mode", ctypes.c_void_p), ("wf_stage_wp", ctypes.c_double), ("wf_stage_input", ctypes.c_double),
                ("wf_input_rate", ctypes.c_double), ("wf_terminal_wp", ctypes.c_double), ("wf_terminal_input", ctypes.c_double),
                ("padded_rows_are_zero", ctypes.c_int)]


class Tube(ctypes.Structure):  # frp_nmpc_tube (include/frp_nmpc.h)
    _fields_ = [("B", ctypes.c_int), ("N", ctypes.c_int), ("mpc_output", ctypes.c_void_p),
                ("mass", ctypes.c_double), ("drag", ctypes.c_double), ("ego_r", ctypes.c_double), ("ego_h", ctypes.c_double),
                ("noise", ctypes.c_double * 3), ("epsilon", ctypes.c_double), ("Ts", ctypes.c_double),
                ("ellipsoid", ctypes.c_void_p)]


class Corridor(ctypes.Structure):  # frp_nmpc_corridor (include/frp_nmpc.h)
    _fields_ = [("B", ctypes.c_int), ("N", ctypes.c_int), ("F", ctypes.c_int), ("P", ctypes.c_int),
                ("cloud", ctypes.c_void_p), ("cloud_per_planner", ctypes.c_int), ("cloud_count", ctypes.c_void_p),
                ("ref_pos", ctypes.c_void_p), ("ref_yaw", ctypes.c_void_p), ("ellipsoid", ctypes.c_void_p),
                ("bbox", ctypes.c_double * 3), ("seed_len", ctypes.c_double), ("inflation", ctypes.c_double),
                ("offset_x", ctypes.c_double),
                ("poly_A", ctypes.c_void_p), ("poly_b", ctypes.c_void_p), ("poly_nfaces", ctypes.c_void_p),
                ("poly_index", ctypes.c_void_p), ("poly_count", ctypes.c_void_p),
                ("grid_origin", ctypes.c_double * 3), ("grid_cell", ctypes.c_double), ("grid_dims", ctypes.c_int * 3),
                ("grid_points", ctypes.c_void_p), ("grid_index", ctypes.c_void_p), ("grid_start", ctypes.c_void_p)]


class Reference(ctypes.Structure):  # frp_nmpc_reference (include/frp_nmpc.h)
    _fields_ = [("B", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("kino_path", ctypes.c_void_p),
                ("path_per_planner", ctypes.c_int), ("kino_size", ctypes.c_void_p), ("time_offset", ctypes.c_void_p),
                ("mpc_output", ctypes.c_void_p), ("Ts", ctypes.c_double), ("pi", ctypes.c_double),
                ("ref_pos", ctypes.c_void_p), ("ref_yaw", ctypes.c_void_p), ("replan", ctypes.c_void_p)]


class Astar(ctypes.Structure):  # frp_nmpc_astar (include/frp_nmpc.h)
    _fields_ = [("B", ctypes.c_int), ("occ", ctypes.c_void_p), ("grid", ctypes.c_int * 3), ("origin", ctypes.c_double * 3),
                ("map_size", ctypes.c_double * 3), ("resolution", ctypes.c_double), ("local_box", ctypes.c_void_p),
                ("ego_r", ctypes.c_double), ("ego_h", ctypes.c_double),
                ("max_tau", ctypes.c_double), ("init_max_tau", ctypes.c_double), ("max_vel", ctypes.c_double), ("max_acc", ctypes.c_double),
                ("w_time", ctypes.c_double), ("horizon", ctypes.c_double), ("lambda_heu", ctypes.c_double), ("tie_breaker", ctypes.c_double),
                ("allocate_num", ctypes.c_int), ("check_num", ctypes.c_int),
                ("start_pt", ctypes.c_void_p), ("start_vel", ctypes.c_void_p), ("start_acc", ctypes.c_void_p), ("end_pt", ctypes.c_void_p),
                ("end_vel", ctypes.c_void_p), ("external_acc", ctypes.c_void_p), ("active", ctypes.c_void_p), ("init_search", ctypes.c_int),
                ("Ts", ctypes.c_double), ("K", ctypes.c_int),
                ("kino_path", ctypes.c_void_p), ("kino_size", ctypes.c_void_p), ("status", ctypes.c_void_p), ("stats", ctypes.c_void_p),
                ("path_nodes", ctypes.c_void_p), ("retry_pt", ctypes.c_void_p), ("retry_vel", ctypes.c_void_p)]


ASTAR_MAX_PATH = 256
ASTAR_REACH_HORIZON, ASTAR_REACH_END, ASTAR_NO_PATH, ASTAR_REACH_END_BUT_SHOT_FAILS = 1, 2, 3, 4

REFERENCE_PI = 3.1415926  # nmpc_solver.cpp:3

# getSikangConst's constants (nmpc_solver.cpp:302, :318, :323)
CORRIDOR_DEFAULTS = dict(bbox=(2.0, 2.0, 1.0), seed_len=0.1, inflation=1.1, offset_x=0.0)
CORRIDOR_MAX_F = 64

# ROS parameter defaults of the tube model (nmpc_solver.cpp:68-74, nmpc_utils.h:188-189)
TUBE_DEFAULTS = dict(mass=0.74, drag=0.33, ego_r=0.27, ego_h=0.0425, noise=(0.5, 0.5, 0.5), epsilon=0.06, Ts=0.05)


class ForcesParams(ctypes.Structure):
    _fields_ = [("xinit", ctypes.c_double * 9), ("x0", ctypes.c_double * 340),
                ("all_parameters", ctypes.c_double * 2600), ("num_of_threads", ctypes.c_uint)]


class ForcesOutput(ctypes.Structure):
    _fields_ = [("x", (ctypes.c_double * 17) * 20)]


class ForcesInfo(ctypes.Structure):
    _fields_ = [("it", ctypes.c_int), ("it2opt", ctypes.c_int)] + \
               [(n, ctypes.c_double) for n in "res_eq res_ineq rsnorm rcompnorm pobj dobj dgap rdgap mu mu_aff sigma".split()] + \
               [("lsit_aff", ctypes.c_int), ("lsit_cc", ctypes.c_int)] + \
               [(n, ctypes.c_double) for n in "step_aff step_cc solvetime fevalstime".split()]


EXTFUNC = ctypes.CFUNCTYPE(None, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                           c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int)

EXPORTS = ["frp_nmpc_default_options", "frp_nmpc_workspace_bytes", "frp_nmpc_solve_batch",
           "frp_nmpc_solve_batch_host", "frp_nmpc_stage_eval", "frp_nmpc_stage_eval_host", "frp_nmpc_time_solve",
           "frp_nmpc_version", "frp_nmpc_device_count", "FORCESNLPsolver_normal_solve",
           "FORCESNLPsolver_final_solve", "frp_nmpc_pack_batch", "frp_nmpc_update_batch", "frp_nmpc_tube_batch",
           "frp_nmpc_corridor_batch", "frp_nmpc_reference_batch",
           "frp_nmpc_coldstart_batch", "frp_nmpc_cloud_grid_build",
           "frp_nmpc_mode_batch", "frp_nmpc_astar_batch", "frp_nmpc_astar_workspace_bytes",
           "frp_nmpc_kernel_timing_begin", "frp_nmpc_kernel_timing_end", "frp_nmpc_set_q4_min_batch",
           "frp_nmpc_abi_version", "frp_nmpc_abi_check", "frp_nmpc_host_register", "frp_nmpc_host_unregister",
           "frp_nmpc_host_registered", "frp_nmpc_host_unregister_all", "frp_nmpc_solve_batch_host_begin", "frp_nmpc_solve_batch_host_wait"]

_lib = None


def lib():
    """Load libfrp_nmpc_amd.so (built in-tree by forces_resilient_planner_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch ships its own HIP runtime; the library links the system one.  Both coexist in one process as long
        # as torch's is initialised first (the reverse order leaves torch with "No HIP GPUs are available"), and
        # DeviceSolver / DeviceFleet hand torch-owned HBM to the library, so initialise torch here if it has a GPU.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        l = ctypes.CDLL(LIB_PATH)
        # the ctypes mirrors below against the library that was actually loaded (include/frp_nmpc.h: FRP_NMPC_ABI_CHECK): a stale
        # library would write another info stride into our arrays / read short option structs
        if not hasattr(l, "frp_nmpc_abi_check"):
            raise RuntimeError(f"{LIB_PATH} predates the ABI check (frp_nmpc.h ABI {ABI_VERSION}): rebuild it")
        l.frp_nmpc_abi_check.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        if l.frp_nmpc_abi_check(ABI_VERSION, ctypes.sizeof(Options), ctypes.sizeof(Batch), INFO_STRIDE) != 0:
            raise RuntimeError(f"{LIB_PATH} was built from another include/frp_nmpc.h than these bindings (ABI {ABI_VERSION}, "
                               f"options {ctypes.sizeof(Options)} B, batch {ctypes.sizeof(Batch)} B, info stride {INFO_STRIDE})")
        l.frp_nmpc_workspace_bytes.restype = ctypes.c_size_t
        l.frp_nmpc_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        l.frp_nmpc_version.restype = ctypes.c_char_p
        l.frp_nmpc_solve_batch.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(Options), ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p]
        l.frp_nmpc_time_solve.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(Options), ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        l.frp_nmpc_solve_batch_host.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(Options)]
        l.frp_nmpc_set_q4_min_batch.argtypes = [ctypes.c_int]
        l.frp_nmpc_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        l.frp_nmpc_host_unregister.argtypes = [ctypes.c_void_p]
        l.frp_nmpc_host_registered.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        l.frp_nmpc_solve_batch_host_begin.argtypes = [ctypes.POINTER(Batch), ctypes.POINTER(Options), ctypes.POINTER(ctypes.c_int)]
        l.frp_nmpc_solve_batch_host_wait.argtypes = [ctypes.c_int]
        l.frp_nmpc_kernel_timing_begin.argtypes = [ctypes.c_int, ctypes.c_int]
        l.frp_nmpc_kernel_timing_end.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
        l.frp_nmpc_pack_batch.argtypes = [ctypes.POINTER(Pack), ctypes.c_void_p]
        l.frp_nmpc_update_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p]
        l.frp_nmpc_tube_batch.argtypes = [ctypes.POINTER(Tube), ctypes.c_void_p]
        l.frp_nmpc_corridor_batch.argtypes = [ctypes.POINTER(Corridor), ctypes.c_void_p]
        l.frp_nmpc_reference_batch.argtypes = [ctypes.POINTER(Reference), ctypes.c_void_p]
        l.frp_nmpc_cloud_grid_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        l.frp_nmpc_mode_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        l.frp_nmpc_coldstart_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                               ctypes.c_void_p, ctypes.c_void_p]
        l.frp_nmpc_astar_workspace_bytes.restype = ctypes.c_size_t
        l.frp_nmpc_astar_workspace_bytes.argtypes = [ctypes.POINTER(Astar)]
        l.frp_nmpc_astar_batch.argtypes = [ctypes.POINTER(Astar), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        _lib = l
    return _lib


def default_options(**kw) -> Options:
    o = Options()
    lib().frp_nmpc_default_options(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with frp error {rc} (no HIP device / HIP error / bad argument; "
                           "this library has no CPU path)")


def solve_batch_host(w, opt: Options | None = None, MF: int | None = None, x0=None, out=None):
    """Solve a workload dict (host numpy arrays) on the GPU through frp_nmpc_solve_batch_host.
    out: (z, flag, iters, info) arrays of an earlier call to write into (a caller in a loop does not re-allocate)."""
    B, N, M = int(w["xinit"].shape[0]), int(w["N"]), int(w["M"])
    xinit = np.ascontiguousarray(w["xinit"], dtype=np.float64)
    z0 = np.ascontiguousarray(w["x0"] if x0 is None else x0, dtype=np.float64)
    params = np.ascontiguousarray(w["params"], dtype=np.float64)
    nf = None if w.get("nfaces") is None else np.ascontiguousarray(w["nfaces"], dtype=np.int32)
    if MF is None:
        MF = int(nf.max()) if nf is not None and nf.size else M
    if out is not None:
        z, flag, iters, info = out
    else:
        z = np.zeros((B, N, L.NZ)); flag = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
        info = np.zeros((B, INFO_STRIDE))
    models = None if w.get("models") is None else np.ascontiguousarray(w["models"], dtype=np.int32)  # per-problem normal / final
    b = Batch(B, N, M, MF, int(w["model"]), xinit.ctypes.data, z0.ctypes.data, params.ctypes.data,
              nf.ctypes.data if nf is not None else None, z.ctypes.data, flag.ctypes.data, iters.ctypes.data,
              info.ctypes.data, models.ctypes.data if models is not None else None)
    _check(lib().frp_nmpc_solve_batch_host(ctypes.byref(b), ctypes.byref(opt) if opt is not None else None),
           "frp_nmpc_solve_batch_host")
    return z, flag, iters, info


def solve_batch_host_begin(w, out, opt: Options | None = None, MF: int | None = None):
    """frp_nmpc_solve_batch_host_begin: every array of `w` (C-contiguous float64 / int32) and of `out` = (z, flag, iters, info) registered with
    host_register; returns the ticket.  The outputs are valid after solve_batch_host_wait(ticket)."""
    B, N, M = int(w["xinit"].shape[0]), int(w["N"]), int(w["M"])
    nf = w.get("nfaces")
    if MF is None:
        MF = int(nf.max()) if nf is not None and nf.size else M
    z, flag, iters, info = out
    models = w.get("models")
    for a in (w["xinit"], w["x0"], w["params"], z, info):
        assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
    for a in (nf, models, flag, iters):
        assert a is None or (a.flags["C_CONTIGUOUS"] and a.dtype == np.int32)
    b = Batch(B, N, M, MF, int(w["model"]), w["xinit"].ctypes.data, w["x0"].ctypes.data, w["params"].ctypes.data,
              nf.ctypes.data if nf is not None else None, z.ctypes.data, flag.ctypes.data, iters.ctypes.data,
              info.ctypes.data, models.ctypes.data if models is not None else None)
    t = ctypes.c_int(-1)
    _check(lib().frp_nmpc_solve_batch_host_begin(ctypes.byref(b), ctypes.byref(opt) if opt is not None else None, ctypes.byref(t)),
           "frp_nmpc_solve_batch_host_begin")
    return int(t.value)


def solve_batch_host_wait(ticket):
    _check(lib().frp_nmpc_solve_batch_host_wait(int(ticket)), "frp_nmpc_solve_batch_host_wait")


_registered = {}  # ptr -> [array, count]: a strong reference for as long as the library holds the pages pinned under that address


def host_register(*arrays):
    """frp_nmpc_host_register for numpy arrays a caller reuses from call to call (inputs and the `out` arrays of solve_batch_host):
    with every array of a call registered, frp_nmpc_solve_batch_host stages nothing.  The wrapper keeps a reference to every
    registered array until host_unregister: an array that were garbage-collected while registered would leave its address range
    in the library's registry, and a later array placed at the same address would be taken for the old mapping."""
    for a in arrays:
        if a is not None:
            assert a.flags["C_CONTIGUOUS"]
            _check(lib().frp_nmpc_host_register(a.ctypes.data, a.nbytes), "frp_nmpc_host_register")
            ent = _registered.setdefault(a.ctypes.data, [a, 0])
            ent[1] += 1


def host_unregister(*arrays):
    for a in arrays:
        if a is not None:
            _check(lib().frp_nmpc_host_unregister(a.ctypes.data), "frp_nmpc_host_unregister")
            ent = _registered.get(a.ctypes.data)
            if ent is not None:
                ent[1] -= 1
                if ent[1] <= 0:
                    del _registered[a.ctypes.data]


def host_unregister_all():
    _check(lib().frp_nmpc_host_unregister_all(), "frp_nmpc_host_unregister_all")
    _registered.clear()


def stage_eval_host(z, params, M, model, want=("f", "gf", "c", "Jc", "h")):
    z = np.ascontiguousarray(z, dtype=np.float64); params = np.ascontiguousarray(params, dtype=np.float64)
    B, N = z.shape[0], z.shape[1]
    out = dict(f=np.zeros((B, N)), gf=np.zeros((B, N, 17)), c=np.zeros((B, N, 13)), Jc=np.zeros((B, N, 221)),
               h=np.zeros((B, N, max(M, 1))))
    ptr = lambda k: out[k].ctypes.data_as(c_double_p) if k in want else None
    _check(lib().frp_nmpc_stage_eval_host(B, N, M, model, z.ctypes.data_as(c_double_p), params.ctypes.data_as(c_double_p),
                                          ptr("f"), ptr("gf"), ptr("c"), ptr("Jc"), ptr("h")), "frp_nmpc_stage_eval_host")
    return out


class DeviceSolver:
    """Device-resident batched solver: torch tensors hold the HBM buffers, the C-ABI gets raw pointers."""

    def __init__(self, B, N, M, MF, model, device="cuda:0"):
        import torch
        self.torch = torch
        self.B, self.N, self.M, self.MF, self.model = B, N, M, MF, model
        self.device = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.xinit = torch.empty((B, L.NX), **f64)
        self.x0 = torch.empty((B, N, L.NZ), **f64)
        self.params = torch.empty((B, N, L.npar(M)), **f64)
        self.nfaces = torch.empty((B, N), dtype=torch.int32, device=self.device)
        self.z = torch.zeros((B, N, L.NZ), **f64)
        # zero = "no solve yet / MAXIT": DeviceFleet's cold start and update branch on exitflag == 1 before the first solve has
        # written it, so a fresh solver must not expose recycled allocator memory (the reference's !initialized_output_)
        self.exitflag = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.iters = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self.info = torch.empty((B, INFO_STRIDE), **f64)
        self.ws_bytes = int(lib().frp_nmpc_workspace_bytes(B, N, MF))
        self.ws = torch.empty((self.ws_bytes // 8 + 1,), **f64)
        self.use_nfaces = True
        # receding-horizon callers: queue the problems by the PREVIOUS solve's iteration counts (frp_nmpc_batch.order_hint);
        # off = by the objective of the initial guess
        self.order_by_last_iters = False
        self.opt = default_options()

    def upload(self, w):
        t = self.torch
        self._packed_by = None  # (a fleet's pack may no longer assume what it left in params / nfaces)
        self.xinit.copy_(t.from_numpy(np.ascontiguousarray(w["xinit"])))
        self.x0.copy_(t.from_numpy(np.ascontiguousarray(w["x0"])))
        self.params.copy_(t.from_numpy(np.ascontiguousarray(w["params"])))
        self.nfaces.copy_(t.from_numpy(np.ascontiguousarray(w["nfaces"], dtype=np.int32)))

    def _batch(self, lo=0, hi=None):
        """The problems [lo, hi) of this solver's buffers as a frp_nmpc_batch (the whole batch by default)."""
        hi = self.B if hi is None else hi
        models = getattr(self, "models", None)
        return Batch(hi - lo, self.N, self.M, self.MF, self.model, self.xinit[lo:].data_ptr(), self.x0[lo:].data_ptr(),
                     self.params[lo:].data_ptr(), self.nfaces[lo:].data_ptr() if self.use_nfaces else None, self.z[lo:].data_ptr(),
                     self.exitflag[lo:].data_ptr(), self.iters[lo:].data_ptr(), self.info[lo:].data_ptr(),
                     models[lo:].data_ptr() if models is not None else None,
                     self.iters[lo:].data_ptr() if self.order_by_last_iters else None)

    def solve_range(self, lo, hi, stream=None, piece=0):
        """Solve the problems [lo, hi) of the batch only (a piece of a shard whose later pieces are still arriving):
        asynchronous on `stream`.  Pieces that may run at the same time (different streams) pass different `piece` numbers:
        each gets a queue workspace of its own (the work-queue head lives there)."""
        if hi <= lo:
            return
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        if not hasattr(self, "_piece_ws"):
            self._piece_ws = {}
        if piece not in self._piece_ws:
            self._piece_ws[piece] = self.ws if piece == 0 else self.torch.empty_like(self.ws)
        ws = self._piece_ws[piece]
        b = self._batch(lo, hi)
        _check(lib().frp_nmpc_solve_batch(ctypes.byref(b), ctypes.byref(self.opt), ws.data_ptr(), self.ws_bytes,
                                          ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_solve_batch")

    def solve(self, stream=None):
        """Asynchronous launch on `stream` (a torch.cuda.Stream) or torch's current stream."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        b = self._batch()
        _check(lib().frp_nmpc_solve_batch(ctypes.byref(b), ctypes.byref(self.opt), self.ws.data_ptr(), self.ws_bytes,
                                          ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_solve_batch")

    def time_solve(self, reps, stream=None):
        """Average kernel duration (ms) over `reps` launches, HIP events on the launch stream."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        b = self._batch()
        ms = ctypes.c_float(0)
        _check(lib().frp_nmpc_time_solve(ctypes.byref(b), ctypes.byref(self.opt), self.ws.data_ptr(), self.ws_bytes,
                                         ctypes.c_void_p(s.cuda_stream), reps, ctypes.byref(ms)), "frp_nmpc_time_solve")
        return ms.value


def kernel_timing_begin(max_launches, stride=1):
    """From here to kernel_timing_end() every `stride`-th solve launch carries a hipEvent pair around its dominant kernel (on its own stream)."""
    _check(lib().frp_nmpc_kernel_timing_begin(int(max_launches), int(stride)), "frp_nmpc_kernel_timing_begin")


def kernel_timing_end():
    """(average ms of the dominant kernel over the launches since kernel_timing_begin, number of launches)."""
    ms, n = ctypes.c_float(0), ctypes.c_int(0)
    _check(lib().frp_nmpc_kernel_timing_end(ctypes.byref(ms), ctypes.byref(n)), "frp_nmpc_kernel_timing_end")
    return ms.value, n.value


def tube_batch_device(mpc_output, ellipsoid, consts=None, stream=None):
    """frp_nmpc_tube_batch on device tensors: mpc_output [B,N+1,17] f64 -> ellipsoid [B,N,3,3] f64 (in place)."""
    import torch
    c = dict(TUBE_DEFAULTS)
    c.update(consts or {})
    B, rows, nz = mpc_output.shape
    assert nz == L.NZ and mpc_output.is_contiguous() and ellipsoid.is_contiguous() and mpc_output.dtype == torch.float64
    assert tuple(ellipsoid.shape) == (B, rows - 1, 3, 3) and ellipsoid.dtype == torch.float64
    s = stream if stream is not None else torch.cuda.current_stream(mpc_output.device)
    tb = Tube(B, rows - 1, mpc_output.data_ptr(), c["mass"], c["drag"], c["ego_r"], c["ego_h"],
              (ctypes.c_double * 3)(*c["noise"]), c["epsilon"], c["Ts"], ellipsoid.data_ptr())
    _check(lib().frp_nmpc_tube_batch(ctypes.byref(tb), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_tube_batch")


def reference_batch_device(kino_path, time_offset, mpc_output, ref_pos, ref_yaw, replan=None, kino_size=None, Ts=0.05,
                           stream=None):
    """frp_nmpc_reference_batch on device tensors: kino_path [K,3] (shared) or [B,K,3]; time_offset [B];
    mpc_output [B,N+1,17] -> ref_pos [B,N,3], ref_yaw [B,N], replan [B] int32."""
    import torch
    B, N, _ = ref_pos.shape
    per = 1 if kino_path.dim() == 3 else 0
    K = kino_path.shape[-2]
    for t in (kino_path, time_offset, mpc_output, ref_pos, ref_yaw):
        assert t.is_contiguous() and t.dtype == torch.float64
    assert tuple(mpc_output.shape) == (B, N + 1, L.NZ) and tuple(ref_yaw.shape) == (B, N) and tuple(time_offset.shape) == (B,)
    s = stream if stream is not None else torch.cuda.current_stream(ref_pos.device)
    rf = Reference(B, N, K, kino_path.data_ptr(), per, kino_size.data_ptr() if kino_size is not None else None,
                   time_offset.data_ptr(), mpc_output.data_ptr(), Ts, REFERENCE_PI, ref_pos.data_ptr(), ref_yaw.data_ptr(),
                   replan.data_ptr() if replan is not None else None)
    _check(lib().frp_nmpc_reference_batch(ctypes.byref(rf), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_reference_batch")


def reference_batch_host(kino_path, time_offset, mpc_output, kino_size=None, Ts=0.05, device="cuda:0"):
    """Host convenience: numpy in, (ref_pos [B,N,3], ref_yaw [B,N], replan [B]) out."""
    import torch
    lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    B, rows, _ = mpc_output.shape
    N = rows - 1
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device=device); ry = torch.zeros((B, N), dtype=torch.float64, device=device)
    fl = torch.zeros((B,), dtype=torch.int32, device=device)
    ks = None if kino_size is None else torch.from_numpy(np.ascontiguousarray(kino_size, dtype=np.int32)).to(device)
    reference_batch_device(dev(kino_path), dev(time_offset), dev(mpc_output), rp, ry, fl, ks, Ts)
    torch.cuda.synchronize(device)
    return rp.cpu().numpy(), ry.cpu().numpy(), fl.cpu().numpy()


class AstarPlanner:
    """SURVEY 8f row f-4 (second half): the kinodynamic A* of NMPCSolver::getKinoPath for B planners on the device
    (frp_nmpc_astar_batch).  `world`: occupancy grid occ[x][y][z] (uint8) + the map / search constants of the reference's launch
    files (forces_resilient_planner_amd.workloads.astar_world).  Outputs stay in HBM: kino_path [B,K,3] / kino_size [B] are what
    DeviceFleet.references() takes as a per-planner path."""

    def __init__(self, world, B, K=1024, Ts=0.05, device="cuda:0", want_path_nodes=False, allocate_num=None):
        import torch
        lib()
        self.torch = torch
        self.B, self.K, self.Ts = B, K, Ts
        self.device = torch.device(device)
        self.world = world
        f64 = dict(dtype=torch.float64, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.occ = torch.from_numpy(np.ascontiguousarray(world["occ"], dtype=np.uint8)).to(self.device)
        self.kino_path = torch.zeros((B, K, 3), **f64)
        self.kino_size = torch.zeros((B,), **i32)
        self.status = torch.zeros((B,), **i32)
        self.stats = torch.zeros((B, 4), **i32)
        self.path_nodes = torch.zeros((B, ASTAR_MAX_PATH, 11), **f64) if want_path_nodes else None
        self.allocate_num = int(allocate_num or world["allocate_num"])
        self._q = [torch.zeros((B, 3), **f64) for _ in range(6)]
        self._retry = None  # (retry_pt, retry_vel): the start of the repeated search, when it differs (upload(..., retry=...))
        a = self._args()
        self.ws_bytes = int(lib().frp_nmpc_astar_workspace_bytes(ctypes.byref(a)))
        self.ws = torch.empty((self.ws_bytes // 8 + 1,), **f64)

    def _args(self, init=True, local_box=None, active=None):
        w = self.world
        a = Astar()
        a.B = self.B; a.occ = self.occ.data_ptr(); a.grid[:] = tuple(self.occ.shape); a.origin[:] = w["origin"]; a.map_size[:] = w["map_size"]
        a.resolution = w["resolution"]; a.local_box = local_box.data_ptr() if local_box is not None else None
        a.ego_r = w["ego_r"]; a.ego_h = w["ego_h"]
        for k in ("max_tau", "init_max_tau", "max_vel", "max_acc", "w_time", "horizon", "lambda_heu", "tie_breaker", "check_num"):
            setattr(a, k, w[k])
        a.allocate_num = self.allocate_num
        a.start_pt, a.start_vel, a.start_acc, a.end_pt, a.end_vel, a.external_acc = (t.data_ptr() for t in self._q)
        a.active = active.data_ptr() if active is not None else None
        a.init_search = 1 if init else 0
        a.Ts = self.Ts; a.K = self.K
        a.kino_path = self.kino_path.data_ptr(); a.kino_size = self.kino_size.data_ptr(); a.status = self.status.data_ptr()
        a.stats = self.stats.data_ptr(); a.path_nodes = self.path_nodes.data_ptr() if self.path_nodes is not None else None
        a.retry_pt = self._retry[0].data_ptr() if self._retry is not None else None
        a.retry_vel = self._retry[1].data_ptr() if self._retry is not None else None
        return a

    def upload(self, start_pt, start_v, start_a, end_pt, end_v, f_ext, retry=None):
        """retry = (pt [B,3], vel [B,3]): the start of the repeated search (getKinoPath retries from the odometry state,
        nmpc_solver.cpp:190-193); None: the same start."""
        t = self.torch
        cp = lambda dst, src: dst.copy_(src if t.is_tensor(src) else t.from_numpy(np.ascontiguousarray(src, dtype=np.float64)))
        for dst, src in zip(self._q, (start_pt, start_v, start_a, end_pt, end_v, f_ext)):
            cp(dst, src)
        if retry is None:
            self._retry = None
        else:
            if self._retry is None:
                self._retry = [t.zeros_like(self._q[0]), t.zeros_like(self._q[0])]
            cp(self._retry[0], retry[0]); cp(self._retry[1], retry[1])

    def plan(self, init=True, local_box=None, stream=None, active=None):
        """Asynchronous on `stream` (or torch's current stream): every planner's search + retry + getKinoTraj.
        active: int32 [B] device tensor -- only planners with a non-zero entry search (the others keep their path);
        a planner whose search ends in NO_PATH keeps its previous path as well."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        a = self._args(init, local_box, active)
        _check(lib().frp_nmpc_astar_batch(ctypes.byref(a), ctypes.c_void_p(self.ws.data_ptr()), self.ws_bytes, ctypes.c_void_p(s.cuda_stream)),
               "frp_nmpc_astar_batch")


class CloudGrid:
    """Uniform grid over a shared obstacle cloud (frp_nmpc_cloud_grid_build): built once per cloud, handed to
    corridor_batch_device / DeviceFleet.corridor so that a decomposition reads only the cells its local box touches."""

    def __init__(self, cloud, cell=0.5, origin=None, dims=None, stream=None):
        import torch
        assert cloud.dim() == 2 and cloud.shape[1] == 3 and cloud.is_contiguous() and cloud.dtype == torch.float64
        P = cloud.shape[0]
        if origin is None or dims is None:  # bounds of the cloud itself (host round trip; a map normally knows its bounds)
            finite = cloud[torch.isfinite(cloud).all(dim=1)]
            lo = finite.min(dim=0).values.cpu().numpy() if finite.numel() else np.zeros(3)
            hi = finite.max(dim=0).values.cpu().numpy() if finite.numel() else np.ones(3)
            origin = lo - 1e-9
            dims = np.maximum(1, np.ceil((hi - origin) / cell + 1e-9)).astype(int)
        self.origin = tuple(float(v) for v in origin); self.dims = tuple(int(v) for v in dims); self.cell = float(cell)
        cells = self.dims[0] * self.dims[1] * self.dims[2]
        dev = cloud.device
        self.points = torch.empty((max(P, 1), 3), dtype=torch.float64, device=dev)
        self.index = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
        self.start = torch.empty((cells + 1,), dtype=torch.int32, device=dev)
        scratch = torch.empty((cells,), dtype=torch.int32, device=dev)
        s = stream if stream is not None else torch.cuda.current_stream(dev)
        _check(lib().frp_nmpc_cloud_grid_build(ctypes.c_void_p(cloud.data_ptr()) if P else None, P, (ctypes.c_double * 3)(*self.origin),
                                               self.cell, (ctypes.c_int * 3)(*self.dims), ctypes.c_void_p(self.points.data_ptr()),
                                               ctypes.c_void_p(self.index.data_ptr()), ctypes.c_void_p(self.start.data_ptr()),
                                               ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_cloud_grid_build")
        s.synchronize()  # scratch may be freed


def corridor_batch_device(cloud, ref_pos, ref_yaw, ellipsoid, poly_A, poly_b, poly_nfaces, poly_index, poly_count=None,
                          cloud_count=None, consts=None, stream=None, grid=None):
    """frp_nmpc_corridor_batch on device tensors.  cloud [P,3] (shared) or [B,P,3]; ref_pos [B,N,3]; ref_yaw [B,N];
    ellipsoid [B,N,3,3]; outputs poly_A [B,N,F,3], poly_b [B,N,F], poly_nfaces / poly_index [B,N] int32."""
    import torch
    c = dict(CORRIDOR_DEFAULTS)
    c.update(consts or {})
    B, N, F, _ = poly_A.shape
    per = 1 if cloud.dim() == 3 else 0
    P = cloud.shape[-2]
    for t in (cloud, ref_pos, ref_yaw, ellipsoid, poly_A, poly_b):
        assert t.is_contiguous() and t.dtype == torch.float64
    assert tuple(ref_pos.shape) == (B, N, 3) and tuple(ellipsoid.shape) == (B, N, 3, 3) and tuple(poly_b.shape) == (B, N, F)
    assert poly_nfaces.dtype == torch.int32 and poly_index.dtype == torch.int32
    s = stream if stream is not None else torch.cuda.current_stream(ref_pos.device)
    cr = Corridor(B, N, F, P, cloud.data_ptr() if P else None, per, cloud_count.data_ptr() if cloud_count is not None else None,
                  ref_pos.data_ptr(), ref_yaw.data_ptr(), ellipsoid.data_ptr(), (ctypes.c_double * 3)(*c["bbox"]),
                  c["seed_len"], c["inflation"], c["offset_x"], poly_A.data_ptr(), poly_b.data_ptr(),
                  poly_nfaces.data_ptr(), poly_index.data_ptr(), poly_count.data_ptr() if poly_count is not None else None)
    if grid is not None:
        assert per == 0, "the grid belongs to a shared cloud"
        cr.grid_origin = (ctypes.c_double * 3)(*grid.origin); cr.grid_cell = grid.cell; cr.grid_dims = (ctypes.c_int * 3)(*grid.dims)
        cr.grid_points = grid.points.data_ptr(); cr.grid_index = grid.index.data_ptr(); cr.grid_start = grid.start.data_ptr()
    _check(lib().frp_nmpc_corridor_batch(ctypes.byref(cr), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_corridor_batch")


def corridor_batch_host(cloud, ref_pos, ref_yaw, ellipsoid, F=CORRIDOR_MAX_F, consts=None, device="cuda:0", grid_cell=None):
    """Host convenience: numpy in, (poly_index [B,N], poly_A [B,N,F,3], poly_b [B,N,F], poly_nfaces [B,N], poly_count [B]) out."""
    import torch
    lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    B, N, _ = ref_pos.shape
    A = torch.zeros((B, N, F, 3), dtype=torch.float64, device=device); b = torch.zeros((B, N, F), dtype=torch.float64, device=device)
    nf = torch.zeros((B, N), dtype=torch.int32, device=device); pi = torch.zeros((B, N), dtype=torch.int32, device=device)
    cnt = torch.zeros((B,), dtype=torch.int32, device=device)
    d_cloud = dev(cloud).reshape(-1, 3)
    grid = CloudGrid(d_cloud, grid_cell) if grid_cell else None
    corridor_batch_device(d_cloud, dev(ref_pos), dev(ref_yaw), dev(ellipsoid), A, b, nf, pi, cnt, consts=consts, grid=grid)
    torch.cuda.synchronize(device)
    return pi.cpu().numpy(), A.cpu().numpy(), b.cpu().numpy(), nf.cpu().numpy(), cnt.cpu().numpy()


def tube_batch_host(plans, consts=None, device="cuda:0"):
    """Host convenience: plans [B,N,17] (rows 0..N-1 of the plan deque) -> E [B,N,3,3]."""
    import torch
    lib()
    plans = np.ascontiguousarray(plans, dtype=np.float64)
    B, N, _ = plans.shape
    mo = torch.zeros((B, N + 1, L.NZ), dtype=torch.float64, device=device)
    mo[:, :N] = torch.from_numpy(plans).to(device)
    E = torch.empty((B, N, 3, 3), dtype=torch.float64, device=device)
    tube_batch_device(mo, E, consts)
    torch.cuda.synchronize(device)
    return E.cpu().numpy()


class DeviceFleet:
    """B planners whose receding-horizon loop lives in HBM: per tick
        pack (forces_normal.cpp:55-136 on the device)  ->  solve  ->  update (forces_normal.cpp:142-168,
        nmpc_solver.cpp:524-543 on the device).
    Host data is uploaded once (plans, polytopes, tube matrices); references / external forces per tick are device
    tensors handed to tick()."""

    def __init__(self, B, N, M, F, model, weights, device="cuda:0", npoly=None, weights_final=None):
        """weights_final: setParasFinal's five weights.  When given, every planner carries its own mode
        (self.mode [B], all FRP_MODEL_NORMAL at first): pack() picks its weights and solve() its objective by it, and
        update_mode() applies the reference's switch rule (nmpc_solver.cpp:436-447)."""
        import torch
        self.torch = torch
        self.B, self.N, self.M, self.F, self.model = B, N, M, F, model
        self.NPOLY = N if npoly is None else npoly
        self.weights = tuple(float(x) for x in weights)  # (w_stage_wp, w_stage_input, w_input_rate, w_terminal_wp, w_terminal_input)
        self.solver = DeviceSolver(B, N, M, min(M, F), model, device)
        # a fleet ticks: this tick's problem is the previous one shifted by a stage, and its iteration count is the best predictor of
        # this one's (frp_nmpc_batch.order_hint; zero before the first solve = "typical").  Saves the key kernel's pass over the
        # parameters as well (27 of a tick's 1590 us at 4096 planners).  FRP_FLEET_ORDER_HINT=0: the key of the initial guess, as before
        self.solver.order_by_last_iters = os.environ.get("FRP_FLEET_ORDER_HINT", "1") != "0"
        dev = self.solver.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.mpc_output = torch.zeros((B, N + 1, L.NZ), **f64)
        self.ellipsoid = torch.zeros((B, N, 3, 3), **f64)
        self.poly_A = torch.zeros((B, self.NPOLY, F, 3), **f64)
        self.poly_b = torch.zeros((B, self.NPOLY, F), **f64)
        self.poly_nfaces = torch.zeros((B, self.NPOLY), dtype=torch.int32, device=dev)
        self.poly_index = None
        # per planner: polytopes produced by the last corridor() (>= 1), negated when one of them needed more than F rows and
        # was truncated (getSikangConst tests all rows; callers that size F below FRP_CORRIDOR_MAX_F should check overflowed())
        self.poly_count = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.weights_final = None if weights_final is None else tuple(float(x) for x in weights_final)
        self.mode = None
        if self.weights_final is not None:
            self.mode = torch.full((B,), int(model), dtype=torch.int32, device=dev)
            self.solver.models = self.mode

    def update_mode(self, time_offset, kino_size, end_pt, Ts=0.05, radius=1.0, stream=None):
        """switch_to_final for every planner (nmpc_solver.cpp:436-447): time_offset [B] f64, kino_size int32 [1] or [B],
        end_pt f64 [3] or [B,3] (device tensors).  Sticky until reset_mode()."""
        assert self.mode is not None, "construct the fleet with weights_final"
        s = stream if stream is not None else self.torch.cuda.current_stream(self.solver.device)
        _check(lib().frp_nmpc_mode_batch(self.B, self.N, ctypes.c_void_p(self.mpc_output.data_ptr()), ctypes.c_void_p(time_offset.data_ptr()),
                                         ctypes.c_void_p(kino_size.data_ptr()), 1 if kino_size.numel() > 1 else 0,
                                         ctypes.c_void_p(end_pt.data_ptr()), 1 if end_pt.dim() == 2 else 0, float(Ts), float(radius),
                                         ctypes.c_void_p(self.mode.data_ptr()), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_mode_batch")

    def reset_mode(self):
        """A new kinodynamic path puts every planner back on the normal solver (nmpc_solver.cpp:218)."""
        self.mode.fill_(L.MODEL_NORMAL)

    def to_device(self, a, dtype=None):
        t = self.torch
        return t.from_numpy(np.ascontiguousarray(a)).to(self.solver.device, dtype=dtype)

    def pack(self, external_acc, ref_pos, ref_yaw, stream=None):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.solver.device)
        ds = self.solver
        pk = Pack(self.B, self.N, self.M, self.NPOLY, self.F, 1 if external_acc.dim() == 3 else 0,
                  self.mpc_output.data_ptr(), external_acc.data_ptr(),
                  ref_pos.data_ptr(), ref_yaw.data_ptr(), self.ellipsoid.data_ptr(), self.poly_A.data_ptr(),
                  self.poly_b.data_ptr(), self.poly_nfaces.data_ptr(),
                  self.poly_index.data_ptr() if self.poly_index is not None else None, *self.weights,
                  ds.xinit.data_ptr(), ds.x0.data_ptr(), ds.params.data_ptr(), ds.nfaces.data_ptr(),
                  self.mode.data_ptr() if self.mode is not None else None, *(self.weights_final or (0.0,) * 5),
                  # (from the second pack on: this fleet's solver buffers are as its own previous pack left them -- DeviceSolver.upload takes the promise back)
                  1 if getattr(ds, "_packed_by", None) is self else 0)
        _check(lib().frp_nmpc_pack_batch(ctypes.byref(pk), ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_pack_batch")
        ds._packed_by = self

    def update(self, stream=None, keep_failed=True):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.solver.device)
        ds = self.solver
        _check(lib().frp_nmpc_update_batch(self.B, self.N, ctypes.c_void_p(ds.z.data_ptr()),
                                           ctypes.c_void_p(ds.exitflag.data_ptr()) if keep_failed else None,
                                           ctypes.c_void_p(self.mpc_output.data_ptr()), ctypes.c_void_p(s.cuda_stream)),
               "frp_nmpc_update_batch")

    def tube(self, consts=None, stream=None):
        """ellipsoid_matrices_ of all B planners from their current plans
        (NMPCSolver::setFORCESParams, nmpc_solver.cpp:484-521) -> self.ellipsoid, on the device."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.solver.device)
        tube_batch_device(self.mpc_output, self.ellipsoid, consts, s)

    def corridor(self, cloud, ref_pos, ref_yaw, consts=None, stream=None, cloud_count=None, grid=None):
        """polytopes and poly_indices of all B planners from the obstacle cloud, the stage
        references and the current tube (getSikangConst, nmpc_solver.cpp:288-332) -> self.poly_*, on the device."""
        t = self.torch
        assert self.NPOLY == self.N
        if self.poly_index is None:
            self.poly_index = t.zeros((self.B, self.N), dtype=t.int32, device=self.solver.device)
        corridor_batch_device(cloud, ref_pos, ref_yaw, self.ellipsoid, self.poly_A, self.poly_b, self.poly_nfaces,
                              self.poly_index, self.poly_count, cloud_count, consts, stream, grid)

    def overflowed(self):
        """Planners whose last corridor() truncated a polytope to F rows (device tensor of bool; all False before the first
        corridor() call)."""
        return self.poly_count < 0

    def coldstart(self, state=None, only_failed=True, thrust=7.3, stream=None):
        """initMPCOutput for the planners whose last solve failed (nmpc_solver.cpp:363-364, :265-286), on the device.
        state [B,9] = stateMpc_ (odometry), or None: each planner restarts from its plan's stage-1 state."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.solver.device)
        _check(lib().frp_nmpc_coldstart_batch(self.B, self.N, ctypes.c_void_p(state.data_ptr()) if state is not None else None,
                                              ctypes.c_void_p(self.solver.exitflag.data_ptr()) if only_failed else None,
                                              float(thrust), ctypes.c_void_p(self.mpc_output.data_ptr()),
                                              ctypes.c_void_p(s.cuda_stream)), "frp_nmpc_coldstart_batch")

    def references(self, kino_path, time_offset, ref_pos, ref_yaw, replan=None, kino_size=None, Ts=0.05, stream=None):
        """SURVEY 8f row f-4 (first half): ref_total_pos_ / ref_total_yaw_ of all B planners from the kinodynamic
        path (getCurTraj + calculate_yaw, nmpc_solver.cpp:109-142, 834-862), on the device."""
        reference_batch_device(kino_path, time_offset, self.mpc_output, ref_pos, ref_yaw, replan, kino_size, Ts, stream)

    def replan(self, planner, end_pt, external_acc, replan, time_offset=None, end_vel=None, init=True, mass=0.74, g=9.81, stream=None,
               t_cur=None, odom=None, Ts=0.05):
        """The FSM's REPLAN_TRAJ step (nmpc_manage.cpp:215-235 -> NMPCSolver::getKinoPath, nmpc_solver.cpp:145-223) for the planners
        whose tick raised kino_replan_ (`replan` [B] int32, as written by references() / full_tick): a kinodynamic A* to end_pt
        [B,3] with external_acc [B,3] in the primitives, on the device (AstarPlanner = frp_nmpc_astar_batch).
        Start state as in the reference (:159-186): a planner whose last solve succeeded (solver.exitflag == 1) starts from its plan
        interpolated at t_cur [B] seconds after the plan's start -- row floor(t_cur / Ts) towards the next one; None = the plan's
        stage-1 row (mpc_output[:, 1], the state the tick is about to apply) -- with the acceleration the planned thrust produces (:169-181), provided floor(t_cur / Ts) < N - 1 and t_cur >= 0;
        every other planner starts from odom = (pos [B,3], vel [B,3]) with zero acceleration.  The repeated search after NO_PATH starts
        from the odometry state too (:190-193).  odom = None is NOT the reference's behaviour: fallback and retry then start from the
        plan's stage-1 state (a warning is issued once); a caller that has odometry passes it.
        The planner object owns the per-planner paths (planner.kino_path / kino_size): pass them to references() / full_tick as the
        path.  Planners that found a path get time_offset = 0 (kino_start_time_ = now, :219) and go back to the normal solver (:218).
        Returns the mask (bool [B]) of planners that received a new path.  Everything here -- the state gather, the search, the
        masks -- is enqueued on `stream` (torch's current stream when None)."""
        t = self.torch
        s = stream if stream is not None else t.cuda.current_stream(self.solver.device)
        with t.cuda.stream(s):
            B, N = self.B, self.N
            rows = self.mpc_output[:, :N]                      # pre_mpc_output_: rows 0..N-1 of the deque
            tc = t_cur if t_cur is not None else t.zeros((B,), dtype=t.float64, device=rows.device)
            idx = t.floor(tc / Ts).to(t.int64)
            use_plan = (self.solver.exitflag == 1) & (idx < N - 1) & (tc >= 0.0)
            i0 = idx.clamp(0, N - 2)
            ar = t.arange(B, device=rows.device)
            r0, r1 = rows[ar, i0], rows[ar, i0 + 1]
            mo = r0 + (t.fmod(tc, Ts) / Ts)[:, None] * (r1 - r0)
            if t_cur is None:
                mo = self.mpc_output[:, 1].clone()             # (the tick's convention here: the plan's next state)
            e = mo[:, 14:17]
            sr, cr, sp, cp, sy, cy = t.sin(e[:, 0]), t.cos(e[:, 0]), t.sin(e[:, 1]), t.cos(e[:, 1]), t.sin(e[:, 2]), t.cos(e[:, 2])
            zb = t.stack([cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr], 1)  # eulerToRot(e) [0 0 1]'
            acc = zb * (mo[:, 3:4] / mass)
            acc[:, 2] -= g
            if odom is None and not getattr(self, "_warned_no_odom", False):
                import warnings
                warnings.warn("DeviceFleet.replan without odom: fallback and retry start from the plan's stage-1 state, not from the "
                              "odometry state as in the reference (nmpc_solver.cpp:151-153, 190-193)")
                self._warned_no_odom = True
            o_pt, o_v = (odom if odom is not None else (self.mpc_output[:, 1, 8:11], self.mpc_output[:, 1, 11:14]))
            up = use_plan[:, None]
            s_pt = t.where(up, mo[:, 8:11], o_pt); s_v = t.where(up, mo[:, 11:14], o_v); s_a = t.where(up, acc, t.zeros_like(acc))
            ev = end_vel if end_vel is not None else t.zeros_like(end_pt)
            planner.upload(s_pt.contiguous(), s_v.contiguous(), s_a.contiguous(), end_pt, ev, external_acc,
                           retry=(o_pt.contiguous(), o_v.contiguous()))
            planner.plan(init=init, active=replan, stream=s)
            ok = (replan != 0) & (planner.status != ASTAR_NO_PATH)
            if time_offset is not None:
                time_offset.masked_fill_(ok, 0.0)
            if self.mode is not None:
                self.mode.masked_fill_(ok, L.MODEL_NORMAL)
        return ok

    def full_tick(self, external_acc, kino_path, time_offset, cloud, ref_pos, ref_yaw, stream=None, replan=None,
                  kino_size=None, tube_consts=None, corridor_consts=None, Ts=0.05, coldstart=True, state=None, grid=None):
        """The reference's whole per-tick computation downstream of the A* (NMPCSolver::solveNMPC,
        nmpc_solver.cpp:351-482) for B planners, asynchronous on `stream`, nothing touching the host:
        stage references (f-4) -> tube (f-2) -> corridor (f-3) -> parameter packing (f-1) -> NLP solve -> result
        bookkeeping.  ref_pos [B,N,3] / ref_yaw [B,N] are caller-owned scratch that receives the references.
        With coldstart (default) planners whose previous solve failed first restart from the constant plan (:363-364)."""
        if coldstart:
            self.coldstart(state, True, stream=stream)
        self.references(kino_path, time_offset, ref_pos, ref_yaw, replan, kino_size, Ts, stream)
        self.tube(tube_consts, stream)
        self.corridor(cloud, ref_pos, ref_yaw, corridor_consts, stream, grid=grid)
        self.pack(external_acc, ref_pos, ref_yaw, stream)
        self.solver.solve(stream)
        self.update(stream)

    def tick(self, external_acc, ref_pos, ref_yaw, stream=None, tube_consts=None, propagate_tube=False):
        """One receding-horizon tick of all B planners, asynchronous on `stream`.  With propagate_tube the
        tube matrices are recomputed from the current plans first, as the reference does every tick."""
        if propagate_tube:
            self.tube(tube_consts, stream)
        self.pack(external_acc, ref_pos, ref_yaw, stream)
        self.solver.solve(stream)
        self.update(stream)
