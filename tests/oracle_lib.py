"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
D = ctypes.POINTER(ctypes.c_double)
IP = ctypes.POINTER(ctypes.c_int)


class OrcInfo(ctypes.Structure):
    _fields_ = [("it", ctypes.c_int)] + [(n, ctypes.c_double) for n in
                                         "res_eq res_ineq rsnorm rcompnorm pobj mu mu_aff sigma step_aff step_cc".split()] + [("nfallback", ctypes.c_int)]


class OrcOptions(ctypes.Structure):
    _fields_ = [("maxit", ctypes.c_int)] + [(n, ctypes.c_double) for n in
                                            "tol_stat tol_eq tol_ineq tol_comp mu0 ftb".split()] + [("hessian", ctypes.c_int), ("diverge_mu", ctypes.c_double), ("twist", ctypes.c_int)]


def P(a):
    return a.ctypes.data_as(D) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORC_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORC_DIR, "liboracle.so"])
        _lib = ctypes.CDLL(so)
        _lib.orc_solve.restype = ctypes.c_int
    return _lib


def default_options(**kw):
    o = OrcOptions()
    lib().orc_default_options(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def stage_eval(z, p, M, stage_class, model):
    f = np.zeros(1); gf = np.zeros(17); c = np.zeros(13); Jc = np.zeros(13 * 17); h = np.zeros(M); Jh = np.zeros(M * 17)
    z = np.ascontiguousarray(z, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    lib().orc_stage_eval(P(z), P(p), M, stage_class, model, P(f), P(gf), P(c), P(Jc), P(h), P(Jh))
    return dict(f=f[0], gf=gf, c=c, Jc=Jc, h=h, Jh=Jh)


def solve_batch(w, opt=None, nthreads=0, x0=None):
    """w: workload dict (forces_resilient_planner_amd.workloads).  Returns z [B,N,17], flags, infos."""
    B, N, M = w["xinit"].shape[0], w["N"], w["M"]
    xinit = np.ascontiguousarray(w["xinit"]); z0 = np.ascontiguousarray(w["x0"] if x0 is None else x0)
    params = np.ascontiguousarray(w["params"]); nf = np.ascontiguousarray(w["nfaces"], dtype=np.int32)
    z = np.zeros((B, N, 17)); fl = np.zeros(B, dtype=np.int32); info = (OrcInfo * B)()
    lib().orc_solve_batch(B, N, M, int(w["model"]), P(xinit), P(z0), P(params), nf.ctypes.data_as(IP),
                          ctypes.byref(opt) if opt is not None else None, P(z), fl.ctypes.data_as(IP), info, nthreads)
    return z, fl, info


def solve_one(xinit, x0, params, nfaces, N, M, model, opt=None):
    xinit = np.ascontiguousarray(xinit); x0 = np.ascontiguousarray(x0); params = np.ascontiguousarray(params)
    z = np.zeros((N, 17)); info = OrcInfo()
    nf = None if nfaces is None else np.ascontiguousarray(nfaces, dtype=np.int32)
    fl = lib().orc_solve(N, M, model, P(xinit), P(x0), P(params), nf.ctypes.data_as(IP) if nf is not None else None,
                         ctypes.byref(opt) if opt is not None else None, P(z), ctypes.byref(info))
    return z, fl, info


def ref_model_available():
    return os.path.exists(os.path.join(ORC_DIR, "_ref", "libref_model_normal.so"))


def reference_kkt(z, xinit, params, nfaces, N, M, model, active=1e-3):
    """KKT residuals of a returned plan z [N,17] for the REFERENCE NLP, measured with the reference's own
    CasADi callbacks (oracle/_ref, FORCESNLPsolver_*_casadi2forces.c:42-245) -- no oracle arithmetic.

    The reference's output struct carries no multipliers (FORCESNLPsolver_normal.h:129-190), so they are
    recovered here: least squares over (y free, lambda >= 0 on the constraints within `active` of their
    bound) of |grad f + A'y - C'lambda|.  Returns dict(stat, eq, ineq, bound) of infinity norms.
    """
    import sys
    from scipy.optimize import lsq_linear
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import gen_golden as G
    nlp = G.RefNLP(N, M, model, np.asarray(xinit, float), np.asarray(params, float), np.asarray(nfaces))
    Z = np.ascontiguousarray(z, dtype=np.float64).ravel()
    _, g = nlp.fun(Z)
    A = nlp.eq_jac(Z)
    cols = [A.T]
    lo = [-np.inf] * A.shape[0]
    c = nlp.ineq(Z)
    if c.size:
        act = np.nonzero(c < active)[0]
        cols.append(-nlp.ineq_jac(Z)[act].T)
        lo += [0.0] * len(act)
    lb, ub = np.tile(nlp.lb, N), np.tile(nlp.ub, N)
    n = Z.size
    for i in np.nonzero(Z - lb < active)[0]:
        e = np.zeros((n, 1)); e[i] = -1.0; cols.append(e); lo.append(0.0)
    for i in np.nonzero(ub - Z < active)[0]:
        e = np.zeros((n, 1)); e[i] = 1.0; cols.append(e); lo.append(0.0)
    Bm = np.concatenate(cols, axis=1)
    res = lsq_linear(Bm, -g, bounds=(np.array(lo), np.full(len(lo), np.inf)), tol=1e-14, max_iter=200)
    r = g + Bm @ res.x
    return dict(stat=float(np.max(np.abs(r))), eq=float(np.max(np.abs(nlp.eq(Z)))),
                ineq=float(max(0.0, -c.min())) if c.size else 0.0,
                bound=float(max(0.0, (lb - Z).max(), (Z - ub).max())))
