#!/usr/bin/env python3
"""A second, independent local method on the instances of tests/golden/solutions_hard.npz that SciPy SLSQP did not settle
(VERDICT r05 item 5a): scipy's trust-constr (a trust-region barrier / SQP method -- no line search, another globalisation than
SLSQP's) on the REFERENCE NLP, evaluated through the reference's own callbacks only.

Round 5 could not finish a single instance: gen_golden.RefNLP crosses ctypes N x 3 times per NLP evaluation and hands scipy dense
256 x 340 Jacobians.  Here an evaluation is ONE C call (tests/tools/ref_nlp_shim.c -> oracle/_ref/libref_nlp_shim.so, which calls the
reference callback for all N stages) and the Jacobians are scipy.sparse with the stage structure (block bidiagonal).

  * the 4 instances the interior-point iteration exits -7 on (2, 30, 62, 134): from the planner's cold start, from the iteration's last
    iterate and from a perturbation of it.  A feasible optimum on one of them would reopen the line-search question
    (FORCESNLPsolver_normal.h:89-95); none = two local methods agree that there is no feasible point to be found from there.
  * the instances SLSQP stalled on although the solver converges (status != 0 in the fixture, mostly of the `far` kind): started 0.02
    from the solver's own 1e-8 solution.  Does the second method end at the solver's point?

Run in the BUILD container (needs oracle/_ref):  python tests/tools/hard_trust_constr.py [workers] [exits|stalled|all]
Appends one JSON line per run to profiles/r06_hard_trust_constr.jsonl (resumable: finished (instance, start) pairs are skipped)."""
import ctypes
import json
import os
import subprocess
import sys
import time
from multiprocessing import Pool

import numpy as np
import scipy.sparse as sp
from scipy.optimize import Bounds, NonlinearConstraint, minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
import tests.oracle_lib as OL  # noqa: E402
from forces_resilient_planner_amd import layout as L  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "solutions_hard.npz")
OUT = os.path.join(ROOT, "profiles", "r06_hard_trust_constr.jsonl")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_nlp_shim.so")
D = ctypes.POINTER(ctypes.c_double)


def build_shim():
    src = os.path.join(ROOT, "tests", "tools", "ref_nlp_shim.c")
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", src, "-o", SHIM])
    return SHIM


class RefNLP2:
    """The reference NLP (matlab_code/setup.m:36-66) from the reference callbacks: one C call per evaluation, sparse Jacobians."""

    def __init__(self, N, M, model, xinit, params, nfaces):
        self.N, self.M, self.model = N, M, model
        self.xinit = np.asarray(xinit, float)
        self.nf = np.asarray(nfaces).astype(int)
        self.p130 = np.ascontiguousarray(np.stack([G.to_p130(np.asarray(params[k], float), M) for k in range(N)]))
        lb, ub = L.bounds()
        self.lb, self.ub = np.tile(lb, N), np.tile(ub, N)
        self.shim = ctypes.CDLL(build_shim()).ref_nlp_eval
        self.shim.restype = None
        self.shim.argtypes = [ctypes.c_void_p, ctypes.c_int] + [D] * 8
        self.cb = ctypes.cast(G.ref_callback(model), ctypes.c_void_p)
        self.f = np.zeros(N); self.gf = np.zeros((N, 17)); self.c = np.zeros((N, 13)); self.Jc = np.zeros((N, 221))
        self.h = np.zeros((N, 30)); self.Jh = np.zeros((N, 510))
        self.last = None
        self.nev = 0
        # sparsity of the equality Jacobian: rows [x0 = xinit (9)] + 13 per transition; block (k, k) dense 13 x 17, block (k, k+1) = -E
        n = 17 * N
        E = np.zeros((13, 17)); E[0:9, 8:17] = np.eye(9); E[9:13, 4:8] = np.eye(4)
        self.Eneg = sp.csr_matrix(-E)
        self.J0 = sp.csr_matrix((np.ones(9), (np.arange(9), 8 + np.arange(9))), shape=(9, n))

    def _eval(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if self.last is not None and np.array_equal(Z, self.last):
            return
        P = lambda a: a.ctypes.data_as(D)
        self.shim(self.cb, self.N, P(Z), P(self.p130), P(self.f), P(self.gf), P(self.c), P(self.Jc), P(self.h), P(self.Jh))
        self.last = Z.copy()
        self.nev += 1

    def fun(self, Z):
        self._eval(Z)
        return float(self.f.sum()), self.gf.ravel().copy()

    def eq(self, Z):
        self._eval(Z)
        z = np.asarray(Z).reshape(self.N, 17)
        r = [z[0, 8:17] - self.xinit]
        nxt = np.concatenate([z[1:, 8:17], z[1:, 4:8]], axis=1)
        r.append((self.c[:self.N - 1] - nxt).ravel())
        return np.concatenate(r)

    def eq_jac(self, Z):
        self._eval(Z)
        N = self.N
        blocks = [[None] * N for _ in range(N)]
        rows = [self.J0]
        for k in range(N - 1):
            row = [None] * N
            row[k] = sp.csr_matrix(self.Jc[k].reshape(17, 13).T)
            row[k + 1] = self.Eneg
            for j in range(N):
                if row[j] is None:
                    row[j] = sp.csr_matrix((13, 17))
            rows.append(sp.hstack(row, format="csr"))
        return sp.vstack(rows, format="csr")

    def ineq(self, Z):  # >= 0
        self._eval(Z)
        return np.concatenate([L.HU - self.h[k, :self.nf[k]] for k in range(self.N)]) if self.nf.sum() else np.zeros(0)

    def ineq_jac(self, Z):
        self._eval(Z)
        N = self.N
        rows = []
        for k in range(N):
            if self.nf[k] == 0:
                continue
            Jh = -self.Jh[k].reshape(17, 30).T[:self.nf[k]]
            row = [sp.csr_matrix((self.nf[k], 17)) for _ in range(N)]
            row[k] = sp.csr_matrix(Jh)
            rows.append(sp.hstack(row, format="csr"))
        return sp.vstack(rows, format="csr") if rows else sp.csr_matrix((0, 17 * N))


def check_against_refnlp(g, i):
    """RefNLP2 == gen_golden.RefNLP (the stage-by-stage form every fixture was made with) at a random point."""
    N, M = int(g["N"]), int(g["M"])
    a = G.RefNLP(N, M, int(g["model"][i]), g["xinit"][i], g["params"][i], g["nfaces"][i])
    b = RefNLP2(N, M, int(g["model"][i]), g["xinit"][i], g["params"][i], g["nfaces"][i])
    rng = np.random.default_rng(5)
    Z = np.clip(g["x0"][i].ravel() + 0.3 * rng.normal(size=17 * N), b.lb, b.ub)
    fa, ga = a.fun(Z); fb, gb = b.fun(Z)
    assert abs(fa - fb) <= 1e-12 * max(1.0, abs(fa)) and np.array_equal(ga, gb)
    assert np.array_equal(a.eq(Z), b.eq(Z)) and np.array_equal(a.eq_jac(Z), b.eq_jac(Z).toarray())
    assert np.array_equal(a.ineq(Z), b.ineq(Z)) and np.array_equal(a.ineq_jac(Z), b.ineq_jac(Z).toarray())


def run(args):
    i, name, z0, zo, flag_ipm, maxiter = args
    g = dict(np.load(PATH, allow_pickle=False))
    N, M = int(g["N"]), int(g["M"])
    nlp = RefNLP2(N, M, int(g["model"][i]), g["xinit"][i], g["params"][i], g["nfaces"][i])
    cons = [NonlinearConstraint(nlp.eq, 0.0, 0.0, jac=nlp.eq_jac)]
    if int(nlp.nf.sum()) > 0:
        cons.append(NonlinearConstraint(nlp.ineq, 0.0, np.inf, jac=nlp.ineq_jac))
    t = time.time()
    res = minimize(nlp.fun, np.clip(z0, nlp.lb, nlp.ub), jac=True, method="trust-constr", bounds=Bounds(nlp.lb, nlp.ub), constraints=cons,
                   options=dict(maxiter=maxiter, gtol=1e-8, xtol=1e-11, barrier_tol=1e-9, sparse_jacobian=True))
    x = res.x
    c = nlp.ineq(x)
    eq, ineq = float(np.max(np.abs(nlp.eq(x)))), float(max(0.0, -c.min())) if c.size else 0.0
    k = OL.reference_kkt(x.reshape(N, 17), g["xinit"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
    out = dict(instance=int(i), kind=["far", "force", "tight", "replan"][int(i) % 4], start=name, method="trust-constr", status=int(res.status), message=str(res.message)[:80],
               nit=int(res.nit), nlp_evaluations=int(nlp.nev), f=float(nlp.fun(x)[0]), eq=eq, ineq=ineq, kkt=k,
               feasible_optimum=bool(eq < 1e-7 and ineq < 1e-7 and k["stat"] < 1e-5), ipm_flag=int(flag_ipm),
               dist_to_ipm_point=(float(np.max(np.abs(x - zo.ravel()))) if flag_ipm == 1 else None),
               f_ipm=(float(nlp.fun(zo.ravel())[0]) if flag_ipm == 1 else None), secs=round(time.time() - t, 1))
    with open(OUT, "a") as f:
        f.write(json.dumps(out) + "\n")
    return out


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    maxiter = int(os.environ.get("FRP_TC_MAXITER", "3000"))
    build_shim()
    g = dict(np.load(PATH, allow_pickle=False))
    N, M = int(g["N"]), int(g["M"])
    check_against_refnlp(g, 2); check_against_refnlp(g, 5)
    print("RefNLP2 == gen_golden.RefNLP (bitwise) on two instances", flush=True)
    done = set()
    if os.path.exists(OUT):
        for line in open(OUT):
            r = json.loads(line); done.add((r["instance"], r["start"]))
    tight = OL.default_options(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)
    jobs = []
    for i in np.where(g["status"] != 0)[0]:
        zo, fl, info = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]), tight)
        rng = np.random.default_rng(11000 + int(i))
        if fl != 1 and which in ("exits", "all"):
            starts = {"cold": g["x0"][i].ravel().copy(), "ipm_last_iterate": zo.ravel().copy(),
                      "ipm_last_iterate + 0.05 noise": zo.ravel() + 0.05 * rng.normal(size=zo.size)}
        elif fl == 1 and which in ("stalled", "all"):
            starts = {"ipm_solution + 0.02 noise": zo.ravel() + 0.02 * rng.normal(size=zo.size), "cold": g["x0"][i].ravel().copy()}
        else:
            starts = {}
        for name, z0 in starts.items():
            if (int(i), name) not in done:
                jobs.append((int(i), name, z0, zo, int(fl), maxiter))
    # the stalled instances from near the solver's point first (minutes each), then the exits (their equality Jacobian goes singular along the way and
    # scipy falls back to dense SVD projections: an hour each at FRP_TC_MAXITER_EXITS iterations), the cold starts of the stalled instances last
    mx_exits = int(os.environ.get("FRP_TC_MAXITER_EXITS", "400"))
    jobs = [(j[0], j[1], j[2], j[3], j[4], (mx_exits if j[4] != 1 else j[5])) for j in jobs]
    jobs.sort(key=lambda j: (0 if (j[4] == 1 and j[1] != "cold") else (1 if j[4] != 1 else 2)))
    print(f"{len(jobs)} runs", flush=True)
    with Pool(workers) as pool:
        for r in pool.imap_unordered(run, jobs, chunksize=1):
            print(r["instance"], r["kind"], r["start"], "status", r["status"], "nit", r["nit"], "feasible_optimum", r["feasible_optimum"],
                  "eq %.1e ineq %.1e stat %.1e" % (r["eq"], r["ineq"], r["kkt"]["stat"]), "dist", r["dist_to_ipm_point"], r["secs"], "s", flush=True)
