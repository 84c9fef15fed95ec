"""Second pass over tests/golden/solutions_hard.npz: the instances SciPy SLSQP did not solve in
gen_golden.py's pass (status 8 at a feasible point: 41 of 192, 29 of them of the `far` kind) are tried again on the reference NLP
(reference callbacks only: gen_golden.RefNLP)
  * by SLSQP started near the oracle's 1e-8 solution (perturbed by 0.02, then 0.002, 250 iterations each: a local method confirming -- or not --
    that the point the interior-point iteration found is a local solution of the REFERENCE problem), and
  * for the instances the interior-point iteration itself gives up on (exit -7), by SLSQP and trust-constr from several starts (the
    planner's cold start, the iteration's last iterate, perturbations): is there a feasible optimum at all?
The fixture is rewritten in place with the new solutions (status 0, start = 'near-retry'); the study of the exits goes to
profiles/r05_hard_exits_study.json.     python tests/tools/extend_hard_golden.py [workers]"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
import tests.oracle_lib as OL  # noqa: E402

PATH = os.path.join(ROOT, "tests", "golden", "solutions_hard.npz")


def _nlp(g, i):
    return G.RefNLP(int(g["N"]), int(g["M"]), int(g["model"][i]), g["xinit"][i], g["params"][i], g["nfaces"][i])


def _retry(args):
    i, zo, g = args
    nlp = _nlp(g, i)
    t = time.time()
    for k, eps in enumerate((0.02, 0.002)):
        rng = np.random.default_rng(7000 + 10 * i + k)
        res = nlp.solve(zo.ravel() + eps * rng.normal(size=zo.size), maxiter=250)
        c = nlp.ineq(res.x)
        eq, ineq = float(np.max(np.abs(nlp.eq(res.x)))), float(max(0.0, -c.min())) if c.size else 0.0
        if res.status == 0 and eq < 1e-8 and ineq < 1e-8:
            return dict(i=i, ok=True, eps=eps, z=res.x.reshape(zo.shape), f=float(res.fun), nit=int(res.nit), eq=eq, ineq=ineq, secs=time.time() - t)
    return dict(i=i, ok=False, status=int(res.status), secs=time.time() - t)


def _exit_study(args):
    i, g, zlast = args
    from scipy.optimize import minimize, NonlinearConstraint, Bounds
    nlp = _nlp(g, i)
    N = int(g["N"])
    lb, ub = np.tile(nlp.lb, N), np.tile(nlp.ub, N)
    starts = {"cold": g["x0"][i].ravel().copy(), "ipm_last_iterate": zlast.ravel().copy()}
    rng = np.random.default_rng(9000 + i)
    light = os.environ.get("FRP_HARD_STUDY_LIGHT") == "1"  # (two starts, 150 trust-constr iterations: minutes instead of hours per instance)
    for k in range(0 if light else 2):
        starts[f"ipm_last_iterate + {0.05 * (k + 1):.2f} noise"] = zlast.ravel() + 0.05 * (k + 1) * rng.normal(size=zlast.size)
    out = []
    for name, z0 in starts.items():
        for method in ("SLSQP", "trust-constr"):
            t = time.time()
            if method == "SLSQP":
                res = nlp.solve(z0, maxiter=400)
                x, status, nit = res.x, int(res.status), int(res.nit)
            else:
                cons = [NonlinearConstraint(nlp.eq, 0.0, 0.0, jac=nlp.eq_jac)]
                if int(np.sum(nlp.nf)) > 0:
                    cons.append(NonlinearConstraint(nlp.ineq, 0.0, np.inf, jac=nlp.ineq_jac))
                res = minimize(nlp.fun, np.clip(z0, lb, ub), jac=True, method="trust-constr", bounds=Bounds(lb, ub), constraints=cons,
                               options=dict(maxiter=150 if light else 600, gtol=1e-8, xtol=1e-12))
                x, status, nit = res.x, int(res.status), int(res.nit)
            c = nlp.ineq(x)
            eq, ineq = float(np.max(np.abs(nlp.eq(x)))), float(max(0.0, -c.min())) if c.size else 0.0
            k = OL.reference_kkt(x.reshape(N, 17), g["xinit"][i], g["params"][i], g["nfaces"][i], N, int(g["M"]), int(g["model"][i]))
            out.append(dict(instance=int(i), start=name, method=method, status=status, nit=nit, f=float(nlp.fun(x)[0]), eq=eq, ineq=ineq,
                            kkt=k, feasible_optimum=bool(eq < 1e-8 and ineq < 1e-8 and k["stat"] < 1e-6), secs=time.time() - t))
    return out


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    g = dict(np.load(PATH, allow_pickle=False))
    N, M, n = int(g["N"]), int(g["M"]), g["z"].shape[0]
    tight = OL.default_options(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)
    jobs, exits = [], []
    for i in np.where(g["status"] != 0)[0]:
        zo, fl, info = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]), tight)
        if fl == 1:
            jobs.append((int(i), zo, g))
        else:
            exits.append((int(i), g, zo))
    print(f"{len(jobs)} instances to retry near the oracle's solution, {len(exits)} exits to study", flush=True)
    skip_study = os.environ.get("FRP_HARD_SKIP_EXIT_STUDY") == "1"  # (the trust-constr runs of the study take hours on a few cores)
    with Pool(workers) as pool:
        # the retries first: the study's tasks are long and would hold every worker but one
        res = []
        for r in pool.imap_unordered(_retry, jobs, chunksize=1):
            res.append(r); print('retry', r['i'], r['ok'], round(r['secs']), flush=True)
        study = [] if skip_study else pool.map(_exit_study, exits, chunksize=1)
    start = g["start"].astype("U12")
    for r in res:
        if r["ok"]:
            i = r["i"]
            g["z"][i], g["f"][i], g["status"][i], g["nit"][i], g["eq"][i], g["ineq"][i] = r["z"], r["f"], 0, r["nit"], r["eq"], r["ineq"]
            start[i] = "near-retry"
    g["start"] = start
    np.savez_compressed(PATH, **g)
    print("retry: solved", sum(r["ok"] for r in res), "of", len(res), "; still unsolved:", [r["i"] for r in res if not r["ok"]], flush=True)
    flat = [e for s in study for e in s]
    if skip_study:
        sys.exit(0)
    with open(os.path.join(ROOT, "profiles", "r05_hard_exits_study.json"), "w") as f:
        json.dump(dict(what="instances of tests/golden/solutions_hard.npz the interior-point iteration exits -7 on: SciPy SLSQP and trust-constr on the "
                            "reference NLP (reference callbacks) from four starts each", runs=flat,
                       feasible_optima_found={str(i): int(sum(e["feasible_optimum"] for e in flat if e["instance"] == i)) for i, _, _ in exits}), f, indent=1)
    print("exit study:", {i: sum(e["feasible_optimum"] for e in flat if e["instance"] == i) for i, _, _ in exits})
