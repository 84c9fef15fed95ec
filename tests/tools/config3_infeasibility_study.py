#!/usr/bin/env python3
"""configs[3]: are the instances the interior-point solver gives up on (exit -7 / not converged) feasible at all?
SciPy SLSQP on the NLP assembled from the REFERENCE callbacks (oracle/_ref), started from the planner's cold start, on a sample
of those instances, next to a sample of instances the solver converges on (control group).  Build container only.
   python tests/tools/config3_infeasibility_study.py [n_failing] [n_control] [workers] -> profiles/r03_config3_slsqp_study.json"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import gen_golden as G  # noqa: E402
from forces_resilient_planner_amd import workloads as W  # noqa: E402
import tests.oracle_lib as OL  # noqa: E402


def job(a):
    w1, tag = a
    nlp = G.RefNLP(w1["N"], w1["M"], w1["model"], w1["xinit"], w1["params"], w1["nfaces"])
    t = time.time()
    res = nlp.solve(w1["x0"].ravel().copy(), maxiter=400)
    Z = res.x
    c = nlp.ineq(Z)
    viol = float(max(0.0, -c.min())) if c.size else 0.0
    lb, ub = np.tile(nlp.lb, w1["N"]), np.tile(nlp.ub, w1["N"])
    return dict(tag=tag, status=int(res.status), nit=int(res.nit), eq=float(np.max(np.abs(nlp.eq(Z)))), ineq=viol,
                bound=float(max(0.0, (lb - Z).max(), (Z - ub).max())), secs=time.time() - t)


if __name__ == "__main__":
    nf_, nc_, workers = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 200), (2, 40), (3, 8)))
    B = 4096
    w = W.config3(B)
    z, fl, info = OL.solve_batch(w, OL.default_options(diverge_mu=10.0), nthreads=workers)
    it = np.array([i.it for i in info])
    bad = np.nonzero(fl != 1)[0]; good = np.nonzero(fl == 1)[0]
    print(f"oracle on {B} instances: converged {len(good)} ({len(good) / B:.4f}), exit flags of the rest {np.unique(fl[bad], return_counts=True)}", flush=True)
    rng = np.random.default_rng(1)
    pick = lambda idx, n: rng.choice(idx, size=min(n, len(idx)), replace=False)
    one = lambda b: dict(N=w["N"], M=w["M"], model=w["model"], xinit=w["xinit"][b], params=w["params"][b], nfaces=w["nfaces"][b], x0=w["x0"][b])
    jobs = [(one(b), "solver_failed") for b in pick(bad, nf_)] + [(one(b), "solver_converged") for b in pick(good, nc_)]
    t0 = time.time()
    with Pool(workers) as p:
        out = p.map(job, jobs, chunksize=1)
    feas = lambda r: r["status"] == 0 and r["eq"] < 1e-6 and r["ineq"] < 1e-6 and r["bound"] < 1e-9
    summary = {}
    for tag in ("solver_failed", "solver_converged"):
        rs = [r for r in out if r["tag"] == tag]
        summary[tag] = dict(n=len(rs), slsqp_feasible_optimum=sum(feas(r) for r in rs),
                            slsqp_status_counts={str(k): int(v) for k, v in zip(*np.unique([r["status"] for r in rs], return_counts=True))},
                            median_constraint_violation_when_not_solved=float(np.median([max(r["eq"], r["ineq"]) for r in rs if not feas(r)] or [0.0])))
    res = dict(what="SciPy SLSQP (reference-callback NLP, cold start, ftol 1e-13, <= 400 iterations) on configs[3] instances; status 0 = optimum found, "
                    "4 = inequality constraints incompatible, 8 = positive directional derivative, 9 = iteration limit",
               batch=B, oracle_converged_frac=len(good) / B, summary=summary, seconds=time.time() - t0, workers=workers)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "r03_config3_slsqp_study.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
