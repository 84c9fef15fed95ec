"""The 4 exits -7 of the hard family: SLSQP on the reference NLP from two starts each (the quick half of the study in
extend_hard_golden.py).  Appends one JSON line per run to gpurun_out/hard_exits_slsqp.jsonl.   python tests/tools/hard_exits_slsqp.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests','tools'))
import gen_golden as G
import tests.oracle_lib as OL
from multiprocessing import Pool
g = dict(np.load(os.path.join(ROOT,'tests','golden','solutions_hard.npz'), allow_pickle=False))
N, M = int(g["N"]), int(g["M"])
def run(args):
    i, name = args
    nlp = G.RefNLP(N, M, int(g["model"][i]), g["xinit"][i], g["params"][i], g["nfaces"][i])
    tight = OL.default_options(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)
    zo, fl, info = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]), tight)
    z0 = g["x0"][i].ravel().copy() if name == "cold" else zo.ravel().copy()
    t = time.time()
    res = nlp.solve(z0, maxiter=400)
    c = nlp.ineq(res.x)
    eq, ineq = float(np.max(np.abs(nlp.eq(res.x)))), float(max(0.0, -c.min())) if c.size else 0.0
    k = OL.reference_kkt(res.x.reshape(N, 17), g["xinit"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
    out = dict(instance=int(i), start=name, method="SLSQP", status=int(res.status), nit=int(res.nit), f=float(nlp.fun(res.x)[0]), eq=eq, ineq=ineq, kkt=k,
               feasible_optimum=bool(eq < 1e-8 and ineq < 1e-8 and k["stat"] < 1e-6), ipm_flag=int(fl), secs=time.time() - t)
    with open(os.path.join(ROOT, 'gpurun_out', 'hard_exits_slsqp.jsonl'), 'a') as f: f.write(json.dumps(out) + "\n")
    return out
if __name__ == "__main__":
    jobs = [(i, s) for i in (2, 30, 62, 134) for s in ("cold", "ipm_last_iterate")]
    with Pool(3) as p:
        for r in p.imap_unordered(run, jobs): print(r["instance"], r["start"], r["status"], r["feasible_optimum"], round(r["secs"]), flush=True)
