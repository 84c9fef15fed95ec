#!/usr/bin/env python3
"""GPU box: WHERE do the two implementations part on an exit-flag mismatch of tests/tools/soak.py?
Replays the soak's launch sequence up to each listed (kind, B, seed), then solves the SAME batch with maxit = m on the HIP path and on
the oracle for growing m (bisection on the first m at which the named problem's iterates differ by more than 1e-9) and prints, around
that iteration, what each side reports: residuals, mu, centring parameter, step lengths, Gauss-Newton redos -- one line per mismatch.

    python tests/tools/soak_diverge.py "kind,B,seed,problem" ...        (defaults: the mismatches listed in profiles/r05_soak.txt)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads  # noqa: E402
import tests.oracle_lib as OL  # noqa: E402

DEFAULT = ["4,3286,335800933,1100", "3,1928,470741047,277", "3,3455,56926856,264", "4,3542,243681085,1836", "2,2028,459693920,1466"]


def replay(target):
    """The workload of soak.py's launch (kind, B, seed): the same draws in the same order."""
    rng = np.random.default_rng(2026)
    for _ in range(200000):
        kind = int(rng.integers(0, 5)); B = int(rng.integers(1, 5000)); seed = int(rng.integers(0, 1 << 30))
        if kind == 0: mk = lambda: workloads.config1(B, seed=seed)
        elif kind == 1: m_ = int(rng.integers(0, 2)); mk = lambda: workloads.config2(B, seed=seed, model=m_)
        elif kind == 2: N_ = int(rng.integers(2, 41)); mk = lambda: workloads.config3(min(B, 1500), seed=seed, N=N_, M=15)
        elif kind == 3: N_ = int(rng.integers(41, 65)); M_ = int(rng.integers(15, 31)); mk = lambda: workloads.config3(min(B, 800), seed=seed, N=N_, M=M_)
        else: m_ = int(rng.integers(0, 2)); mk = lambda: workloads.config_hard(min(B, 2000), seed=seed, model=m_)
        if (kind, B, seed) == target:
            return mk()
    raise SystemExit(f"launch {target} is not in the soak's sequence")


def both(w, b, m):
    og = solver.default_options(maxit=m); oo = OL.default_options(maxit=m)
    z, fl, it, info = solver.solve_batch_host(w, og)
    # (the oracle on the named problem alone: its arithmetic does not depend on the batch; the HIP path on the whole batch, so that the
    # launch takes the kernel variant the soak's launch took)
    zo, flo, i = OL.solve_one(w["xinit"][b], w["x0"][b], w["params"][b], w["nfaces"][b], int(w["N"]), int(w["M"]), int(w["model"]), oo)
    return dict(dz=float(np.max(np.abs(z[b] - zo))), fl=(int(fl[b]), int(flo)), it=(int(it[b]), int(i.it)),
                gpu=dict(eq=info[b, 0], ineq=info[b, 1], stat=info[b, 2], comp=info[b, 3], mu=info[b, 5], step=info[b, 6], gn_redos=info[b, 7], mu_aff=info[b, 8], sigma=info[b, 9], step_aff=info[b, 10]),
                orc=dict(eq=i.res_eq, ineq=i.res_ineq, stat=i.rsnorm, comp=i.rcompnorm, mu=i.mu, step=getattr(i, "step_cc", float("nan")), gn_redos=getattr(i, "nfallback", -1),
                         mu_aff=getattr(i, "mu_aff", float("nan")), sigma=getattr(i, "sigma", float("nan")), step_aff=getattr(i, "step_aff", float("nan"))))


def fmt(d):
    return " ".join(f"{k} {float(v):.3e}" for k, v in d.items())


for spec in (sys.argv[1:] or DEFAULT):
    kind, B, seed, b = (int(x) for x in spec.split(","))
    w = replay((kind, B, seed))
    full = both(w, b, 200)
    print(f"\n== launch kind {kind} B {B} seed {seed} (N {w['N']}, M {w['M']}, {len(w['xinit'])} problems), problem {b}: flags gpu/oracle {full['fl']}, iterations {full['it']}", flush=True)
    hi = max(1, min(full["it"]))
    if both(w, b, hi)["dz"] <= 1e-9:
        print(f"   the plans z agree to 1e-9 through iteration {hi}: the flags part on the termination test -- what each side reports around it (m = maxit):")
        for m in range(max(1, hi - 4), hi + 6):
            r = both(w, b, m)
            print(f"   m {m:3d} |dz| {r['dz']:.1e} flags {r['fl']} its {r['it']}\n      gpu    {fmt(r['gpu'])}\n      oracle {fmt(r['orc'])}")
        continue
    lo = 0  # iterates after `lo` iterations agree, after `hi` they do not
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if both(w, b, mid)["dz"] <= 1e-9: lo = mid
        else: hi = mid
    growth = []
    for m in (1, max(1, lo // 2), lo, hi, hi + 1, hi + 3):
        r = both(w, b, m); growth.append((m, r["dz"]))
    print("   |dz| after m iterations: " + "  ".join(f"{m}: {d:.2e}" for m, d in sorted(set(growth))))
    for m in (lo, hi):
        if m < 1: continue
        r = both(w, b, m)
        print(f"   after iteration {m} (|dz| {r['dz']:.2e})\n      gpu    {fmt(r['gpu'])}\n      oracle {fmt(r['orc'])}")
    a, o = both(w, b, hi)["gpu"], both(w, b, hi)["orc"]
    why = []
    if a["gn_redos"] != o["gn_redos"]: why.append(f"a Gauss-Newton redo on one side only (indefinite pivot block: gpu {int(a['gn_redos'])}, oracle {int(o['gn_redos'])} redos)")
    if abs(a["step"] - o["step"]) > 1e-6 * max(1.0, abs(o["step"])): why.append(f"the step length differs (gpu {a['step']:.6f}, oracle {o['step']:.6f}: another constraint limits the fraction-to-boundary rule)")
    if abs(a["sigma"] - o["sigma"]) > 1e-6 * max(1e-3, abs(o["sigma"])): why.append(f"the centring parameter differs (gpu {a['sigma']:.3e}, oracle {o['sigma']:.3e})")
    print(f"   FIRST iteration with |dz| > 1e-9: {hi} of {full['it']}: " + ("; ".join(why) if why else "no discrete event: rounding differences of the Newton direction, amplified by the conditioning of the iteration"), flush=True)
