"""CPU tests (no GPU): pin the oracle.

 * orc_stage_eval vs the golden vectors produced by the reference's own CasADi callbacks (G1), and --
   when oracle/_ref was built in this container -- vs the live reference callbacks on fresh points;
 * orc_solve vs the committed SciPy SLSQP solutions of the reference NLP (G3) and SURVEY Appendix B.
The ForcesPro binary itself is licence-locked (-100): solver-level parity with it is unpinned.
"""
import ctypes
import os

import numpy as np
import pytest

from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import workloads

from . import oracle_lib as OL


def _sc(st):
    return 0 if st == 0 else (2 if st == 19 else 1)


def test_stage_functions_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "stage_vectors.npz"))
    assert g["z"].shape[0] >= 200
    for i in range(g["z"].shape[0]):
        st = int(g["stage"][i])
        o = OL.stage_eval(g["z"][i], g["p"][i], 30, _sc(st), int(g["model"][i]))
        keys = ["f", "gf", "h", "Jh"] + ([] if st == 19 else ["c", "Jc"])
        for key in keys:
            ref = g[key][i] if key != "f" else g[key][i, 0]
            assert np.max(np.abs(o[key] - ref) / (1 + np.abs(ref))) < 1e-12, (i, key)


@pytest.mark.skipif(not OL.ref_model_available(), reason="oracle/_ref not built (reference tree absent)")
def test_stage_functions_match_live_reference_callbacks():
    rng = np.random.default_rng(7)
    lb, ub = L.bounds()
    D = ctypes.POINTER(ctypes.c_double)
    for model, name in ((0, "normal"), (1, "final")):
        lib = ctypes.CDLL(os.path.join(OL.ORC_DIR, "_ref", f"libref_model_{name}.so"))
        fn = getattr(lib, f"FORCESNLPsolver_{name}_casadi2forces")
        for t in range(300):
            st = [0, 3, 19][t % 3]
            z = lb + (ub - lb) * rng.random(17)
            p = np.zeros(130); p[:10] = rng.uniform(-3, 3, 10); p[6:9] = rng.uniform(0.1, 90, 3)
            nf = int(rng.integers(0, 31))
            p[10:10 + 3 * nf] = rng.normal(size=3 * nf); p[100:100 + nf] = rng.normal(size=nf)
            f = np.zeros(1); gf = np.zeros(17); c = np.zeros(13); Jc = np.zeros(221); h = np.zeros(30); Jh = np.zeros(510)
            y = np.zeros(13); lam = np.zeros(64)
            fn(*[a.ctypes.data_as(D) for a in (z, y, lam, p, f, gf, c, Jc, h, Jh)], None, st, 0, 0)
            o = OL.stage_eval(z, p, 30, _sc(st), model)
            assert abs(o["f"] - f[0]) / (1 + abs(f[0])) < 1e-12
            for key, ref in (("gf", gf), ("h", h), ("Jh", Jh)) + ((() if st == 19 else (("c", c), ("Jc", Jc)))):
                assert np.max(np.abs(o[key] - ref) / (1 + np.abs(ref))) < 1e-12, (t, key)


@pytest.mark.parametrize("model,fext,fstar", [(0, (0, 0, 0), 23.1594329641), (1, (0, 0, 0), 48.4610568794),
                                              (0, (1.5, -2.0, 0.5), 16.3615772657)])
def test_solver_known_answers_appendix_b(model, fext, fstar):
    w = workloads.config0(model, fext, workloads.NORMAL_WEIGHTS)
    z, fl, info = OL.solve_batch(w)
    assert fl[0] == 1
    # at the reference's tolerances (complementarity <= 1e-4 per pair) the objective is within ~(active pairs) x mu
    assert abs(info[0].pobj - fstar) / fstar < 1e-5
    zt, flt, infot = OL.solve_batch(w, OL.default_options(tol_stat=1e-9, tol_eq=1e-9, tol_ineq=1e-9, tol_comp=1e-9))
    assert flt[0] == 1 and abs(infot[0].pobj - fstar) / fstar < 1e-8
    if model == 0 and fext[0] == 0:
        assert np.allclose(z[0, 0, :8], [0, 0.573693778, 0, 7.4752233435, 0, 0.5099500276, 0, 7.4752233456], atol=1e-4)
        assert abs(z[0, 19, 11] - 2.0) < 1e-4  # vx at its bound on the last stage


@pytest.mark.parametrize("fam", ["config0", "config1", "config2", "config3"])
def test_solver_matches_scipy_fixtures(fam, golden_dir):
    g = np.load(os.path.join(golden_dir, f"solutions_{fam}.npz"))
    N, M = int(g["N"]), int(g["M"])
    n = g["z"].shape[0]
    nconv = 0
    for i in range(n):
        z, fl, info = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
        if g["status"][i] != 0:
            # SLSQP could not solve it either (infeasible corridor/dynamics combination): the oracle must
            # report a non-optimal flag or a point SciPy simply missed -- never a NaN
            assert np.all(np.isfinite(z))
            continue
        assert fl == 1, (fam, i, fl)
        nconv += 1
        # at the reference's tolerances (1e-4) a weakly active bound (multiplier ~ 0) is only resolved to
        # ~sqrt(tol_comp) ~ 3e-3 by any interior-point method; the tight-tolerance solve below must hit SLSQP's point
        assert np.max(np.abs(z - g["z"][i])) < 5e-3, (fam, i, np.max(np.abs(z - g["z"][i])))
        zt, flt, _ = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]),
                                  OL.default_options(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8))
        assert flt == 1 and np.max(np.abs(zt - g["z"][i])) < 3e-4, (fam, i, flt, np.max(np.abs(zt - g["z"][i])))
        assert abs(info.pobj - g["f"][i]) / max(1e-9, abs(g["f"][i])) < 1e-4
        assert info.res_eq <= 1e-4 and info.rsnorm <= 1e-4 and info.rcompnorm <= 1e-4 and info.res_ineq <= 1e-4
    assert nconv == int((g["status"] == 0).sum()) and nconv >= 0.7 * n


@pytest.mark.parametrize("fam", ["config1", "config2", "hard"])
def test_twisted_solve_reaches_the_same_points(fam, golden_dir):
    """orc_options.twist (what frp_nmpc_options.twist runs on the GPU): the Newton system solved from both ends of the
    horizon -- stages 0..m-1 by an arrival-cost recursion with the pinned x_0 as a 1e12 penalty, stages m..N-1 by the backward
    Riccati recursion, a 13 x 13 system where they meet -- is the SAME linear system, so the iteration is the plain one up to the
    penalty's rounding: same flags, same iteration counts (but for a residual within rounding of its tolerance), same points;
    it also converges at 1e-8 tolerances (the penalty leaves x_0 ~1e-9 off xinit)."""
    g = np.load(os.path.join(golden_dir, f"solutions_{fam}.npz"))
    N, M = int(g["N"]), int(g["M"])
    good = np.where(g["status"] == 0)[0][:40]
    tight = dict(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)
    same_it = 0
    for m in (-1, 3, N - 2):
        for i in good:
            a = (g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
            z0, f0, i0 = OL.solve_one(*a)
            z1, f1, i1 = OL.solve_one(*a, OL.default_options(twist=m))
            assert f0 == 1 and f1 == 1, (fam, m, i, f0, f1)
            assert np.max(np.abs(z1 - z0)) < 2e-4 and abs(i1.pobj - i0.pobj) < 1e-5 * (1 + abs(i0.pobj)), (fam, m, i, np.max(np.abs(z1 - z0)))
            same_it += int(i1.it == i0.it)
            if m == -1:
                zt, ft, _ = OL.solve_one(*a, OL.default_options(twist=m, **tight))
                assert ft == 1 and np.max(np.abs(zt - g["z"][i])) < (6e-4 if fam == "hard" else 3e-4), (fam, i, ft)
    assert same_it >= 0.95 * 3 * len(good)
    # outside its range (m = N - 1, N < 4) the option is ignored: bit-identical to the plain solve
    a = (g["xinit"][0], g["x0"][0], g["params"][0], g["nfaces"][0], N, M, int(g["model"][0]))
    z0, _, _ = OL.solve_one(*a)
    z1, _, _ = OL.solve_one(*a, OL.default_options(twist=N - 1))
    assert np.array_equal(z0, z1)


def hard_family_check(g, z, fl, pobj, zt, flt):
    """The judgement both the oracle (here) and the HIP path (tests/test_gpu_parity.py) are held to on tests/golden/solutions_hard.npz
    (workloads.config_hard: reference 3..5 m away, |f_ext| 6..9 m/s^2, 5..10 cm of corridor slack, post-replan warm starts):
    every instance SLSQP solved on the reference callbacks is solved, at the DEFAULT options (no workload-tuned diverge_mu, no
    line search), to SLSQP's point -- or, on the printed exception list, to a KKT point of the reference NLP with a LOWER objective
    (a non-convex NLP has several; which one a method lands on is not a parity question), certified with the reference's callbacks only."""
    N, M = int(g["N"]), int(g["M"])
    good = np.where(g["status"] == 0)[0]
    #  the fixture's 'near-retry' entries -- SLSQP started 0.02 / 0.002 from the solver's OWN 1e-8 solution (tests/tools/extend_hard_golden.py) --
    # confirm that point as a local solution of the reference NLP but are near-guaranteed to agree with the solver under test: they are held to
    # the same bounds below, and they do NOT count towards the independent agreement the family is asked for
    retry = np.array([str(s_).startswith("near-retry") for s_ in g["start"]])
    independent = good[~retry[good]]
    assert len(independent) >= 100, len(independent)
    assert np.all(fl[good] == 1), ("not converged", good[fl[good] != 1], fl[good][fl[good] != 1])
    assert np.all(flt[good] == 1)
    exceptions = []
    for i in good:
        dz, dzt = np.max(np.abs(z[i] - g["z"][i])), np.max(np.abs(zt[i] - g["z"][i]))
        df = abs(pobj[i] - g["f"][i]) / max(1e-9, abs(g["f"][i]))
        if dz < 5e-3 and dzt < 6e-4 and df < 1e-4:
            continue
        # another local solution: it must be a better one, and a KKT point by the reference's own functions
        assert pobj[i] < g["f"][i] - 1e-6 * abs(g["f"][i]), (i, dz, pobj[i], g["f"][i])
        k = OL.reference_kkt(zt[i], g["xinit"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
        assert k["stat"] < 1e-6 and k["eq"] < 1e-8 and k["ineq"] < 1e-8 and k["bound"] < 1e-8, (i, k)
        exceptions.append((int(i), float(dz), float(pobj[i]), float(g["f"][i])))
    print("hard family: %d SLSQP-solved instances (%d from a start independent of the solver under test, %d 'near-retry' confirmations), all converged; "
          "other (better) KKT point on:" % (len(good), len(independent), len(good) - len(independent)), exceptions)
    assert len(exceptions) <= 0.03 * len(good)
    #  EVERY instance the solver converges on -- also the ones SciPy did not solve, which used to be counted as wins
    # on status alone -- is a KKT point of the reference NLP by the reference's own functions (the 1e-8 solve: stationarity <= 1e-6,
    # feasibility <= 1e-8; three instances read 1e-6 .. 3e-6 with the bounded-least-squares multipliers, see tests/tools/twist_certify.py)
    conv = np.where(flt == 1)[0]
    worst = dict(stat=0.0, eq=0.0, ineq=0.0, bound=0.0)
    for i in conv:
        k = OL.reference_kkt(zt[i], g["xinit"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
        assert k["stat"] < 3e-6 and k["eq"] < 1e-8 and k["ineq"] < 1e-8 and k["bound"] < 1e-8, (int(i), k)
        worst = {q: max(worst[q], k[q]) for q in worst}
    unsolved = [int(i) for i in range(len(fl)) if fl[i] != 1]
    print("hard family: %d of %d instances converge at 1e-8, every one a certified KKT point of the reference NLP (worst residuals %s); "
          "exits at the default options: %s" % (len(conv), len(fl), worst, [(i, int(fl[i])) for i in unsolved]))
    return exceptions


def test_solver_converges_on_the_hard_family_without_globalisation(golden_dir):
    """VERDICT r03 item 3.  The reference solver carries a filter line search (FORCESNLPsolver_normal.h:89-95); this one takes
    Mehrotra steps with a multiplier safeguard.  The claim "that is enough" rests on this family, not on benign fixtures."""
    g = np.load(os.path.join(golden_dir, "solutions_hard.npz"))
    N, M = int(g["N"]), int(g["M"])
    n = g["z"].shape[0]
    z = np.zeros((n, N, 17)); zt = np.zeros((n, N, 17)); fl = np.zeros(n, dtype=int); flt = np.zeros(n, dtype=int); pobj = np.zeros(n)
    tight = OL.default_options(tol_stat=1e-8, tol_eq=1e-8, tol_ineq=1e-8, tol_comp=1e-8)
    for i in range(n):
        z[i], fl[i], info = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]))
        pobj[i] = info.pobj
        zt[i], flt[i], _ = OL.solve_one(g["xinit"][i], g["x0"][i], g["params"][i], g["nfaces"][i], N, M, int(g["model"][i]), tight)
    assert np.all(np.isfinite(z))
    hard_family_check(g, z, fl, pobj, zt, flt)


def test_padding_detection_equals_explicit_face_counts():
    w = workloads.config2(16)
    za, fa, _ = OL.solve_batch(w)
    w2 = dict(w); w2["nfaces"] = None
    zb = np.zeros_like(za)
    for b in range(16):
        zb[b], fl, _ = OL.solve_one(w["xinit"][b], w["x0"][b], w["params"][b], None, w["N"], w["M"], w["model"])
        assert fl == fa[b]
    assert np.max(np.abs(za - zb)) == 0.0


def test_infeasible_problem_reports_failure_not_nan():
    w = workloads.config2(4)
    # shrink the corridor of stage 10 to an empty set: x <= -1 and -x <= -1 in the path frame
    p = w["params"].copy()
    p[:, 10, L.NPRE + 3 * 30 + 0] = p[:, 10, L.NPRE + 3 * 30 + 1] * -1.0 - 5.0
    w["params"] = p
    z, fl, info = OL.solve_batch(w)
    assert np.all(fl != 1)
    assert np.all(np.isfinite(z))


@pytest.mark.skipif(not OL.ref_model_available(), reason="oracle/_ref not built (reference tree absent)")
def test_plans_satisfy_reference_kkt_measured_with_reference_callbacks():
    """the returned plan is a KKT point of the REFERENCE NLP.  Gradient, dynamics, corridor rows
    and their Jacobians come from the reference's CasADi callbacks only (OL.reference_kkt); solved to 1e-8
    so that the multipliers of inactive constraints (<= tol/slack) drop below the check."""
    for w in (workloads.config0(), workloads.config2(4), workloads.config3(2)):
        for tol, stat_max, eq_max in ((1e-4, 5e-3, 1e-4), (1e-8, 1e-6, 1e-8)):
            opt = OL.default_options(tol_stat=tol, tol_eq=tol, tol_ineq=tol, tol_comp=tol)
            z, fl, _ = OL.solve_batch(w, opt)
            assert np.all(fl == 1)
            for b in range(z.shape[0]):
                k = OL.reference_kkt(z[b], w["xinit"][b], w["params"][b], w["nfaces"][b], w["N"], w["M"], w["model"])
                assert k["stat"] < stat_max and k["eq"] < eq_max and k["ineq"] < eq_max and k["bound"] < eq_max, (tol, b, k)


def test_tube_oracle_is_self_consistent():
    """Row f-2's oracle (oracle/tube_oracle.py) cannot be pinned against the reference (Eigen absent).  What can be
    checked on the CPU: its Sylvester solutions satisfy the reference's equation (nmpc_solver.cpp:592-596), they
    equal the Gramian integral the GPU kernel evaluates (computed here with scipy.integrate) whenever the Sylvester
    operator is nonsingular (no two eigenvalues of Phi summing to zero -- Phi is Hurwitz near zero yaw, where the
    reference's gain was designed, and merely nonsingular elsewhere), and sqrtm3 is the principal root."""
    import sys
    import scipy.integrate as si
    import scipy.linalg as sl
    sys.path.insert(0, OL.ROOT)
    from oracle import tube_oracle as T
    c = T.default_consts()
    rng = np.random.default_rng(11)
    lb, ub = L.bounds()
    for _ in range(6):
        z = lb + (ub - lb) * rng.random(17)
        z[11:14] = rng.uniform(-6, 6, 3)
        Phi, R = T.update_matrix(z[14:17], z[11:14], z[3], c)
        lam = np.linalg.eigvals(Phi)
        assert np.min(np.abs(lam[:, None] + lam[None, :])) > 1e-2
        assert np.max(np.abs(R @ R.T - np.eye(3))) < 1e-14
        t = c["Ts"]
        d = np.zeros(9); d[4] = 1.0
        Nt = t * 0.25 * np.outer(d, d)
        Em = sl.expm(-Phi * t)
        W = Nt - Em @ Nt @ Em.T
        X = sl.solve_sylvester(Phi, Phi.T, W)
        assert np.max(np.abs(Phi @ X + X @ Phi.T - W)) < 1e-14
        G, _ = si.quad_vec(lambda s: np.outer(sl.expm(-Phi * s) @ d, sl.expm(-Phi * s) @ d), 0, t, epsabs=1e-16, epsrel=1e-13)
        assert np.max(np.abs(X - t * 0.25 * G)) < 1e-13 * np.abs(X).max() + 1e-18
    Phi0, _ = T.update_matrix(np.array([0.05, -0.03, 0.1]), np.array([1.0, 0.5, -0.2]), c["mass"] * 9.81, c)
    assert np.linalg.eigvals(Phi0).real.max() < 0
    E = T.tube_one(workloads.config2(1)["x0"][0])
    assert np.linalg.eigvalsh(E).min() > 0 and np.max(np.abs(E - np.swapaxes(E, -1, -2))) < 1e-13
    # stage 0 is the bare ego ellipsoid (nmpc_solver.cpp:503-506)
    assert np.max(np.abs(np.linalg.eigvalsh(E[0]) - np.array([c["ego_h"], c["ego_r"], c["ego_r"]]))) < 1e-13


def test_corridor_oracle_properties():
    """Row f-3's oracle (oracle/corridor_oracle.py, parity unpinned): what the reference's algorithm guarantees by
    construction must hold -- no obstacle strictly inside a polytope, the seed inside, the local box always present,
    a polytope reused exactly while the inflated tube ellipsoid fits (nmpc_solver.cpp:291-313)."""
    import sys
    sys.path.insert(0, OL.ROOT)
    from oracle import corridor_oracle as C
    rng = np.random.default_rng(3)
    cloud = np.c_[rng.uniform(-3, 8, 3000), rng.uniform(-4, 4, 3000), rng.uniform(-0.5, 3, 3000)]
    cloud = cloud[np.hypot(cloud[:, 1], cloud[:, 2] - 1.0) > 0.6]
    N = 20
    ref = np.c_[np.linspace(0, 4, N), 0.1 * np.sin(np.linspace(0, 3, N)), np.ones(N)]
    yaw = np.full(N, 0.1)
    E = np.tile(np.diag([0.27, 0.27, 0.05]), (N, 1, 1))
    idx, polys = C.corridor_one(ref, yaw, E, cloud)
    assert idx[0] == 0 and np.all(np.diff(idx) >= 0) and np.all(np.diff(idx) <= 1) and idx[-1] == len(polys) - 1 >= 1
    for k, (A, b) in enumerate(polys):
        assert not np.any(np.all(cloud @ A.T - b < -1e-9, axis=1))
        assert np.max(np.abs(np.linalg.norm(A, axis=1) - 1)) < 1e-12
        first = int(np.argmax(idx == k))
        assert np.all(A @ ref[first] - b < 0)                     # the seed point is inside
        assert len(b) >= 6                                        # six local-box rows close every polytope
    for i in range(1, N):
        A, b = polys[idx[i - 1]] if idx[i] == idx[i - 1] else polys[idx[i] - 1]
        fits = np.all(A @ ref[i] - (b - 1.1 * np.linalg.norm(A @ E[i].T, axis=1)) <= 0)
        assert fits == (idx[i] == idx[i - 1])
    # an obstacle inside the seed ellipsoid shrinks it: the ellipsoid ends up touching, not containing, the obstacle
    p1, p2 = np.array([0.0, 0, 1]), np.array([1.5, 0, 1])
    obs = np.array([[0.7, 0.3, 1.1], [0.9, -0.2, 0.8], [0.2, 0.1, 1.4]])
    Cm, d = C.find_ellipsoid(p1, p2, obs)
    dist = np.linalg.norm((obs - d) @ np.linalg.inv(Cm).T, axis=1)
    assert dist.min() > 1 - 1e-9 and np.isclose(dist, 1, atol=1e-9).any()


def test_reference_oracle_known_answers():
    """Row f-4's oracle: straight path along +y sampled at the stage times; the yaw filter converges geometrically
    (0.2 / 0.8 weights, nmpc_solver.cpp:858) from the plan's stage-1 yaw towards pi/2; stage 0 raises the replan flag
    only when it is more than 1 m from the plan (:136)."""
    import sys
    sys.path.insert(0, OL.ROOT)
    from oracle import reference_oracle as R
    K, N = 60, 20
    path = np.c_[np.zeros(K), 0.1 * np.arange(K), np.ones(K)]
    plan = np.zeros((N + 1, 17)); plan[1, 8:11] = [0.0, 0.3, 1.0]; plan[1, 16] = 0.0
    pos, yaw, replan = R.references_one(path, K, 0.125, plan, N)
    assert np.allclose(pos[:, 1], 0.1 * (0.125 / 0.05 + np.arange(N)), atol=1e-12) and not replan
    assert np.allclose(yaw, np.arctan2(1, 0) * (1 - 0.2 ** np.arange(1, N + 1)), atol=1e-12)
    plan[1, 8:11] = [1.5, 0.3, 1.0]
    assert R.references_one(path, K, 0.125, plan, N)[2]
    # past the end of the path: last sample, direction shorter than 0.1 m -> the yaw is held by the filter
    pos, yaw, _ = R.references_one(path, K, 0.05 * (K + 3), plan, N)
    assert np.all(pos == path[-1]) and np.allclose(yaw, 0.0)
