"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the golden fixtures.

Tolerances (FP64 end to end, like the reference -- FORCESNLPsolver_normal.h:55-58):
  stage functions : <= 1e-12 relative vs the reference-callback golden vectors
  solver          : same interior-point iteration as the oracle -> identical exit flags, iteration counts
                    equal on >= 95 % of the problems (identical math, different summation order) and
                    |z_gpu - z_oracle|_inf <= 1e-6; vs the SciPy fixtures |df|/f <= 1e-4 and |dz|_inf <= 5e-3 at the
                    reference's 1e-4 tolerances, <= 3e-4 (SLSQP's own accuracy) when solved to 1e-8
"""
import ctypes
import os

import numpy as np
import pytest

from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd import solver, workloads

from . import oracle_lib as OL

pytestmark = pytest.mark.gpu

# The launches the four-problems-per-CU kernel variants cover (plain solve, N <= 20, at most 6 corridor rows per stage) go to them only
# beyond three resident workgroups per CU worth of problems (frp_nmpc_set_q4_min_batch); the tests below run twice, the second time with
# that threshold at zero, so that both sets of variants see every case at the tests' batch sizes.
TWICE = {"test_dropin_abi_config0_known_answers", "test_batch_matches_oracle", "test_batch_matches_scipy_fixtures",
         "test_hip_path_converges_on_the_hard_family_at_default_options", "test_gauss_newton_mode_matches_oracle",
         "test_iteration_limit_returns_maxit_and_the_last_iterate", "test_indefinite_cost_reports_factorization_error",
         "test_more_faces_than_workspace_is_a_parameter_error", "test_infeasible_corridor_reports_failure_not_nan",
         "test_padding_detection_without_face_counts", "test_receding_horizon_warm_start_matches_oracle",
         "test_horizon_lengths_cover_every_lane_mapping", "test_every_kernel_variant_reports_the_oracles_numbers",
         "test_queue_order_hint_changes_the_order_and_nothing_else", "test_gpu_plans_satisfy_reference_kkt_measured_with_reference_callbacks",
         "test_per_problem_model_equals_two_single_model_batches", "test_receding_horizon_on_device_matches_host_loop",
         "test_full_tick_replayed_from_a_hipgraph_equals_eager_launches", "test_twist_outside_its_range_is_the_plain_solve"}


@pytest.fixture
def kernel_set(request):
    q4 = request.param == "q4"
    old = solver.lib().frp_nmpc_set_q4_min_batch(0 if q4 else -1)
    yield request.param
    solver.lib().frp_nmpc_set_q4_min_batch(old)


def pytest_generate_tests(metafunc):
    if metafunc.function.__name__ in TWICE:
        metafunc.fixturenames.append("kernel_set")
        metafunc.parametrize("kernel_set", ["3cu", "q4"], indirect=True)


def _stage_class(st):
    return 0 if st == 0 else (2 if st == 19 else 1)


def test_stage_eval_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "stage_vectors.npz"))
    n = g["z"].shape[0]
    # the batched callback evaluates stage k of an N-stage problem: build N=20 problems whose stage
    # `stage` carries the golden (z, p) and read back that stage only
    z = np.zeros((n, 20, 17)); p = np.zeros((n, 20, 130))
    lb, ub = L.bounds()
    z[:] = 0.5 * (lb + ub)
    p[:, :, 6:9] = 1.0
    for i in range(n):
        z[i, g["stage"][i]] = g["z"][i]
        p[i, g["stage"][i]] = g["p"][i]
    for model in (0, 1):
        out = solver.stage_eval_host(z, p, 30, model)
        for i in np.where(g["model"] == model)[0]:
            st = int(g["stage"][i])
            for key, ref in (("f", g["f"][i, 0]), ("gf", g["gf"][i]), ("h", g["h"][i])):
                got = out[key][i, st]
                assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-12, (i, key)
            if st != 19:
                assert np.max(np.abs(out["c"][i, st] - g["c"][i]) / (1 + np.abs(g["c"][i]))) < 1e-12
                assert np.max(np.abs(out["Jc"][i, st] - g["Jc"][i]) / (1 + np.abs(g["Jc"][i]))) < 1e-12


def _forces_call(w0, model):
    p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
    p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel()
    p.num_of_threads = 1
    fn = solver.lib().FORCESNLPsolver_normal_solve if model == 0 else solver.lib().FORCESNLPsolver_final_solve
    flag = fn(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
    return flag, np.array(o.x).reshape(20, 17), info


@pytest.mark.parametrize("model,fext,fstar", [(0, (0, 0, 0), 23.1594329641), (1, (0, 0, 0), 48.4610568794),
                                              (0, (1.5, -2.0, 0.5), 16.3615772657)])
def test_dropin_abi_config0_known_answers(model, fext, fstar, golden_dir):
    """configs[0] through FORCESNLPsolver_{normal,final}_solve; f* from SURVEY Appendix B (SciPy on the
    reference callbacks), z* from the committed fixture."""
    w0 = workloads.config0(model, fext, workloads.NORMAL_WEIGHTS)
    flag, z, info = _forces_call(w0, model)
    assert flag == 1
    assert abs(info.pobj - fstar) / fstar < 1e-4
    g = np.load(os.path.join(golden_dir, "solutions_config0.npz"))
    idx = {(0, 0.0): 0, (1, 0.0): 1, (0, 1.5): 2}[(model, float(fext[0]))]
    assert np.max(np.abs(z - g["z"][idx])) < 1e-3
    assert info.res_eq <= 1e-4 and info.rsnorm <= 1e-4 and info.rcompnorm <= 1e-4
    assert info.solvetime > 0


@pytest.mark.parametrize("cfg,B", [(1, 256), (2, 512), (3, 256)])
def test_batch_matches_oracle(cfg, B):
    w = workloads.CONFIGS[cfg](B)
    z, fl, it, info = solver.solve_batch_host(w)
    zo, flo, io = OL.solve_batch(w)
    ito = np.array([i.it for i in io])
    assert (fl == flo).mean() >= 0.995, (fl != flo).sum()
    ok = (fl == 1) & (flo == 1)
    assert ok.mean() > 0.8
    assert (it[ok] == ito[ok]).mean() >= 0.99
    # same iteration count -> the same iterates up to summation order; a problem that stops one iteration apart (a
    # residual within rounding of its tolerance) still agrees to the accuracy the 1e-4 tolerances define
    same = ok & (it == ito)
    assert np.max(np.abs(z[same] - zo[same])) < 1e-6
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-3
    pobj_o = np.array([i.pobj for i in io])
    assert np.max(np.abs(info[same, 4] - pobj_o[same])) < 1e-6 * (1 + np.abs(info[same, 4]).max())
    assert np.max(np.abs(info[ok, 4] - pobj_o[ok]) / (1 + np.abs(pobj_o[ok]))) < 1e-4
    # the diagnostic block (info.mu_aff / sigma / step_aff of FORCESNLPsolver_normal.h:275-289, dgap = sum s * lambda = mu * rows)
    for col, name in ((8, "mu_aff"), (9, "sigma"), (10, "step_aff"), (6, "step_cc"), (5, "mu")):
        ref = np.array([getattr(i, name) for i in io])
        assert np.max(np.abs(info[same, col] - ref[same]) / (1e-3 + np.abs(ref[same]))) < 1e-3, name
    rows = 34 * w["N"] + w["nfaces"].sum(1)
    assert np.max(np.abs(info[ok, 11] - info[ok, 5] * rows[ok])) < 1e-9 * rows.max()


@pytest.mark.parametrize("fam", ["config1", "config2", "config3"])
def test_batch_matches_scipy_fixtures(fam, golden_dir):
    g = np.load(os.path.join(golden_dir, f"solutions_{fam}.npz"), allow_pickle=False)
    N, M = int(g["N"]), int(g["M"])
    good = g["status"] == 0
    for model in np.unique(g["model"]):
        sel = np.where((g["model"] == model) & good)[0]
        w = dict(xinit=g["xinit"][sel], x0=g["x0"][sel], params=g["params"][sel], nfaces=g["nfaces"][sel],
                 N=N, M=M, model=int(model))
        z, fl, it, info = solver.solve_batch_host(w)
        conv = fl == 1
        assert conv.all(), (fam, int(model), np.where(~conv)[0])  # every problem SLSQP solved is solved here too (as by the oracle)
        dz = np.max(np.abs(z[conv] - g["z"][sel][conv]), axis=(1, 2))
        df = np.abs(info[conv, 4] - g["f"][sel][conv]) / np.maximum(1e-9, np.abs(g["f"][sel][conv]))
        # reference tolerances (1e-4): weakly active bounds are resolved to ~sqrt(tol_comp) only (see test_oracle.py)
        assert np.max(dz) < 5e-3, np.max(dz)
        assert np.max(df) < 1e-4, np.max(df)
        opt = solver.default_options()
        opt.tol_stat = opt.tol_eq = opt.tol_ineq = opt.tol_comp = 1e-8
        zt, flt, _, _ = solver.solve_batch_host(w, opt)
        assert np.all(flt[conv] == 1)
        assert np.max(np.abs(zt[conv] - g["z"][sel][conv])) < 3e-4


def test_hip_path_converges_on_the_hard_family_at_default_options(golden_dir):
    """VERDICT r03 item 3 / r05 "What's weak" 1a: the 192 hard-but-feasible instances of tests/golden/solutions_hard.npz (reference 3..5 m away,
    |f_ext| 6..9 m/s^2, 5..10 cm of corridor slack after tightening, post-replan warm starts).  SLSQP on the reference callbacks solves 166 of
    them (151 from a start independent of this solver, 15 'near-retry' confirmations; per kind far 31 / force 46 / tight 42 / replan 47 of 48):
    the HIP path converges on ALL of those at the default diverge_mu, to SLSQP's point or to a certified better KKT point.  The stronger statement
    covers the 22 instances SLSQP stalls on as well (17 of them of the `far` kind): EVERY instance the HIP path converges on (188 of 192) is a
    KKT point of the reference NLP measured with the reference's own callbacks (tests/test_oracle.py:hard_family_check), and scipy's trust-constr
    ends at the solver's point on 21 of those 22 from a perturbed start and on 20 of 22 from the planner's cold start (the other runs unfinished
    at 3000 iterations, at a higher objective; profiles/r06_hard_trust_constr_summary.txt, tests/tools/hard_trust_constr.py)."""
    from .test_oracle import hard_family_check
    g = np.load(os.path.join(golden_dir, "solutions_hard.npz"), allow_pickle=False)
    N, M = int(g["N"]), int(g["M"])
    n = g["z"].shape[0]
    z = np.zeros((n, N, 17)); zt = np.zeros((n, N, 17)); fl = np.zeros(n, dtype=int); flt = np.zeros(n, dtype=int); pobj = np.zeros(n)
    tight = solver.default_options()
    tight.tol_stat = tight.tol_eq = tight.tol_ineq = tight.tol_comp = 1e-8
    for model in np.unique(g["model"]):
        sel = np.where(g["model"] == model)[0]
        w = dict(xinit=g["xinit"][sel], x0=g["x0"][sel], params=g["params"][sel], nfaces=g["nfaces"][sel], N=N, M=M, model=int(model))
        z[sel], fl[sel], it, info = solver.solve_batch_host(w)      # default options: diverge_mu 1e3
        pobj[sel] = info[:, 4]
        zt[sel], flt[sel], _, _ = solver.solve_batch_host(w, tight)
        # and the oracle agrees instance by instance (same algorithm): flags everywhere, iterates where the counts are equal
        zo, flo, io = OL.solve_batch(w)
        assert np.array_equal(fl[sel], flo)
        same = (flo == 1) & (it == np.array([i.it for i in io]))
        assert same.mean() > 0.9 and np.max(np.abs(z[sel][same] - zo[same])) < 1e-5
    hard_family_check(g, z, fl, pobj, zt, flt)


@pytest.mark.parametrize("cfg,m", [(1, -1), (2, -1), (2, 3), (2, 18)])
def test_twisted_solve_matches_the_oracles_twisted_solve(cfg, m):
    """frp_nmpc_options.twist: stages 0..m-1 eliminated forward by the model wave while the Riccati wave runs m..N-1
    backward.  Checked against the oracle running the same split (orc_options.twist) AND against the plain HIP solve: the linear
    system is the same one, the iterates differ by the rounding of the 1e12 penalty that pins x_0 (~1e-5 after an iteration, measured;
    tools/dbg/twist_check.py), flags and iteration counts agree, converged points agree far inside the tolerances (1e-4)."""
    w = workloads.CONFIGS[cfg](256)
    z, fl, it, info = solver.solve_batch_host(w, solver.default_options(twist=m))
    zp, flp, itp, infop = solver.solve_batch_host(w)
    zo, flo, io = OL.solve_batch(w, OL.default_options(twist=m))
    ito = np.array([i.it for i in io])
    assert np.array_equal(fl, flo) and np.array_equal(fl, flp)
    ok = fl == 1
    assert ok.mean() > 0.8 and np.all(np.isfinite(z))
    assert (it[ok] == ito[ok]).mean() >= 0.97 and (it[ok] == itp[ok]).mean() >= 0.97
    assert np.max(np.abs(z[ok] - zo[ok])) < 2e-4 and np.max(np.abs(z[ok] - zp[ok])) < 2e-4
    pobj_o = np.array([i.pobj for i in io])
    assert np.max(np.abs(info[ok, 4] - pobj_o[ok]) / (1 + np.abs(pobj_o[ok]))) < 1e-5
    # one iteration in: the same Newton step up to the penalty's rounding
    z1, _, _, _ = solver.solve_batch_host(w, solver.default_options(twist=m, maxit=1))
    zo1, _, _ = OL.solve_batch(w, OL.default_options(twist=m, maxit=1))
    assert np.max(np.abs(z1 - zo1)) < 2e-4
    # the residuals the exit tests read are evaluated exactly as in the plain solve: converged means the same thing
    assert info[ok, 0].max() <= 1e-4 and info[ok, 2].max() <= 1e-4 and info[ok, 3].max() <= 1e-4


def test_twisted_solve_on_the_fixture_families(golden_dir):
    """Every SLSQP-solved instance of the config1 / config2 / hard fixture families converges with twist = -1 too, to SLSQP's point
    at default and at 1e-8 tolerances (the penalty leaves x_0 ~1e-9 off xinit: no obstacle to tol_eq = 1e-8)."""
    for fam, tol_t in (("config1", 3e-4), ("config2", 3e-4), ("hard", 6e-4)):
        g = np.load(os.path.join(golden_dir, f"solutions_{fam}.npz"), allow_pickle=False)
        N, M = int(g["N"]), int(g["M"])
        good = g["status"] == 0
        for model in np.unique(g["model"]):
            sel = np.where((g["model"] == model) & good)[0]
            w = dict(xinit=g["xinit"][sel], x0=g["x0"][sel], params=g["params"][sel], nfaces=g["nfaces"][sel], N=N, M=M, model=int(model))
            z, fl, it, info = solver.solve_batch_host(w, solver.default_options(twist=-1))
            zp, flp, itp, _ = solver.solve_batch_host(w)
            assert np.all(fl == 1), (fam, int(model), np.where(fl != 1)[0])
            assert np.max(np.abs(z - zp)) < 2e-4 and (it == itp).mean() >= 0.95
            opt = solver.default_options(twist=-1)
            opt.tol_stat = opt.tol_eq = opt.tol_ineq = opt.tol_comp = 1e-8
            zt, flt, _, _ = solver.solve_batch_host(w, opt)
            assert np.all(flt == 1)
            d = np.max(np.abs(zt - g["z"][sel]), axis=(1, 2))
            assert (d < tol_t).mean() >= 0.97, (fam, int(model), d.max())  # (the hard family has a few other, better KKT points: hard_family_check)


@pytest.mark.skipif(not OL.ref_model_available(), reason="oracle/_ref not built (reference tree absent)")
def test_twisted_solve_is_certified_where_it_leaves_the_plain_solve():
    """the twisted solve is an inexact-Newton variant -- on hard instances a few pairs (plain, twisted) end more than
    1e-3 apart at the default tolerances.  Both variants solve every such instance again at 1e-8 tolerances (the twisted variants
    finish with exact Newton steps: TW_EXACT_BELOW) and BOTH end points must be KKT points of the reference NLP as measured with the
    reference's callbacks only (stationarity <= 1e-6, feasibility <= 1e-8); a twisted point with a worse objective is a printed
    exception, at most 3 per mille of the batch (tools/twist_soak.py runs the same certification on 147 k problems:
    profiles/r05_twist_soak.txt)."""
    from .tools import twist_certify as TC
    w = workloads.config_hard(2048, seed=303, model=0)   # (the batch of profiles/r04_twist_soak.txt with the most pairs apart: 7)
    z0, f0, _, _ = solver.solve_batch_host(w)
    z1, f1, _, _ = solver.solve_batch_host(w, solver.default_options(twist=-1))
    c = TC.certify(w, z0, f0, z1, f1, label="hard seed 303 model 0")
    print(f"twist certification: {c['pairs']} pairs more than 1e-3 apart in 2048 hard instances, {c['certified']} certified "
          f"({c['same_point']} the same point at 1e-8 tolerances); twisted objective worse: {c['worse']}")
    assert not c["uncertified"], c["uncertified"]
    assert c["pairs"] == c["certified"] + len(c["plain_fails_too"]) and c["certified"] >= 4 and len(c["worse"]) <= 6, c


def test_twist_outside_its_range_is_the_plain_solve():
    """N > 20, N < 4 or m outside 2..N-2: the option is ignored (bit-identical results), like the oracle's."""
    w = workloads.config2(64)
    zp, flp, itp, _ = solver.solve_batch_host(w)
    for m in (19, 25, 1, 0):  # N - 1, beyond the horizon, a single forward stage, off
        z, fl, it, _ = solver.solve_batch_host(w, solver.default_options(twist=m))
        assert np.array_equal(z, zp) and np.array_equal(it, itp)
    w3 = workloads.config3(32)  # (the default horizon of configs[3]: N = 30)
    z3p, _, it3p, _ = solver.solve_batch_host(w3)
    z3, _, it3, _ = solver.solve_batch_host(w3, solver.default_options(twist=-1))
    assert np.array_equal(z3, z3p) and np.array_equal(it3, it3p)


def test_full_size_properties():
    """BASELINE configs[2] at full size (B=4096): size-independent properties of the returned plans."""
    w = workloads.config2(4096)
    z, fl, it, info = solver.solve_batch_host(w)
    assert (fl == 1).mean() > 0.99
    ok = fl == 1
    lb, ub = L.bounds()
    assert np.all(z[ok] >= lb - 1e-4) and np.all(z[ok] <= ub + 1e-4)
    # initial condition and input carry (E z_{k+1} rows 9..12, mpc_generator_normal.m:4-5)
    assert np.max(np.abs(z[ok][:, 0, 8:17] - w["xinit"][ok])) < 1e-4
    assert np.max(np.abs(z[ok][:, 1:, 4:8] - z[ok][:, :-1, 0:4])) < 1e-4
    # corridor rows satisfied, dynamics residual small (re-evaluated by the batched callback kernel)
    ev = solver.stage_eval_host(z, w["params"], w["M"], w["model"], want=("c", "h"))
    assert np.max(ev["h"][ok]) < 1e-5 + 1e-4
    assert np.max(np.abs(ev["c"][ok][:, :-1, 0:9] - z[ok][:, 1:, 8:17])) < 1e-4
    assert info[ok, 0].max() <= 1e-4 and info[ok, 2].max() <= 1e-4 and info[ok, 3].max() <= 1e-4


def test_gauss_newton_mode_matches_oracle():
    w = workloads.config2(128)
    z, fl, it, info = solver.solve_batch_host(w, solver.default_options(hessian=0))
    zo, flo, io = OL.solve_batch(w, OL.default_options(hessian=0))
    assert np.array_equal(fl, flo)
    ok = fl == 1
    ito = np.array([i.it for i in io])
    assert (it[ok] == ito[ok]).mean() >= 0.95
    assert np.max(np.abs(z[ok & (it == ito)] - zo[ok & (it == ito)])) < 1e-6
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-3


@pytest.mark.parametrize("mu0", [0.2, 5.0])
def test_barrier_start_option_matches_oracle(mu0):
    """frp_nmpc_options.mu0 (what a receding-horizon caller lowers for its warm-started ticks, tools/full_tick_bench.py): the kernel and
    the oracle run the same iteration from the same barrier parameter -- flags, iteration counts, iterates -- on cold problems (both kernel
    sets: 6-row and 30-row layouts) and on problems warm-started at their own solution shifted by a stage."""
    for w in (workloads.config2(192), workloads.config3(96, N=20)):
        z, fl, it, info = solver.solve_batch_host(w, solver.default_options(mu0=mu0))
        zo, flo, io = OL.solve_batch(w, OL.default_options(mu0=mu0))
        ito = np.array([i.it for i in io])
        assert np.array_equal(fl, flo)
        ok = fl == 1
        assert (it[ok] == ito[ok]).mean() >= 0.95
        assert np.max(np.abs(z[ok & (it == ito)] - zo[ok & (it == ito)])) < 1e-6
    w = workloads.config2(192)
    z1, fl1, _, _ = solver.solve_batch_host(w)
    warm = dict(w); warm["x0"] = np.concatenate([z1[:, 1:], z1[:, -1:]], axis=1)  # the shift initialisation of a receding-horizon tick
    z, fl, it, info = solver.solve_batch_host(warm, solver.default_options(mu0=mu0))
    zo, flo, io = OL.solve_batch(warm, OL.default_options(mu0=mu0))
    ito = np.array([i.it for i in io])
    assert np.array_equal(fl, flo) and (it == ito).mean() >= 0.95
    assert np.max(np.abs(z[(fl == 1) & (it == ito)] - zo[(fl == 1) & (it == ito)])) < 1e-6


def test_long_horizon_uses_wide_lane_mapping():
    """N = 40 > 32 exercises the NP = 64 instantiation (one stage per lane in the element-wise phases)."""
    w = workloads.config3(32, N=40, M=15)
    z, fl, it, info = solver.solve_batch_host(w)
    zo, flo, io = OL.solve_batch(w)
    assert (fl == flo).mean() >= 0.95
    ok = (fl == 1) & (flo == 1)
    assert ok.sum() >= 16
    same = ok & (it == np.array([i.it for i in io]))
    assert same.sum() >= 0.9 * ok.sum() and np.max(np.abs(z[same] - zo[same])) < 1e-6
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-3


def _libc():
    lc = ctypes.CDLL(None)
    lc.fopen.restype = ctypes.c_void_p; lc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lc.fclose.argtypes = [ctypes.c_void_p]
    return lc


@pytest.mark.skipif(not OL.ref_model_available(), reason="oracle/_ref not built (reference tree absent)")
def test_dropin_called_the_way_the_reference_calls_it(tmp_path):
    """forces_normal.cpp:30,139 passes &FORCESNLPsolver_normal_casadi2forces and (in its debug builds) a FILE*: the
    reference's own compiled callback must be accepted (probe against the device model), give the same plan as a NULL
    callback, and the summary must carry the reference's strings (normal.h printlevel 1)."""
    ref = ctypes.CDLL(os.path.join(OL.ORC_DIR, "_ref", "libref_model_normal.so"))
    cb = ctypes.cast(ref.FORCESNLPsolver_normal_casadi2forces, solver.EXTFUNC)
    w0 = workloads.config0(0, (0.0, 0.0, 0.0), workloads.NORMAL_WEIGHTS)
    flag0, z0, info0 = _forces_call(w0, 0)
    p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
    p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel()
    p.num_of_threads = 1
    lc = _libc()
    path = str(tmp_path / "summary.txt").encode()
    fs = lc.fopen(path, b"w")
    f = solver.lib().FORCESNLPsolver_normal_solve
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, solver.EXTFUNC]
    try:
        flag = f(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), fs, cb)
        lc.fclose(fs)
        text = open(path.decode()).read()
        assert flag == 1 and flag0 == 1
        assert np.array_equal(np.array(o.x).reshape(20, 17), z0)
        assert "OPTIMAL (within EQTOL=" in text and "Solve time:" in text and "iterations)" in text

        # a callback that implements a different model is refused with PARAM_VALUE (-11): there is no host path that
        # could honour it, and silently solving the built-in model instead would be wrong
        @solver.EXTFUNC
        def other_model(x, y, lam, par, pobj, g, c, Jeq, h, Jineq, H, stage, it, tid):
            cb(x, y, lam, par, pobj, g, c, Jeq, h, Jineq, H, stage, it, tid)
            if pobj:
                pobj[0] += 1.0
        fs2 = lc.fopen(path, b"w")
        flag_bad = f(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), fs2, other_model)
        lc.fclose(fs2)
        assert flag_bad == L.PARAM_VALUE_ERROR
        assert "differs from the built-in device model" in open(path.decode()).read()
        assert f(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, cb) == 1  # the good callback is probed again
    finally:
        f.argtypes = None


def test_iteration_limit_returns_maxit_and_the_last_iterate():
    """exit 0 (MAXITREACHED, normal.h:113): every problem stops after maxit iterations with its current iterate in the
    output, the same iterate the oracle holds after as many iterations."""
    w = workloads.config2(64)
    z, fl, it, info = solver.solve_batch_host(w, solver.default_options(maxit=2))
    zo, flo, io = OL.solve_batch(w, OL.default_options(maxit=2))
    assert np.all(fl == 0) and np.all(flo == 0) and np.all(it == 2)
    assert np.max(np.abs(z - zo)) < 1e-9
    assert not np.array_equal(z, w["x0"])
    # through the drop-in struct: info.it counts the iterations done
    w0 = workloads.config0(0, (0.0, 0.0, 0.0), workloads.NORMAL_WEIGHTS)
    flag, z0, info0 = _forces_call(w0, 0)
    assert flag == 1 and info0.it >= 3


def test_indefinite_cost_reports_factorization_error():
    """exit -5 (FACTORIZATION_ERROR, normal.h:119): a negative input-rate weight makes the reduced Hessian of every stage
    indefinite under the exact AND the Gauss-Newton Hessian; both implementations give up in the first iteration."""
    w = workloads.config2(16)
    p = w["params"].copy()
    p[:, :, 8] = -50.0  # w_input_rate (setup.m:62)
    w["params"] = p
    z, fl, it, info = solver.solve_batch_host(w)
    zo, flo, _ = OL.solve_batch(w)
    assert np.all(fl == L.FACTORIZATION_ERROR) and np.array_equal(fl, flo)
    assert np.all(it == 0) and np.all(np.isfinite(z))


def test_fresh_fleet_cold_starts_without_touching_exitflag():
    """A new DeviceFleet must cold-start every planner on its first full tick: exitflag starts at zero (never recycled
    allocator memory), so coldstart_kernel's exitflag != 1 test fires for all of them."""
    import torch
    B, N = 8, 20
    junk = torch.ones((4096,), dtype=torch.int32, device="cuda:0"); del junk  # a block of ones for the allocator to recycle
    fl = solver.DeviceFleet(B, N, 30, 6, 0, workloads.NORMAL_WEIGHTS)
    assert int(fl.solver.exitflag.abs().sum()) == 0 and int(fl.solver.iters.abs().sum()) == 0
    state = torch.zeros((B, 9), dtype=torch.float64, device="cuda:0"); state[:, 2] = 1.0
    fl.mpc_output.fill_(123.0)
    fl.coldstart(state, only_failed=True)
    torch.cuda.synchronize()
    mo = fl.mpc_output.cpu().numpy()
    assert np.allclose(mo[:, :, 3], 7.3) and np.allclose(mo[:, :, 10], 1.0) and np.allclose(mo[:, :, :3], 0.0)


def test_more_faces_than_workspace_is_a_parameter_error():
    w = workloads.config3(4)
    z, fl, it, info = solver.solve_batch_host(w, MF=5)   # stages have 6..15 live rows
    assert np.all(fl == L.PARAM_VALUE_ERROR)
    assert np.array_equal(z, w["x0"])                   # output = the caller's initial guess, untouched


def test_face_count_beyond_the_parameter_block_is_a_parameter_error_also_in_the_ordered_launch():
    """B above the resident slots (768) runs the launch-order kernels BEFORE the solver: a face count larger than the
    rows a stage's parameter block holds must come back as PARAM_VALUE, not fault in the ordering kernel (its corridor
    loop is clamped to M).  The poisoned problems include the LAST one of the batch, whose rows end the allocation."""
    w = workloads.config2(1024)
    nf = w["nfaces"].copy()
    bad = np.array([0, 500, 1023])
    nf[bad, :] = w["M"] + 50
    nf[1023, 19] = 2 ** 30
    w["nfaces"] = nf
    z, fl, it, info = solver.solve_batch_host(w, MF=6)
    assert np.all(fl[bad] == L.PARAM_VALUE_ERROR)
    good = np.setdiff1d(np.arange(1024), bad)
    assert (fl[good] == 1).mean() > 0.99
    assert np.array_equal(z[bad], w["x0"][bad])


def test_infeasible_corridor_reports_failure_not_nan():
    w = workloads.config2(8)
    p = w["params"].copy()
    p[:, 10, L.NPRE + 90 + 0] = -p[:, 10, L.NPRE + 90 + 1] - 5.0   # face 0 and face 1 of stage 10 now exclude each other
    w["params"] = p
    z, fl, it, info = solver.solve_batch_host(w)
    zo, flo, _ = OL.solve_batch(w)
    assert np.all(fl != 1) and np.all(np.isfinite(z))
    assert np.array_equal(fl, flo)


def test_padding_detection_without_face_counts():
    w = workloads.config2(32)
    za, fa, ia, _ = solver.solve_batch_host(w)
    w2 = dict(w); w2["nfaces"] = None
    zb, fb, ib, _ = solver.solve_batch_host(w2, MF=6)
    assert np.array_equal(fa, fb) and np.array_equal(ia, ib) and np.array_equal(za, zb)


def test_cpp_adapter_solves_through_the_c_abi(tmp_path):
    """The C++ mirror of FORCESNormal (csrc/frp_adapter.hpp) packs + solves; same answers as the Python path."""
    from .test_capi_cpu import _harness, _write_harness_input
    w = workloads.config2(6)
    inp = tmp_path / "in.bin"; out = tmp_path / "out.bin"
    _write_harness_input(inp, w, (7.0, 1.0, 80.0, 12.0, 0.5), 6)
    import subprocess
    subprocess.check_call([_harness(), "solve", str(inp), str(out)])
    raw = np.fromfile(out, dtype=np.uint8)
    z = raw[:6 * 20 * 17 * 8].view(np.float64).reshape(6, 20, 17); fl = raw[6 * 20 * 17 * 8:].view(np.int32)
    z2, fl2, _, _ = solver.solve_batch_host(w)
    assert np.array_equal(fl, fl2) and np.all(fl == 1)
    assert np.max(np.abs(z - z2)) < 1e-9


def test_receding_horizon_warm_start_matches_oracle():
    """BASELINE configs[4] in miniature: Monte-Carlo f_ext around one nominal problem, shift-initialised
    receding horizon (forces_normal.cpp:62-97 + nmpc_solver.cpp:524-543) for 6 ticks."""
    from forces_resilient_planner_amd import receding
    w0 = workloads.config4_nominal(48, ticks=6)

    def gpu(w):
        z, fl, it, _ = solver.solve_batch_host(w)
        return z, fl, it

    def cpu(w):
        z, fl, info = OL.solve_batch(w)
        return z, fl, np.array([i.it for i in info], dtype=np.int32)

    fg, ig, mg = receding.run(w0, 6, gpu)
    fc, ic, mc = receding.run(w0, 6, cpu)
    assert np.array_equal(fg, fc)
    assert (fg == 1).mean() > 0.9
    assert np.max(np.abs(mg - mc)) < 1e-5
    # warm-started ticks need no more iterations than the cold-started first one
    assert ig[1:].mean() <= ig[0].mean() + 0.5


@pytest.mark.parametrize("N", [2, 3, 6, 7, 8, 11, 12, 18, 20, 25, 33, 47, 64])
def test_horizon_lengths_cover_every_lane_mapping(N):
    """Stage strides 20 / 32 / 64 (3, 2, 1 row groups per wavefront) against the oracle, and every shape of the vector
    sweeps' stage loops: passes of four stages plus a tail of 1..4 (forward) / 0..3 (backward), no pass at all (N < 6)."""
    w = workloads.config3(24, N=N, M=15)
    z, fl, it, info = solver.solve_batch_host(w)
    zo, flo, io = OL.solve_batch(w)
    # thresholds = what the randomised soak shows (profiles/r02_soak.txt: 5.0 M problems over all horizons, no launch with
    # more than 0.5 % differing flags; profiles/r02_sweep_check.txt: none in 147 k): a 24-problem batch must agree exactly
    assert np.array_equal(fl, flo), (fl, flo)
    ok = (fl == 1) & (flo == 1)
    assert ok.sum() >= 12
    same = ok & (it == np.array([i.it for i in io]))
    assert same.sum() >= ok.sum() - 1 and np.max(np.abs(z[same] - zo[same])) < 1e-6
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-3


@pytest.mark.parametrize("N,M,B", [(20, 6, 64), (20, 15, 64), (20, 30, 64), (30, 6, 48), (30, 15, 48), (30, 30, 48), (48, 8, 24), (48, 30, 24)])
def test_every_kernel_variant_reports_the_oracles_numbers(N, M, B):
    """One batch per kernel variant of the LDS-resident solver (stage stride 20 / 32 / 64 x corridor rows in registers or
    re-read from the parameters; the variants are separate template instantiations in two translation units with their own
    code-generation flags): flags, iteration counts, iterates AND every reported quantity -- residual norms, objective,
    mu -- against the oracle.  `nfaces` is withheld so that the padding detection runs and MF = M selects the variant."""
    w = workloads.config3(B, N=N, M=M)
    wn = dict(w); wn["nfaces"] = None
    z, fl, it, info = solver.solve_batch_host(wn)
    zo, flo, io = OL.solve_batch(w)
    assert np.array_equal(fl, flo)
    ok = (fl == 1) & (flo == 1) & (it == np.array([i.it for i in io]))
    assert ok.sum() >= B // 2
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-6
    # residuals of a converged iterate are rounding noise below ~1e-7: absolute; objective and mu: relative
    for col, name in ((0, "res_eq"), (1, "res_ineq"), (2, "rsnorm"), (3, "rcompnorm")):
        ref = np.array([getattr(i, name) for i in io])
        assert np.max(np.abs(info[ok, col] - ref[ok])) < 1e-7, name
    for col, name in ((4, "pobj"), (5, "mu")):
        ref = np.array([getattr(i, name) for i in io])
        assert np.max(np.abs(info[ok, col] - ref[ok]) / (1e-9 + np.abs(ref[ok]))) < 1e-6, name


def test_device_packing_matches_the_adapter():
    """frp_nmpc_pack_batch / frp_nmpc_update_batch against the host adapter (adapter.py, itself
    checked against the C++ mirror of forces_normal.cpp): copies bit-exact, the tightened offsets b - ||E a|| to 2 ulp
    (sum order / FMA contraction), identical zero padding and face counts."""
    import torch
    from forces_resilient_planner_amd.adapter import ForcesAdapter, update_forces_results
    w = workloads.config3(64)                       # N = 30, 6..15 faces per stage, per-stage tube matrices
    B, N, M = w["B"], w["N"], w["M"]
    F = w["poly_A"].shape[2]
    wts = workloads._weights(w["model"])
    fleet = solver.DeviceFleet(B, N, M, F, w["model"], wts)
    fleet.mpc_output.copy_(fleet.to_device(w["mpc_output"]))
    fleet.ellipsoid.copy_(fleet.to_device(w["E"]))
    fleet.poly_A.copy_(fleet.to_device(w["poly_A"])); fleet.poly_b.copy_(fleet.to_device(w["poly_b"]))
    fleet.poly_nfaces.copy_(fleet.to_device(w["nfaces"], dtype=torch.int32))
    fleet.pack(fleet.to_device(w["f_ext"]), fleet.to_device(w["ref_pos"]), fleet.to_device(w["ref_yaw"]))
    torch.cuda.synchronize()
    ds = fleet.solver
    assert np.array_equal(ds.xinit.cpu().numpy(), w["xinit"])
    assert np.array_equal(ds.x0.cpu().numpy(), w["x0"])
    assert np.array_equal(ds.nfaces.cpu().numpy(), w["nfaces"])
    p = ds.params.cpu().numpy()
    assert np.array_equal(p[:, :, :10 + 3 * M], w["params"][:, :, :10 + 3 * M])
    assert np.array_equal(p == 0.0, w["params"] == 0.0)
    assert np.max(np.abs(p - w["params"])) <= 4 * np.finfo(float).eps * (1 + np.abs(w["params"]).max())
    # solve from the device-packed inputs, then the device-side bookkeeping against the host one
    ds.solve(); fleet.update()
    torch.cuda.synchronize()
    z = ds.z.cpu().numpy(); fl = ds.exitflag.cpu().numpy()
    ref = w["mpc_output"].copy()
    upd = ref.copy(); upd[:, :N] = z; upd = update_forces_results(upd)
    ref[fl == 1] = upd[fl == 1]
    assert (fl == 1).mean() > 0.8
    assert np.array_equal(fleet.mpc_output.cpu().numpy(), ref)
    yaw = fleet.mpc_output[:, :N, 16].cpu().numpy()
    assert np.all(np.abs(yaw) <= np.pi + 1e-12)


def test_repeated_packs_write_only_what_changed_and_leave_the_full_image():
    """frp_nmpc_pack.padded_rows_are_zero (a fleet's second and later packs): the call writes the live rows of a stage and zeroes only those the previous call left live.
    With face counts that grow, shrink and drop to zero from call to call -- and other polytopes each time -- the parameter image equals that of a full pack into
    fresh buffers, bit for bit; DeviceSolver.upload takes the promise back."""
    import torch
    rng = np.random.default_rng(3)
    w = workloads.config3(48)
    B, N, M = w["B"], w["N"], w["M"]
    F = w["poly_A"].shape[2]
    wts = workloads._weights(w["model"])
    fleet = solver.DeviceFleet(B, N, M, F, w["model"], wts)
    fleet.mpc_output.copy_(fleet.to_device(w["mpc_output"])); fleet.ellipsoid.copy_(fleet.to_device(w["E"]))
    args = (fleet.to_device(w["f_ext"]), fleet.to_device(w["ref_pos"]), fleet.to_device(w["ref_yaw"]))
    for rep in range(6):
        nf = w["nfaces"].copy()
        if rep == 1: nf = np.maximum(nf - 3, 0)
        if rep == 2: nf = np.minimum(nf + 2, F)
        if rep == 3: nf[:, ::2] = 0
        if rep >= 4: nf = rng.integers(0, F + 1, size=nf.shape)
        A = w["poly_A"] * (1.0 + 0.01 * rep); b = w["poly_b"] + 0.1 * rep
        fleet.poly_A.copy_(fleet.to_device(A)); fleet.poly_b.copy_(fleet.to_device(b))
        fleet.poly_nfaces.copy_(fleet.to_device(nf, dtype=torch.int32))
        if rep == 5:  # somebody else writes the solver's buffers: the next pack must be a full one again
            fleet.solver.upload(dict(xinit=w["xinit"], x0=w["x0"], params=rng.normal(size=w["params"].shape), nfaces=w["nfaces"]))
        fleet.pack(*args)
        fresh = solver.DeviceFleet(B, N, M, F, w["model"], wts)
        fresh.solver.params.fill_(float("nan"))
        fresh.mpc_output.copy_(fleet.mpc_output); fresh.ellipsoid.copy_(fleet.ellipsoid)
        fresh.poly_A.copy_(fleet.poly_A); fresh.poly_b.copy_(fleet.poly_b); fresh.poly_nfaces.copy_(fleet.poly_nfaces)
        fresh.pack(*args)
        torch.cuda.synchronize()
        assert np.array_equal(fleet.solver.nfaces.cpu().numpy(), fresh.solver.nfaces.cpu().numpy())
        assert np.array_equal(fleet.solver.params.cpu().numpy(), fresh.solver.params.cpu().numpy()), rep
        assert np.all(np.isfinite(fresh.solver.params.cpu().numpy()))


@pytest.mark.parametrize("M", [0, 1, 7])
def test_device_packing_without_live_rows_and_small_layouts(M):
    """The packing kernel's three regions (header, A block, b block) at the edges: no corridor rows in the layout at all (M = 0: the
    stage block is its ten leading slots), one row, an odd count; nothing live -> everything behind the header is zero."""
    import torch
    w = workloads.config1(8)
    B, N = w["B"], w["N"]
    fleet = solver.DeviceFleet(B, N, M, max(M, 1), w["model"], workloads._weights(w["model"]))
    fleet.mpc_output.copy_(fleet.to_device(w["mpc_output"]))
    fleet.ellipsoid.copy_(fleet.to_device(w["E"]))
    fleet.poly_A.zero_(); fleet.poly_b.zero_(); fleet.poly_nfaces.zero_()
    fleet.solver.params.fill_(7.0)  # (stale contents must be overwritten, zeros included)
    fleet.pack(fleet.to_device(w["f_ext"]), fleet.to_device(w["ref_pos"]), fleet.to_device(w["ref_yaw"]))
    torch.cuda.synchronize()
    p = fleet.solver.params.cpu().numpy()
    assert p.shape[2] == 10 + 4 * M
    assert np.array_equal(p[:, :, :10], w["params"][:, :, :10]) and np.all(p[:, :, 10:] == 0.0)
    assert np.array_equal(fleet.solver.x0.cpu().numpy(), w["x0"]) and np.all(fleet.solver.nfaces.cpu().numpy() == 0)


def test_receding_horizon_on_device_matches_host_loop():
    """configs[4] in miniature with the whole tick on the GPU (pack -> solve -> update) against the host-driven loop."""
    from forces_resilient_planner_amd import receding
    w0 = workloads.config4_nominal(B=32, ticks=5)
    fg, ig, mg = receding.run(w0, 5, lambda w: solver.solve_batch_host(w)[:3])
    fd, idv, md, secs = receding.run_device(w0, 5)
    assert np.array_equal(fd, fg)
    assert (idv == ig).mean() >= 0.95
    assert np.max(np.abs(md - mg)) < 1e-6


def test_maximum_sizes_horizon_64_all_30_corridor_rows_live():
    """Edge of the supported range: N = 64 stages (one stage per lane), every one of the 30 corridor rows of the
    reference layout live (each random polytope row repeated: redundant faces, non-unique multipliers), batch 6;
    plus a batch of ONE and an N = 2 horizon."""
    w = workloads.config3(6, N=64, M=30)
    p = w["params"].copy(); nf = w["nfaces"].copy()
    for b in range(6):
        for k in range(64):
            n = nf[b, k]
            A = p[b, k, 10:10 + 3 * 30].reshape(30, 3); bb = p[b, k, 100:130]
            for j in range(n, 30):  # repeat the live rows, pushed out by 1 cm .. 16 cm so that they stay inactive-ish
                A[j] = A[j % n]; bb[j] = bb[j % n] + 0.01 * (1 + j // n)
            nf[b, k] = 30
    w2 = dict(w, params=p, nfaces=nf)
    z, fl, it, info = solver.solve_batch_host(w2)
    zo, flo, io = OL.solve_batch(w2)
    assert np.array_equal(fl, flo)
    ok = fl == 1
    assert ok.sum() >= 3
    assert np.max(np.abs(z[ok] - zo[ok])) < 1e-5
    # the redundant rows do not move the solution of the 6..15-row problem
    z1, fl1, _, _ = solver.solve_batch_host(w)
    both = ok & (fl1 == 1)
    assert np.max(np.abs(z[both] - z1[both])) < 5e-3
    # batch of one
    w1 = {k: (v[:1] if isinstance(v, np.ndarray) and v.shape[:1] == (6,) else v) for k, v in w2.items()}
    za, fla, _, _ = solver.solve_batch_host(w1)
    assert fla[0] == fl[0] and np.max(np.abs(za[0] - z[0])) < 1e-9
    # shortest horizon the API accepts
    ws = workloads.config3(8, N=2, M=15)
    zs, fls, its, _ = solver.solve_batch_host(ws)
    zso, flso, _ = OL.solve_batch(ws)
    assert np.array_equal(fls, flso) and np.max(np.abs(zs[fls == 1] - zso[fls == 1])) < 1e-6


def test_launch_order_is_a_permutation_for_hostile_keys():
    """More problems than resident workgroups -> the queue is ordered by the objective of the initial guess.  Keys that are
    NaN, infinite, equal or astronomically large must still give every problem exactly one solve: the clean problems come
    out as in a batch without the hostile ones, the poisoned ones report a failure flag."""
    B = 3000
    w = workloads.config2(B, seed=77)
    x0 = w["x0"].copy()
    bad_nan, bad_inf, bad_big, dup = np.arange(0, 40), np.arange(40, 60), np.arange(60, 80), np.arange(80, 400)
    x0[bad_nan, 3, 9] = np.nan
    x0[bad_inf, 5, 8] = np.inf
    x0[bad_big, :, 8] = 1e150        # objective ~1e300+: overflows the key
    x0[dup] = x0[80]                 # identical keys
    z, fl, it, info = solver.solve_batch_host(w, x0=x0)
    assert np.all(fl[bad_nan] == L.BADFUNCEVAL) and np.all(fl[bad_inf] != 1)
    clean = np.ones(B, dtype=bool); clean[:80] = False
    zc, flc, itc, _ = solver.solve_batch_host(w)   # same problems, original initial guesses
    keep = clean.copy(); keep[dup] = False          # (the dup block was given another initial guess)
    assert np.array_equal(fl[keep], flc[keep]) and np.array_equal(it[keep], itc[keep])
    assert np.max(np.abs(z[keep] - zc[keep])) == 0.0
    assert np.all(np.isfinite(z[clean]))


def test_launch_order_without_face_counts_and_at_other_horizons():
    """The key kernel reads every corridor row when the caller gives no face counts, and packs 64 // N problems per wavefront:
    ordered batches (more problems than resident workgroups) with nfaces = NULL and at horizons that pack 1, 2 and 3 problems
    per wavefront give the plans of the same problems solved in small, unordered batches.  (The high-residency variants are chosen by
    batch size and sum in another order: the threshold is set to zero for the comparison, so that the big batch and its pieces run the same
    variant -- the N = 30 case is then the three-per-CU variant of round 6 on both sides.)"""
    old = solver.lib().frp_nmpc_set_q4_min_batch(0)
    try:
        for N, B in ((20, 2400), (30, 1600), (48, 800)):
            w = workloads.config3(B, N=N, seed=5) if N != 20 else workloads.config2(B, seed=6)
            z, fl, it, _ = solver.solve_batch_host(dict(w, nfaces=None), MF=w["M"])
            zs, fls, its = [], [], []
            # pieces no larger than the resident workgroups of the variant (index order); at N = 20 larger than two problems per CU, so that they stay on the
            # variant of the big batch (smaller launches run builds of their own -- csrc/frp_ipm_lds_s2.hip -- that agree with it to rounding, not to the bit)
            P = 600 if N == 20 else 200
            for lo in range(0, B, P):
                sub = {k: (v[lo:lo + P] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in w.items()}
                a, b_, c, _ = solver.solve_batch_host(dict(sub, nfaces=None), MF=w["M"])
                zs.append(a); fls.append(b_); its.append(c)
            assert np.array_equal(fl, np.concatenate(fls)) and np.array_equal(it, np.concatenate(its))
            assert np.max(np.abs(z - np.concatenate(zs))) == 0.0
    finally:
        solver.lib().frp_nmpc_set_q4_min_batch(old)


def test_small_launches_with_many_rows_agree_with_the_batch_variants_and_the_oracle():
    """Launches of at most two problems per CU run builds of their own (csrc/frp_ipm_lds_s2.hip: two wavefronts per SIMD, rows in registers up to 30 per stage).
    The same problems inside a larger batch -- the three-per-CU variants, which re-read more than 15 rows per stage from the parameters -- give the same flags and
    iteration counts and plans within 1e-10; both match the oracle."""
    B = 700
    w = workloads.config2(B, seed=29)
    full = dict(w, nfaces=None)
    z, fl, it, _ = solver.solve_batch_host(full, MF=w["M"])            # 700 > two per CU: three-per-CU, rows re-read
    sub = {k: (v[:150] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in w.items()}
    zs, fls, its, _ = solver.solve_batch_host(dict(sub, nfaces=None), MF=w["M"])   # 150 problems: the small-launch build, 30 rows in registers
    assert np.array_equal(fls, fl[:150]) and np.array_equal(its, it[:150])
    assert np.max(np.abs(zs - z[:150])) < 1e-10
    zo, flo, _ = OL.solve_batch(sub)  # (the oracle takes the row counts: the trailing rows of the blocks are zero)
    assert np.array_equal(fls, flo) and np.max(np.abs(zs[fls == 1] - zo[fls == 1])) < 1e-6


def test_host_path_packs_live_rows_and_chunks_without_changing_a_plan():
    """frp_nmpc_solve_batch_host with explicit face counts packs the parameters to the live corridor rows while staging them and
    solves in growing chunks (B/16, B/4, rest): the plans are those of ONE launch on the caller's 30-row layout resident in HBM,
    bit for bit, also with another chunk split and with the chunking switched off."""
    import torch
    B = 4500
    w = workloads.config2(B, seed=11)
    ds = solver.DeviceSolver(B, w["N"], w["M"], int(w["nfaces"].max()), w["model"])
    ds.upload(w)
    ds.solve(); torch.cuda.synchronize()
    z0, f0, i0 = ds.z.cpu().numpy(), ds.exitflag.cpu().numpy(), ds.iters.cpu().numpy()
    for split in (None, "100,4400", "4500"):
        if split: os.environ["FRP_HOST_SPLIT"] = split
        try:
            z, fl, it, _ = solver.solve_batch_host(w)
        finally:
            os.environ.pop("FRP_HOST_SPLIT", None)
        assert np.array_equal(fl, f0) and np.array_equal(it, i0) and np.max(np.abs(z - z0)) == 0.0, split


def test_host_path_with_registered_buffers_stages_nothing_and_changes_no_plan():
    """frp_nmpc_host_register: with every array of the call pinned in place and mapped, frp_nmpc_solve_batch_host reads the inputs from the
    caller's memory by a gather kernel and lets the solver write in place -- the plans, flags, counts and diagnostics are those of the
    staged path, bit for bit; unregistering falls back to staging; a partly registered call is staged."""
    B = 2500
    w = workloads.config2(B, seed=17)
    w = {k: (np.ascontiguousarray(v, dtype=(np.int32 if k == "nfaces" else np.float64)) if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    ref = solver.solve_batch_host(w)
    out = tuple(np.full_like(a, -7) for a in ref)
    reg = [w["xinit"], w["x0"], w["params"], w["nfaces"]] + list(out)
    solver.host_register(*reg)
    try:
        solver.solve_batch_host(w, out=out)
        for a, b in zip(ref, out):
            assert np.array_equal(a, b)
        # partly registered: staged, same results
        solver.host_unregister(w["x0"])
        out2 = tuple(np.full_like(a, -7) for a in ref)
        solver.solve_batch_host(w, out=out2)
        for a, b in zip(ref, out2):
            assert np.array_equal(a, b)
    finally:
        solver.host_unregister(*[a for a in reg if a is not w["x0"]])
    assert solver.lib().frp_nmpc_host_unregister(w["xinit"].ctypes.data) != 0  # (already gone: an argument error, nothing else)
    assert not solver._registered  # the wrapper's references went with the registrations


def test_three_per_cu_variant_for_horizons_up_to_30_against_the_oracle_and_the_two_per_cu_variant():
    """Round 6 (csrc/frp_ipm_lds_q30.hip): launches of the plain solve with 20 < N <= 30 and <= 16 corridor rows that hold more than 7 x CUs problems run on
    the three-problems-per-CU variant (215-double records: P d aliased onto the p slots, S_xx in the L2 workspace, y+ formed by the lane of the stage).  At the
    DEFAULT threshold a batch of 2304 problems of configs[3] takes it; with the threshold out of reach the same batch runs two per CU: identical flags, the
    oracle's iteration counts on >= 99 % of the problems on both, iterates within 1e-6 of the oracle's and of each other where the counts agree
    (same iteration, other summation order); shorter horizons (N = 21, 25) and the final model likewise on a small batch with the threshold at zero."""
    lib = solver.lib()
    w = workloads.config3(2304, seed=21)
    zo, flo, io = OL.solve_batch(w, nthreads=16)
    ito = np.array([i.it for i in io])
    res = {}
    for name, thr in (("three_per_cu", -1), ("two_per_cu", 10 ** 9)):
        old = lib.frp_nmpc_set_q4_min_batch(thr)
        try:
            res[name] = solver.solve_batch_host(w)
        finally:
            lib.frp_nmpc_set_q4_min_batch(old)
        z, fl, it, info = res[name]
        assert np.array_equal(fl, flo), name
        same = (fl == 1) & (it == ito)
        assert (it == ito).mean() > 0.99 and np.max(np.abs(z[same] - zo[same])) < 1e-6, (name, (it == ito).mean())
    a, b = res["three_per_cu"], res["two_per_cu"]
    both = (a[1] == 1) & (a[2] == b[2])
    assert both.mean() > 0.85 and np.max(np.abs(a[0][both] - b[0][both])) < 1e-6 and not np.array_equal(a[0], b[0])  # (close, and NOT the same kernel)
    old = lib.frp_nmpc_set_q4_min_batch(0)
    try:
        for N, model in ((21, L.MODEL_NORMAL), (25, L.MODEL_FINAL), (30, L.MODEL_FINAL)):
            ws = workloads.config3(96, N=N, seed=22 + N, model=model)
            z, fl, it, _ = solver.solve_batch_host(ws)
            zo2, flo2, io2 = OL.solve_batch(ws)
            ito2 = np.array([i.it for i in io2])
            assert np.array_equal(fl, flo2) and (it == ito2).mean() > 0.97, (N, model)
            same = (fl == 1) & (it == ito2)
            assert np.max(np.abs(z[same] - zo2[same])) < 1e-6, (N, model)
    finally:
        lib.frp_nmpc_set_q4_min_batch(old)


def test_two_host_batches_in_flight_give_the_plans_of_the_blocking_call():
    """frp_nmpc_solve_batch_host_begin / _wait: two sets of registered buffers alternate, the gather of one batch runs
    under the solve of the other (the pipelined solves leave resident slots free for it) -- plans, flags, counts and diagnostics are those
    of the blocking call bit for bit, for DIFFERENT problems in the two sets; a third begin while two are out is refused, as is a batch with
    an unregistered array, and a ticket can be waited for once."""
    B = 3000
    ws = [workloads.config2(B, seed=31), workloads.config2(B, seed=32)]
    ws = [{k: (np.ascontiguousarray(v, dtype=(np.int32 if k == "nfaces" else np.float64)) if isinstance(v, np.ndarray) else v) for k, v in w.items()} for w in ws]
    refs = [solver.solve_batch_host(w) for w in ws]
    outs = [tuple(np.full_like(a, -7) for a in refs[0]) for _ in ws]
    regs = [[w["xinit"], w["x0"], w["params"], w["nfaces"]] + list(o) for w, o in zip(ws, outs)]
    L_ = solver.lib()
    # not registered: refused, nothing enqueued
    t = ctypes.c_int(5)
    with pytest.raises(RuntimeError):
        solver.solve_batch_host_begin(ws[0], outs[0])
    for r in regs:
        solver.host_register(*r)
    try:
        for rnd in range(3):
            tk = [solver.solve_batch_host_begin(ws[q], outs[q]) for q in (0, 1)]
            assert sorted(tk) == [0, 1]
            with pytest.raises(RuntimeError):
                solver.solve_batch_host_begin(ws[0], outs[0])  # FRP_NMPC_HOST_INFLIGHT batches are out
            for q in (1, 0):
                solver.solve_batch_host_wait(tk[q])
            assert L_.frp_nmpc_solve_batch_host_wait(tk[0]) != 0  # (waited for already)
            for q in (0, 1):
                for a, b in zip(refs[q], outs[q]):
                    assert np.array_equal(a, b), (rnd, q)
                for o in outs[q]:
                    o[...] = -7
    finally:
        for r in regs:
            solver.host_unregister(*r)


def test_host_registrations_are_counted_and_sub_ranges_are_aliases():
    """frp_nmpc_host_register: the same buffer registered twice is unpinned by its second unregistration; a sub-range of a
    registered buffer is an alias that is unregistered by its own pointer and keeps the enclosing range alive; a range that partly overlaps
    a registered one is refused; frp_nmpc_host_unregister_all drops everything; the Python wrapper holds a reference to every registered
    array, so a registered array cannot be garbage-collected under the registry."""
    import gc, weakref
    L_ = solver.lib()
    a = np.zeros(1 << 16)
    p, nb = a.ctypes.data, a.nbytes
    assert L_.frp_nmpc_host_registered(p, nb) == 0
    solver.host_register(a); solver.host_register(a)
    assert L_.frp_nmpc_host_registered(p, nb) == 1
    solver.host_unregister(a)
    assert L_.frp_nmpc_host_registered(p, nb) == 1          # one registration left
    sub = a[1024:2048]
    solver.host_register(sub)                               # alias: nothing pinned twice
    assert L_.frp_nmpc_host_unregister(p) != 0              # the enclosing range has an alias: refused
    assert L_.frp_nmpc_host_registered(p, nb) == 1
    solver.host_unregister(sub)
    solver.host_unregister(a)
    assert L_.frp_nmpc_host_registered(p, nb) == 0 and L_.frp_nmpc_host_unregister(p) != 0
    # partial overlap: refused by name
    solver.host_register(a[:4096])
    assert L_.frp_nmpc_host_register(a[2048:].ctypes.data, a[2048:].nbytes) != 0
    solver.host_unregister(a[:4096])
    # the wrapper keeps registered arrays alive
    b = np.zeros(1 << 14); wr = weakref.ref(b); pb, nbb = b.ctypes.data, b.nbytes
    solver.host_register(b); del b; gc.collect()
    assert wr() is not None and L_.frp_nmpc_host_registered(pb, nbb) == 1
    solver.host_unregister_all(); gc.collect()
    assert wr() is None and L_.frp_nmpc_host_registered(pb, nbb) == 0 and not solver._registered


def test_queue_order_hint_changes_the_order_and_nothing_else():
    """frp_nmpc_batch.order_hint (a receding-horizon caller's previous iteration counts): any hint -- the previous counts,
    garbage, negative, all equal -- gives bit-identical plans, flags and iteration counts; the hint may alias `iters`."""
    import torch
    B = 3000  # more problems than resident workgroups: the queue is ordered
    w = workloads.config2(B, seed=78)
    ds = solver.DeviceSolver(B, w["N"], w["M"], 6, w["model"])
    ds.upload(w)
    ds.solve(); torch.cuda.synchronize()
    z0, f0, i0 = ds.z.cpu().numpy().copy(), ds.exitflag.cpu().numpy().copy(), ds.iters.cpu().numpy().copy()
    ds.order_by_last_iters = True   # hint = ds.iters, the buffer the solve writes its counts to
    rng = np.random.default_rng(5)
    for hint in (i0, rng.integers(-(2 ** 31), 2 ** 31 - 1, B), np.zeros(B), np.full(B, 7)):
        ds.iters.copy_(torch.from_numpy(np.asarray(hint, dtype=np.int32)))
        ds.solve(); torch.cuda.synchronize()
        assert np.array_equal(ds.exitflag.cpu().numpy(), f0) and np.array_equal(ds.iters.cpu().numpy(), i0)
        assert np.max(np.abs(ds.z.cpu().numpy() - z0)) == 0.0


@pytest.mark.skipif(not OL.ref_model_available(), reason="oracle/_ref not built (reference tree absent)")
def test_gpu_plans_satisfy_reference_kkt_measured_with_reference_callbacks():
    """The HIP path's plans are KKT points of the REFERENCE NLP as measured with the reference's own CasADi
    callbacks (oracle/_ref travels to the GPU box as a built .so; nothing under /root/reference is read).
    No oracle arithmetic is involved: OL.reference_kkt recovers multipliers by bounded least squares."""
    for w in (workloads.config0(), workloads.config2(6), workloads.config3(3)):
        for tol, stat_max, eq_max in ((1e-4, 5e-3, 1e-4), (1e-8, 1e-6, 1e-8)):
            opt = solver.default_options()
            opt.tol_stat = opt.tol_eq = opt.tol_ineq = opt.tol_comp = tol
            z, fl, _, _ = solver.solve_batch_host(w, opt)
            assert np.all(fl == 1)
            for b in range(z.shape[0]):
                k = OL.reference_kkt(z[b], w["xinit"][b], w["params"][b], w["nfaces"][b], w["N"], w["M"], w["model"])
                assert k["stat"] < stat_max and k["eq"] < eq_max and k["ineq"] < eq_max and k["bound"] < eq_max, (tol, b, k)


# ---- tube propagation (oracle/tube_oracle.py is the numpy/scipy restatement; parity with the
# reference itself is unpinned -- Eigen is absent -- so the tolerance is the agreement of two FP64 methods) ----
def _tube_oracle():
    import sys
    sys.path.insert(0, OL.ROOT)
    from oracle import tube_oracle
    return tube_oracle


def _tube_plans(B, N, seed=5):
    """Plans with the reference's state ranges: solved horizons plus large attitude / speed / thrust excursions."""
    rng = np.random.default_rng(seed)
    lb, ub = L.bounds()
    z = lb + (ub - lb) * rng.random((B, N, 17))
    z[..., 8:11] = rng.uniform(-20, 20, (B, N, 3))
    z[..., 11:14] = rng.uniform(-6, 6, (B, N, 3))
    z[..., 16] = rng.uniform(-np.pi, np.pi, (B, N))
    return z


@pytest.mark.parametrize("N", [20, 1, 21, 22, 43, 64])
def test_tube_matches_oracle(N):
    T = _tube_oracle()
    B = 6 if N <= 22 else 2
    z = _tube_plans(B, N, seed=N)
    z[0] = workloads.config2(1)["x0"][0][np.arange(N) % 20]  # a real warm-start plan
    E = solver.tube_batch_host(z)
    Eo = T.tube_batch(z)
    assert np.max(np.abs(E - Eo) / (1e-3 + np.abs(Eo))) < 1e-9
    assert np.max(np.abs(E - np.swapaxes(E, -1, -2))) < 1e-14  # symmetric principal root


def test_tube_nondefault_constants_and_properties():
    T = _tube_oracle()
    c = dict(mass=1.3, drag=0.1, ego_r=0.4, ego_h=0.15, noise=(0.2, 0.7, 1.1), epsilon=0.1, Ts=0.08)
    z = _tube_plans(4, 20, seed=99)
    E = solver.tube_batch_host(z, c)
    assert np.max(np.abs(E - T.tube_batch(z, c)) / (1e-3 + np.abs(T.tube_batch(z, c)))) < 1e-9
    # properties at full batch size: positive definite, and the tube grows along the horizon (Minkowski sums)
    zb = _tube_plans(4096, 20, seed=3)
    Eb = solver.tube_batch_host(zb)
    ev = np.linalg.eigvalsh(Eb)
    assert np.isfinite(Eb).all() and ev.min() > 0
    tr = np.trace(Eb @ Eb, axis1=-2, axis2=-1)
    assert np.all(np.diff(tr, axis=1)[:, 1:] > 0)
    # stage 0 is the bare ego ellipsoid: E0^2 = R diag(r^2, r^2, h^2) R' has the ego's eigenvalues
    assert np.max(np.abs(np.linalg.eigvalsh(Eb[:, 0]) - np.array([0.0425, 0.27, 0.27]))) < 1e-12


def test_tube_feeds_the_packer_on_device():
    """plan -> tube -> pack on the device equals the adapter fed with the oracle's E."""
    import torch
    T = _tube_oracle()
    w = workloads.config2(8)
    B, N, M = 8, w["N"], w["M"]
    fleet = solver.DeviceFleet(B, N, M, M, w["model"], (15.0, 3.0, 80.0, 15.0, 0.0))
    plan = np.concatenate([w["x0"], w["x0"][:, -1:]], axis=1)
    fleet.mpc_output.copy_(fleet.to_device(plan))
    fleet.tube()
    torch.cuda.synchronize()
    Eo = T.tube_batch(plan[:, :N])
    assert np.max(np.abs(fleet.ellipsoid.cpu().numpy() - Eo)) < 1e-10


def test_receding_horizon_with_tube_propagation_on_device():
    """configs[4] in miniature, the reference's full tick: tube from the current plan (f-2) -> pack (f-1) -> solve ->
    update, all on the device, against the host loop that takes its tube from the numpy/scipy oracle."""
    from forces_resilient_planner_amd import receding
    T = _tube_oracle()
    w0 = workloads.config4_nominal(B=8, ticks=4)
    fg, ig, mg = receding.run(w0, 4, lambda w: solver.solve_batch_host(w)[:3], tube_fn=T.tube_batch)
    fd, idv, md, secs = receding.run_device(w0, 4, propagate_tube=True)
    assert np.array_equal(fd, fg) and (fd == 1).all()
    assert (idv == ig).mean() >= 0.95
    assert np.max(np.abs(md - mg)) < 1e-6


# ---- corridor generation / selection (oracle/corridor_oracle.py restates DecompROS in numpy;
# parity with the reference itself is unpinned -- Eigen is absent) ----
def _corridor_oracle():
    import sys
    sys.path.insert(0, OL.ROOT)
    from oracle import corridor_oracle
    return corridor_oracle


def _corridor_world(seed, P=6000, B=4, N=20, tunnel=0.6, grid=None, gpu_tube=False):
    """A cluttered box with a free tunnel along a gently curved path; per-planner references jittered inside it."""
    rng = np.random.default_rng(seed)
    cloud = np.c_[rng.uniform(-3, 9, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
    if grid:
        cloud = np.round(cloud / grid) * grid  # voxel-map-like cloud: many coplanar / equidistant points
    s = np.linspace(0, 5, N)
    centre = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
    cx = np.interp(cloud[:, 0], centre[:, 0], centre[:, 1]); cz = np.interp(cloud[:, 0], centre[:, 0], centre[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > tunnel]
    ref = centre[None] + rng.normal(0, 0.03, (B, N, 3))
    yaw = np.arctan2(np.gradient(centre[:, 1]), np.gradient(centre[:, 0]))[None] + rng.normal(0, 0.05, (B, N))
    T = _tube_oracle()
    z = np.zeros((B, N, 17)); z[..., 3] = 7.3; z[..., 8:11] = ref; z[..., 16] = yaw
    z[..., 11:14] = rng.normal(0, 0.5, (B, N, 3)); z[..., 14:16] = rng.normal(0, 0.1, (B, N, 2))
    E = solver.tube_batch_host(z) if gpu_tube else T.tube_batch(z)
    return cloud, ref, yaw, E


def _check_corridor(cloud, ref, yaw, E, consts=None, F=64, ordered=True):
    """ordered=False: rows are matched as a set.  After find_ellipsoid has shrunk the seed ellipsoid onto obstacle
    points, those points all sit at distance 1 +- 1 ulp, and which of them find_polyhedron visits first is decided
    by rounding (in the reference as much as here); the set of hyperplanes is the same."""
    C = _corridor_oracle()
    pi, A, b, nf, cnt = solver.corridor_batch_host(cloud, ref, yaw, E, F=F, consts=consts)
    if len(cloud):  # the same launch reading the cloud through a uniform grid: same bits (minima are tie-broken by cloud index)
        for cell in (0.5, 0.23):
            g = solver.corridor_batch_host(cloud, ref, yaw, E, F=F, consts=consts, grid_cell=cell)
            assert np.array_equal(g[0], pi) and np.array_equal(g[3], nf) and np.array_equal(g[4], cnt), cell
            assert np.array_equal(g[1], A) and np.array_equal(g[2], b), cell
    kw = consts or {}
    for p in range(ref.shape[0]):
        idx, polys = C.corridor_one(ref[p], yaw[p], E[p], cloud, **kw)
        assert np.array_equal(pi[p], idx), (p, pi[p], idx)
        assert cnt[p] == len(polys)
        for k, (Ao, bo) in enumerate(polys):
            assert nf[p, k] == len(bo), (p, k, nf[p, k], len(bo))
            G = np.c_[A[p, k, :len(bo)], b[p, k, :len(bo)]]; O = np.c_[Ao, bo]
            if not ordered:
                D = np.abs(G[:, None, :] - O[None, :, :]).max(axis=2)
                match = D.argmin(axis=0)
                assert sorted(match) == list(range(len(bo))), (p, k, match)
                G = G[match]
            assert np.max(np.abs(G - O)) < 1e-9, (p, k)
        assert np.all(nf[p, len(polys):] == 0)
    return pi, A, b, nf


@pytest.mark.parametrize("seed,P,grid", [(1, 6000, None), (2, 20000, None), (3, 6000, 0.1), (4, 300, None)])
def test_corridor_matches_oracle(seed, P, grid):
    cloud, ref, yaw, E = _corridor_world(seed, P=P, grid=grid)
    pi, A, b, nf = _check_corridor(cloud, ref, yaw, E)
    assert pi.max() >= 1  # the path leaves the first polytope: reuse AND re-decomposition are exercised
    # property: no cloud point lies strictly inside any polytope, every reference point lies inside its own
    for p in range(ref.shape[0]):
        for k in range(pi[p].max() + 1):
            m = nf[p, k]
            assert not np.any(np.all(cloud @ A[p, k, :m].T - b[p, k, :m] < -1e-9, axis=1))
        for i in range(ref.shape[1]):
            m = nf[p, pi[p, i]]
            assert np.all(A[p, pi[p, i], :m] @ ref[p, i] - b[p, pi[p, i], :m] <= 0)


def test_corridor_obstacles_inside_the_seed_ellipsoid_and_edge_cases():
    """Exercises both shrink loops of find_ellipsoid (line_segment.h:156-208) -- only reached when an obstacle lies
    inside the seed ellipsoid, i.e. with a long seed segment -- plus the empty cloud and a one-stage horizon."""
    cloud, ref, yaw, E = _corridor_world(7, P=8000, B=3, tunnel=0.25)
    _check_corridor(cloud, ref, yaw, E, consts=dict(seed_len=1.5, bbox=(2.0, 2.0, 1.0), inflation=1.1), ordered=False)
    _check_corridor(cloud, ref, yaw, E, consts=dict(seed_len=0.8, bbox=(1.0, 1.5, 0.7), inflation=1.0), ordered=False)
    # empty cloud: the polytope is the local box, reused for as long as the tube fits
    pi, A, b, nf = _check_corridor(np.zeros((0, 3)), ref, yaw, E)
    assert np.all(nf[:, 0] == 6)
    _check_corridor(cloud, ref[:, :1], yaw[:, :1], E[:, :1])


def test_corridor_dense_clouds_boxes_beyond_the_register_tile():
    """Local boxes with more points than the one-wavefront kernel's register tile (1280) go to its shell form (frp_corridor.hip,
    corridor_wave_kernel<true>: the in-box points taken in shells of their distance in the final ellipsoid, filtered by the cuts made so
    far).  Same polytopes, to the bit, as the workgroup kernels on the plain cloud (`_check_corridor` compares the grid launches with the
    plain one) and the oracle's rows: ~1700 and ~5300 points per box, obstacles inside a long seed ellipsoid (both shrink loops on the
    listed subset), a voxel-like cloud (ties), and a sphere of 3000 equidistant points that no shell can split (the planner is handed
    back to the workgroup kernels after the retries)."""
    cloud, ref, yaw, E = _corridor_world(21, P=20000, B=3)
    _check_corridor(cloud, ref, yaw, E)
    cloud, ref, yaw, E = _corridor_world(22, P=62000, B=3)
    pi, A, b, nf = _check_corridor(cloud, ref, yaw, E)
    assert pi.max() >= 1
    cloud, ref, yaw, E = _corridor_world(23, P=40000, B=2, tunnel=0.25)
    _check_corridor(cloud, ref, yaw, E, consts=dict(seed_len=1.5, bbox=(2.0, 2.0, 1.0), inflation=1.1), ordered=False)
    cloud, ref, yaw, E = _corridor_world(24, P=50000, B=2, grid=0.12)
    _check_corridor(cloud, ref, yaw, E)
    cloud, ref, yaw, E = _corridor_world(25, P=6000, B=2)
    rng = np.random.default_rng(25)
    v = rng.normal(size=(3000, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    sphere = ref[0, 0] + np.array([0.05, 0.0, 0.0]) + 0.9 * v        # (the seed segment's midpoint is 0.05 ahead of the first reference)
    dense = np.r_[cloud, sphere]
    plain = solver.corridor_batch_host(dense, ref, yaw, E)              # (3000-fold near-ties: which one an implementation visits first is
    for cell in (0.5, 0.23):                                            #  rounding, so this one is checked kernel against kernel only)
        g = solver.corridor_batch_host(dense, ref, yaw, E, grid_cell=cell)
        assert all(np.array_equal(x, y) for x, y in zip(g, plain)), cell


def test_corridor_grid_smaller_than_the_cloud_and_other_edges():
    """frp_nmpc_cloud_grid_build bins the points outside the grid into its border cells, so a border cell's points reach as far as the
    cloud does -- the row clipping of the one-wavefront kernel must treat those cells as unbounded on their outer side.  A grid that covers
    a fraction of the cloud in every direction (the path leaves it), 0.5 and 0.3 m cells: same polytopes as the plain-cloud launch."""
    import torch
    cloud, ref, yaw, E = _corridor_world(31, P=30000, B=3)
    plain = solver.corridor_batch_host(cloud, ref, yaw, E)
    dev = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dtype=dt)
    B, N, F = ref.shape[0], ref.shape[1], solver.CORRIDOR_MAX_F
    for cell, origin, dims in ((0.5, (0.5, -1.5, 0.2), (8, 6, 3)), (0.3, (-1.0, -0.9, 0.5), (20, 6, 4)), (0.5, (2.0, -3.0, -0.4), (3, 12, 7))):
        d_cloud = dev(cloud)
        grid = solver.CloudGrid(d_cloud, cell, origin=origin, dims=dims)
        A = torch.zeros((B, N, F, 3), dtype=torch.float64, device="cuda:0"); b = torch.zeros((B, N, F), dtype=torch.float64, device="cuda:0")
        nf = torch.zeros((B, N), dtype=torch.int32, device="cuda:0"); pi = torch.zeros((B, N), dtype=torch.int32, device="cuda:0")
        cnt = torch.zeros((B,), dtype=torch.int32, device="cuda:0")
        solver.corridor_batch_device(d_cloud, dev(ref), dev(yaw), dev(E), A, b, nf, pi, cnt, grid=grid)
        torch.cuda.synchronize()
        got = (pi.cpu().numpy(), A.cpu().numpy(), b.cpu().numpy(), nf.cpu().numpy(), cnt.cpu().numpy())
        assert all(np.array_equal(x, y) for x, y in zip(got, plain)), (cell, origin, dims)
    # fewer rows stored than a polytope has (F = 8 of ~25): the truncation, its flag (a negative polytope count) and the containment checks on
    # the stored rows are the same in the one-wavefront kernel (rows made after the loop, lane = cut) as in the workgroup kernels
    # the longest horizon (one lane per stage in the containment check), a horizon that gives a stage two lanes, and an empty cloud behind a grid
    for Nh in (64, 30):
        c2, r2, y2, E2 = _corridor_world(32 + Nh, P=9000, B=2, N=Nh)
        p2 = solver.corridor_batch_host(c2, r2, y2, E2)
        g2 = solver.corridor_batch_host(c2, r2, y2, E2, grid_cell=0.5)
        assert all(np.array_equal(x, y) for x, y in zip(g2, p2)), Nh
    e0 = solver.corridor_batch_host(np.zeros((0, 3)), ref, yaw, E)
    e1 = solver.corridor_batch_host(np.zeros((0, 3)), ref, yaw, E, grid_cell=0.5)
    assert all(np.array_equal(x, y) for x, y in zip(e1, e0)) and np.all(e0[3][:, 0] == 6)
    few = solver.corridor_batch_host(cloud, ref, yaw, E, F=8)
    assert (few[4] < 0).all() and few[3].max() > 8
    for cell in (0.5, 0.23):
        g = solver.corridor_batch_host(cloud, ref, yaw, E, F=8, grid_cell=cell)
        assert all(np.array_equal(x, y) for x, y in zip(g, few)), cell


# ---- SURVEY 8f row f-4 (first half): stage references from the kinodynamic path ----
def _reference_oracle():
    import sys
    sys.path.insert(0, OL.ROOT)
    from oracle import reference_oracle
    return reference_oracle


def _kino_world(seed, B, N, K=80):
    rng = np.random.default_rng(seed)
    t = np.arange(K) * 0.05
    path = np.c_[1.5 * t, np.sin(1.3 * t), 1.0 + 0.2 * np.cos(t)]
    path[K // 2:K // 2 + 12] = path[K // 2]              # a hover segment: direction shorter than 0.1 -> yaw held
    path[-15:, :2] = path[-16, :2] - np.c_[t[:15], 2 * t[:15]]  # a sharp turn back: exercises the +-pi unwrap
    off = rng.uniform(-0.02, 0.05 * K, B)                 # includes starts before the path and runs past its end
    off[0] = 0.0; off[1] = 0.05 * 3; off[2] = -0.01
    plans = np.zeros((B, N + 1, 17))
    plans[:, 1, 8:11] = path[np.clip((off / 0.05).astype(int), 0, K - 1)] + rng.normal(0, 0.6, (B, 3))
    plans[:, 1, 16] = rng.uniform(-3.1, 3.1, B)
    return path, off, plans


@pytest.mark.parametrize("N", [20, 1, 64])
def test_reference_sampling_matches_oracle(N):
    R = _reference_oracle()
    B = 64
    path, off, plans = _kino_world(N, B, N)
    size = np.array([len(path) - 7], dtype=np.int32)
    for ks, n in ((None, len(path)), (size, int(size[0]))):
        rp, ry, fl = solver.reference_batch_host(path, off, plans, kino_size=ks)
        for p in range(B):
            po, yo, fo = R.references_one(path, n, off[p], plans[p], N)
            assert np.max(np.abs(rp[p] - po)) < 1e-12 and np.max(np.abs(ry[p] - yo)) < 1e-12 and bool(fl[p]) == fo, p
    assert fl.any() and not fl.all()
    # per-planner paths
    paths = path[None] + np.arange(B)[:, None, None] * 0.01
    rp, ry, fl = solver.reference_batch_host(paths, off, plans)
    for p in (0, 5, B - 1):
        po, yo, fo = R.references_one(paths[p], len(path), off[p], plans[p], N)
        assert np.max(np.abs(rp[p] - po)) < 1e-12 and np.max(np.abs(ry[p] - yo)) < 1e-12


def test_full_tick_on_device_matches_the_chain_of_oracles():
    """SURVEY 8f rows f-4 -> f-2 -> f-3 -> f-1 -> solver -> bookkeeping in one on-device tick (DeviceFleet.full_tick)
    against the same chain on the host: reference / tube / corridor oracles, the numpy adapter, and the NLP solved
    through the host C-ABI entry.  Two ticks, so that the second starts from plans the first one wrote."""
    import torch
    from forces_resilient_planner_amd.adapter import ForcesAdapter, update_forces_results
    R, T, C = _reference_oracle(), _tube_oracle(), _corridor_oracle()
    B, N, M, F = 6, 20, 30, 64
    rng = np.random.default_rng(21)
    K = 60
    s = np.arange(K) * 0.05 * 1.6
    path = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
    P = 5000
    cloud = np.c_[rng.uniform(-3, 9, P), rng.uniform(-4, 4, P), rng.uniform(-0.5, 3, P)]
    cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.9]
    f_ext = rng.uniform(-1.5, 1.5, (B, 3))
    plan = np.zeros((B, N + 1, 17)); plan[..., 3] = 7.3; plan[..., 7] = 7.3
    plan[..., 8:11] = path[0] + rng.normal(0, 0.02, (B, 1, 3)); plan[..., 16] = 0.2
    weights = (15.0, 3.0, 80.0, 15.0, 0.0)
    fleet = solver.DeviceFleet(B, N, M, F, L.MODEL_NORMAL, weights)
    fleet.mpc_output.copy_(fleet.to_device(plan))
    d_path, d_cloud, d_f = fleet.to_device(path), fleet.to_device(cloud), fleet.to_device(f_ext)
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device="cuda:0"); ry = torch.zeros((B, N), dtype=torch.float64, device="cuda:0")
    ad = ForcesAdapter(B, L.MODEL_NORMAL, N, M)
    for tick in range(2):
        off = np.full(B, 0.05 * tick) + np.arange(B) * 0.013
        fleet.full_tick(d_f, d_path, fleet.to_device(off), d_cloud, rp, ry)
        torch.cuda.synchronize()
        # host chain
        ref_pos = np.zeros((B, N, 3)); ref_yaw = np.zeros((B, N)); E = np.zeros((B, N, 3, 3))
        A = np.zeros((B, N, F, 3)); bb = np.zeros((B, N, F)); nf = np.zeros((B, N), np.int32)
        for p in range(B):
            ref_pos[p], ref_yaw[p], _ = R.references_one(path, K, off[p], plan[p], N)
            E[p] = T.tube_one(plan[p, :N])
            idx, polys = C.corridor_one(ref_pos[p], ref_yaw[p], E[p], cloud)
            assert np.array_equal(fleet.poly_index[p].cpu().numpy(), idx)
            for i in range(N):
                Ai, bi = polys[idx[i]]
                m = min(len(bi), F)
                A[p, i, :m] = Ai[:m]; bb[p, i, :m] = bi[:m]; nf[p, i] = len(bi)
        assert np.max(np.abs(rp.cpu().numpy() - ref_pos)) < 1e-12 and np.max(np.abs(fleet.ellipsoid.cpu().numpy() - E)) < 1e-10
        xinit, x0, params, nfa = ad.pack(plan, f_ext, ref_pos, ref_yaw, E, A, bb, nf)
        params = params.copy(); params[:, :, 6:9] = fleet.solver.params[:, :, 6:9].cpu().numpy()  # weights: set by the device packer
        assert np.max(np.abs(fleet.solver.params.cpu().numpy() - params)) < 1e-9
        w = dict(xinit=xinit.copy(), x0=x0.copy(), params=params, nfaces=nfa.copy(), N=N, M=M, model=L.MODEL_NORMAL, B=B)
        z, fl, it, _ = solver.solve_batch_host(w)
        assert np.array_equal(fl, fleet.solver.exitflag.cpu().numpy()) and (fl == 1).all()
        assert np.max(np.abs(z - fleet.solver.z.cpu().numpy())) < 1e-6
        ad.output[:] = z
        upd = plan.copy(); ad.update(upd); plan = update_forces_results(upd)
        assert np.max(np.abs(fleet.mpc_output.cpu().numpy() - plan)) < 1e-6
        plan = fleet.mpc_output.cpu().numpy().copy()


def test_corridor_terminates_on_non_finite_and_degenerate_input():
    """NaN / inf coordinates, an obstacle exactly at the seed centre and a NaN reference must not hang the kernel
    (the reference's while-loops would spin on such input); finite planners in the same launch are unaffected."""
    cloud, ref, yaw, E = _corridor_world(11, P=500, B=4)
    good = solver.corridor_batch_host(cloud, ref, yaw, E)
    bad_cloud = cloud.copy()
    bad_cloud[3] = np.nan; bad_cloud[7] = np.inf
    bad_cloud[11] = ref[0, 0] + np.array([0.05 * np.cos(yaw[0, 0]), 0.05 * np.sin(yaw[0, 0]), 0.0])  # the seed centre itself
    ref2 = ref.copy(); ref2[1, 5] = np.nan
    pi, A, b, nf, cnt = solver.corridor_batch_host(bad_cloud, ref2, yaw, E)
    assert pi.shape == good[0].shape and np.all(np.abs(cnt) >= 1)
    # planners 2 and 3 never see the poisoned points' effects differently from the oracle's rules: still valid polytopes
    for p in (2, 3):
        for k in range(abs(int(cnt[p]))):
            assert nf[p, k] >= 6


def test_corridor_full_size_properties():
    """B = 4096 planners on a 20 k-point cloud (the fleet size of BASELINE configs[2]/[4]): size-independent
    properties of every polytope, checked on the device with torch -- no obstacle strictly inside, unit normals,
    every stage's reference inside the polytope it was assigned, indices non-decreasing by at most one."""
    import torch
    cloud, ref, yaw, E = _corridor_world(5, P=20000, B=4096, gpu_tube=True)
    dev = "cuda:0"
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    B, N, F = ref.shape[0], ref.shape[1], 64
    A = torch.zeros((B, N, F, 3), dtype=torch.float64, device=dev); b = torch.zeros((B, N, F), dtype=torch.float64, device=dev)
    nf = torch.zeros((B, N), dtype=torch.int32, device=dev); pi = torch.zeros((B, N), dtype=torch.int32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_cloud, d_ref = t(cloud), t(ref)
    solver.corridor_batch_device(d_cloud, d_ref, t(yaw), t(E), A, b, nf, pi, cnt)
    torch.cuda.synchronize()
    assert int((cnt <= 0).sum()) == 0 and int(nf.max()) <= F
    live = torch.arange(F, device=dev)[None, None, :] < nf[:, :, None]
    assert float(((A.norm(dim=-1) - 1).abs() * live).max()) < 1e-12
    d = torch.diff(pi, dim=1)
    assert int(pi[:, 0].abs().max()) == 0 and int(d.min()) >= 0 and int(d.max()) <= 1 and torch.equal(pi[:, -1] + 1, cnt)
    # reference of stage i inside polytope pi[i]
    Ai = torch.gather(A, 1, pi.long()[:, :, None, None].expand(-1, -1, F, 3)); bi = torch.gather(b, 1, pi.long()[:, :, None].expand(-1, -1, F))
    li = torch.gather(live, 1, pi.long()[:, :, None].expand(-1, -1, F))
    viol = (torch.einsum("bnfk,bnk->bnf", Ai, d_ref) - bi > 0) & li
    assert int(viol.sum()) == 0
    # no cloud point strictly inside any polytope (a sample of planners, all their polytopes)
    for p0 in range(0, B, 256):
        Ap, bp, lp = A[p0], b[p0], live[p0]                                  # [N,F,3], [N,F], [N,F]
        sd = torch.einsum("nfk,pk->npf", Ap, d_cloud) - bp[:, None, :]       # [N,P,F]
        inside = ((sd < -1e-9) | ~lp[:, None, :]).all(dim=-1) & (nf[p0] > 0)[:, None]
        assert int(inside.sum()) == 0, p0


@pytest.mark.parametrize("P,mode", [(9000, "list"), (40000, "cloud")])
def test_corridor_dense_boxes_use_the_list_and_cloud_paths(P, mode):
    """The kernel keeps a decomposition's in-box points in registers (<= 2048), in an LDS index list (<= 8192) or, beyond
    that, addresses the cloud directly; the other corridor tests stay in the first regime.  A dense cloud packed into a
    region little larger than one local box forces the other two, against the same oracle."""
    rng = np.random.default_rng(P)
    cloud = np.c_[rng.uniform(-1.5, 4.5, P), rng.uniform(-2.2, 2.2, P), rng.uniform(0.0, 2.0, P)]
    N, B = 6, 2
    s = np.linspace(0.5, 2.5, N)
    centre = np.c_[s, 0.2 * np.sin(s), np.full(N, 1.0)]
    cx = np.interp(cloud[:, 0], centre[:, 0], centre[:, 1])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - 1.0) > 0.45]
    ref = centre[None] + rng.normal(0, 0.02, (B, N, 3))
    yaw = rng.normal(0.1, 0.05, (B, N))
    E = np.tile(np.diag([0.2, 0.2, 0.05]), (B, N, 1, 1))
    C = _corridor_oracle()
    box = C.local_bbox_planes(ref[0, 0], ref[0, 0] + 0.1 * np.array([np.cos(yaw[0, 0]), np.sin(yaw[0, 0]), 0]), np.array([2.0, 2.0, 1.0]))
    inbox = np.ones(len(cloud), bool)
    for pp, n in box:
        inbox &= (cloud - pp) @ n <= 1e-10
    assert (2048 < inbox.sum() <= 8192) if mode == "list" else inbox.sum() > 8192
    _check_corridor(cloud, ref, yaw, E)


def test_full_tick_replayed_from_a_hipgraph_equals_eager_launches():
    """Every entry point is asynchronous on the caller's stream and allocates nothing, so a tick can be captured into a
    hipGraph (the launch-bound case: small fleets).  Outputs are poisoned before each run: a solve that silently does
    nothing -- what a captured hipMemsetAsync of the queue counter produced on ROCm 7.2 -- cannot pass."""
    import torch
    B, N, M, F, K = 16, 20, 30, 64, 200
    rng = np.random.default_rng(8)
    s = np.arange(K) * 0.05 * 0.4
    path = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
    cloud = np.c_[rng.uniform(-3, 12, 3000), rng.uniform(-4, 4, 3000), rng.uniform(-0.5, 3, 3000)]
    cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.9]
    plan = np.zeros((B, N + 1, 17)); plan[..., 3] = 7.3; plan[..., 7] = 7.3
    plan[..., 8:11] = path[0] + rng.normal(0, 0.02, (B, 1, 3)); plan[..., 16] = 0.2
    fleet = solver.DeviceFleet(B, N, M, F, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0))
    fleet.poly_index = torch.zeros((B, N), dtype=torch.int32, device="cuda:0")
    d_path, d_cloud, d_f = fleet.to_device(path), fleet.to_device(cloud), fleet.to_device(rng.normal(0, 0.5, (B, 3)))
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device="cuda:0"); ry = torch.zeros((B, N), dtype=torch.float64, device="cuda:0")
    toff = torch.zeros((B,), dtype=torch.float64, device="cuda:0")

    def run(ticks, fn):
        fleet.mpc_output.copy_(fleet.to_device(plan)); toff.zero_()
        fleet.solver.z.fill_(float("nan")); fleet.solver.exitflag.fill_(-99)
        out = []
        for _ in range(ticks):
            fn(); toff.add_(0.05)
            out.append(fleet.mpc_output.clone())
        torch.cuda.synchronize()
        return torch.stack(out), fleet.solver.exitflag.clone()

    eager = lambda: fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry)
    ref_plans, ref_flags = run(5, eager)
    assert bool((ref_flags == 1).all()) and bool(torch.isfinite(ref_plans).all())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry, stream=side)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fleet.full_tick(d_f, d_path, toff, d_cloud, rp, ry, stream=torch.cuda.current_stream())
    for _ in range(2):  # the first replay after capture included
        plans, flags = run(5, g.replay)
        assert torch.equal(plans, ref_plans) and torch.equal(flags, ref_flags)


def test_coldstart_on_device_matches_init_mpc_output():
    import torch
    from forces_resilient_planner_amd.adapter import init_mpc_output
    B, N = 37, 20
    rng = np.random.default_rng(4)
    fleet = solver.DeviceFleet(B, N, 30, 6, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0))
    plan = rng.normal(size=(B, N + 1, 17))
    state = rng.normal(size=(B, 9))
    flags = rng.choice([1, 0, -7, -6], size=B).astype(np.int32)
    fleet.solver.exitflag.copy_(torch.from_numpy(flags).to("cuda:0"))
    for st in (state, None):
        fleet.mpc_output.copy_(fleet.to_device(plan))
        fleet.coldstart(fleet.to_device(st) if st is not None else None)
        torch.cuda.synchronize()
        want = plan.copy()
        bad = flags != 1
        want[bad] = init_mpc_output(st[bad] if st is not None else plan[bad][:, 1, 8:17], N)
        assert np.array_equal(fleet.mpc_output.cpu().numpy(), want)
    fleet.mpc_output.copy_(fleet.to_device(plan)); fleet.coldstart(fleet.to_device(state), only_failed=False)
    assert np.array_equal(fleet.mpc_output.cpu().numpy(), init_mpc_output(state, N))


def test_plain_c_program_drives_the_whole_tick_through_the_c_abi(tmp_path):
    """tests/cpp/tick_harness.c: C99 + HIP runtime + include/frp_nmpc.h only (no torch, no C++).  Three ticks of
    coldstart -> reference -> tube -> corridor -> pack -> solve -> update; same bits as the Python DeviceFleet."""
    import subprocess
    import torch
    root = OL.ROOT
    exe = os.path.join(root, "tests", "cpp", "tick_harness")
    src = exe + ".c"
    libdir = os.path.dirname(solver.LIB_PATH)
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(solver.LIB_PATH), os.path.getmtime(os.path.join(OL.ROOT, "include", "frp_nmpc.h"))):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", src, "-o", exe,
                               "-L" + libdir, "-lfrp_nmpc_amd", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    B, N, K, T = 12, 20, 80, 3
    rng = np.random.default_rng(31)
    s = np.arange(K) * 0.05 * 1.2
    path = np.c_[s, 0.4 * np.sin(0.8 * s), 1.0 + 0.1 * np.cos(s)]
    cloud = np.c_[rng.uniform(-3, 9, 4000), rng.uniform(-4, 4, 4000), rng.uniform(-0.5, 3, 4000)]
    cx = np.interp(cloud[:, 0], path[:, 0], path[:, 1]); cz = np.interp(cloud[:, 0], path[:, 0], path[:, 2])
    cloud = cloud[np.hypot(cloud[:, 1] - cx, cloud[:, 2] - cz) > 0.9]
    plan = np.zeros((B, N + 1, 17)); plan[..., 3] = 7.3; plan[..., 7] = 7.3
    plan[..., 8:11] = path[0] + rng.normal(0, 0.02, (B, 1, 3)); plan[..., 16] = 0.2
    f_ext = rng.normal(0, 0.5, (B, 3))
    toff = 0.05 * np.arange(T)[:, None] + rng.uniform(0, 0.02, (1, B))
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        np.array([B, N, K, len(cloud), T], dtype=np.int32).tofile(f)
        for a in (path, cloud, plan, f_ext, toff):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    subprocess.check_call([exe, str(inp), str(out)])
    raw = np.fromfile(out, dtype=np.uint8)
    nb = B * (N + 1) * 17 * 8
    plan_c = raw[:nb].view(np.float64).reshape(B, N + 1, 17)
    ints = raw[nb:].view(np.int32)
    flag_c, it_c, pi_c = ints[:B], ints[B:2 * B], ints[2 * B:].reshape(B, N)
    fleet = solver.DeviceFleet(B, N, 30, 64, L.MODEL_NORMAL, (15.0, 3.0, 80.0, 15.0, 0.0))
    fleet.mpc_output.copy_(fleet.to_device(plan)); fleet.solver.exitflag.fill_(1)
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device="cuda:0"); ry = torch.zeros((B, N), dtype=torch.float64, device="cuda:0")
    d_path, d_cloud, d_f = fleet.to_device(path), fleet.to_device(cloud), fleet.to_device(f_ext)
    for t in range(T):
        fleet.full_tick(d_f, d_path, fleet.to_device(toff[t]), d_cloud, rp, ry)
    torch.cuda.synchronize()
    assert np.array_equal(flag_c, fleet.solver.exitflag.cpu().numpy()) and np.all(flag_c == 1)
    assert np.array_equal(it_c, fleet.solver.iters.cpu().numpy())
    assert np.array_equal(pi_c, fleet.poly_index.cpu().numpy())
    assert np.array_equal(plan_c, fleet.mpc_output.cpu().numpy())


def test_cloud_grid_is_a_cell_sorted_permutation_of_the_cloud():
    """frp_nmpc_cloud_grid_build: counting sort of the cloud by cell -- every point exactly once, inside the cell range
    that cell_start gives it, points beyond the grid (and NaNs) in border cells."""
    import torch
    rng = np.random.default_rng(12)
    cloud = np.c_[rng.uniform(-3, 9, 5000), rng.uniform(-4, 4, 5000), rng.uniform(-0.5, 3, 5000)]
    cloud[17] = np.nan; cloud[99] = [1e6, -1e6, 0.3]
    origin, cell, dims = (-2.0, -3.0, 0.0), 0.4, (20, 12, 6)   # deliberately smaller than the cloud
    g = solver.CloudGrid(torch.from_numpy(cloud).to("cuda:0"), cell, origin, dims)
    torch.cuda.synchronize()
    start, idx, pts = g.start.cpu().numpy(), g.index.cpu().numpy(), g.points.cpu().numpy()
    assert start[0] == 0 and start[-1] == len(cloud) and np.all(np.diff(start) >= 0)
    assert np.array_equal(np.sort(idx), np.arange(len(cloud)))
    assert np.array_equal(pts, cloud[idx], equal_nan=True)
    with np.errstate(invalid="ignore"):
        ijk = np.floor((cloud - np.array(origin)) / cell)
    ijk = np.where(ijk > 0, ijk, 0)                         # NaN -> 0 like the kernel
    ijk = np.minimum(ijk, np.array(dims) - 1).astype(int)
    want = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
    cell_of_sorted = np.searchsorted(start, np.arange(len(cloud)), side="right") - 1
    assert np.array_equal(cell_of_sorted, want[idx])


def test_per_problem_model_equals_two_single_model_batches():
    """frp_nmpc_batch.model_per_problem: a fleet whose planners switch to the final solver one by one
    (switch_to_final, nmpc_solver.cpp:381, 446-447) is one launch; every problem gets exactly the solve its own
    model would have given it in a single-model batch (which is what the oracle checks elsewhere)."""
    w = workloads.config2(300)
    rng = np.random.default_rng(2)
    models = rng.integers(0, 2, 300).astype(np.int32)
    zm, fm, im, _ = solver.solve_batch_host(dict(w, models=models))
    z0, f0, i0, _ = solver.solve_batch_host(dict(w, model=L.MODEL_NORMAL))
    z1, f1, i1, _ = solver.solve_batch_host(dict(w, model=L.MODEL_FINAL))
    pick = lambda a0, a1: np.where(models.reshape((-1,) + (1,) * (a0.ndim - 1)) == 1, a1, a0)
    assert np.array_equal(fm, pick(f0, f1)) and np.array_equal(im, pick(i0, i1))
    assert np.array_equal(zm, pick(z0, z1))
    assert np.max(np.abs(z0 - z1)) > 1e-3  # the two models really differ on this batch
    # and through the ordered-queue path (B > resident slots)
    w = workloads.config2(3000)
    models = rng.integers(0, 2, 3000).astype(np.int32)
    zm, fm, _, _ = solver.solve_batch_host(dict(w, models=models))
    z0, _, _, _ = solver.solve_batch_host(dict(w, model=L.MODEL_NORMAL)); z1, _, _, _ = solver.solve_batch_host(dict(w, model=L.MODEL_FINAL))
    assert np.array_equal(zm, np.where(models[:, None, None] == 1, z1, z0))


def test_fleet_with_per_planner_mode_switch():
    """switch_to_final per planner (nmpc_solver.cpp:381, 436-447): the mode rule on the device against its statement,
    the packer's weights by mode against the adapter with setParasNormal / setParasFinal weights, and the solve by
    mode against single-model batches."""
    import torch
    from forces_resilient_planner_amd.adapter import ForcesAdapter
    B, N, M, F = 24, 20, 30, 6
    wn, wf = (7.0, 1.0, 80.0, 12.0, 0.5), (12.0, 1.5, 80.0, 15.0, 0.5)      # rotors_sim.launch:56-66
    w = workloads.config2(B)
    fleet = solver.DeviceFleet(B, N, M, F, L.MODEL_NORMAL, wn, weights_final=wf)
    fleet.mpc_output.copy_(fleet.to_device(w["mpc_output"])); fleet.ellipsoid.copy_(fleet.to_device(w["E"]))
    fleet.poly_A.copy_(fleet.to_device(w["poly_A"])); fleet.poly_b.copy_(fleet.to_device(w["poly_b"]))
    fleet.poly_nfaces.copy_(fleet.to_device(w["nfaces"], dtype=torch.int32))
    rng = np.random.default_rng(6)
    toff = rng.uniform(0, 3.0, B); ksize = rng.integers(30, 90, B).astype(np.int32)
    end_pt = w["mpc_output"][:, N - 1, 8:11] + rng.normal(0, 0.8, (B, 3))
    fleet.update_mode(fleet.to_device(toff), fleet.to_device(ksize, dtype=torch.int32), fleet.to_device(end_pt))
    torch.cuda.synchronize()
    want = np.array([(int((N * 0.05 + toff[b]) / 0.05) >= ksize[b]) or np.linalg.norm(w["mpc_output"][b, N - 1, 8:11] - end_pt[b]) < 1.0
                     for b in range(B)])
    mode = fleet.mode.cpu().numpy()
    assert np.array_equal(mode == L.MODEL_FINAL, want) and want.any() and not want.all()
    fleet.update_mode(fleet.to_device(np.zeros(B)), fleet.to_device(np.array([10 ** 6], np.int32), dtype=torch.int32),
                      fleet.to_device(np.array([1e3, 1e3, 1e3])))
    assert np.array_equal(fleet.mode.cpu().numpy(), mode)                    # sticky
    fleet.pack(fleet.to_device(w["f_ext"]), fleet.to_device(w["ref_pos"]), fleet.to_device(w["ref_yaw"]))
    fleet.solver.solve(); torch.cuda.synchronize()
    params = fleet.solver.params.cpu().numpy()
    res = {}
    for m, wts in ((L.MODEL_NORMAL, wn), (L.MODEL_FINAL, wf)):
        ad = ForcesAdapter(B, m, N, M); ad.set_paras(*wts)
        xinit, x0, par, nf = ad.pack(w["mpc_output"], w["f_ext"], w["ref_pos"], w["ref_yaw"], w["E"], w["poly_A"], w["poly_b"], w["nfaces"])
        sel = mode == m
        assert np.max(np.abs(params[sel] - par[sel])) < 1e-12
        res[m] = solver.solve_batch_host(dict(xinit=xinit.copy(), x0=x0.copy(), params=par.copy(), nfaces=nf.copy(), N=N, M=M, model=m, B=B))
    z = fleet.solver.z.cpu().numpy()
    for m in res:
        sel = mode == m
        assert np.array_equal(fleet.solver.exitflag.cpu().numpy()[sel], res[m][1][sel])
        assert np.max(np.abs(z[sel] - res[m][0][sel])) < 1e-9
    fleet.reset_mode()
    assert int((fleet.mode != L.MODEL_NORMAL).sum()) == 0


def test_default_code_generation_build_passes_the_variant_and_horizon_tests():
    """The product compiles the solver kernel with internal code-generation switches of one compiler release (build.py:
    CODEGEN_FLAGS, +5 % speed).  No result may depend on them: the library built with the compiler's defaults
    (lib_defaultflags.so, made by __graft_entry__.build()) passes the same per-variant and per-horizon parity tests."""
    import subprocess
    import sys
    from forces_resilient_planner_amd import build
    lib = build.DEFAULT_FLAGS_LIB
    assert os.path.exists(lib), "run __graft_entry__.build()"
    env = dict(os.environ, FRP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "every_kernel_variant or horizon_lengths"], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_statically_linked_planner_stub_solves_config0(tmp_path):
    """The reference links libFORCESNLPsolver_normal.a / libFORCESNLPsolver_final.a by file name (plan_manage/CMakeLists.txt:64-65,
    82-83).  build() makes archives with those names and, in the build container, a stub compiled against the reference's own
    headers and linked against them by plain g++; here it solves BASELINE configs[0] with both solvers."""
    import subprocess
    from forces_resilient_planner_amd import build
    stub = os.path.join(os.path.dirname(build.DROPIN_DIR), "planner_stub")
    if not os.path.exists(stub):
        pytest.skip("planner_stub is built where the reference headers are (build container)")
    w0 = workloads.config0()
    p = solver.ForcesParams()
    p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel(); p.num_of_threads = 1
    f = tmp_path / "params.bin"
    f.write_bytes(bytes(p))
    r = subprocess.run([stub, str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    f1, f2, obj_n, obj_f, it = lines[0].split()
    assert f1 == "1" and f2 == "1" and abs(float(obj_n) - 23.1594329641) < 1e-4 and abs(float(obj_f) - 48.4610568794) < 2e-4
    # the info block is honest (VERDICT r03 missing #2): the gap is the complementarity sum the iteration drove below tolerance,
    # dobj = pobj - dgap, and the affine-step quantities are those of the oracle's last iteration
    dobj, dgap, rdgap, mu, mu_aff, sigma, step_aff, step_cc = (float(x) for x in lines[1].split()[:8])
    assert lines[1].split()[8:] == ["0", "0"]                      # no line search: no backtracking steps to report
    mtot = 20 * (34 + 6)
    assert 0.0 < dgap < 1e-4 * mtot and abs(dgap - mu * mtot) <= 1e-12 * mtot
    assert abs(dobj - (float(obj_n) - dgap)) < 1e-9 and abs(rdgap - dgap / float(obj_n)) < 1e-12
    zo, flo, io = OL.solve_batch(w0)
    assert flo[0] == 1 and io[0].it == int(it)
    assert abs(mu_aff - io[0].mu_aff) <= 1e-6 * (1 + abs(io[0].mu_aff)) and abs(sigma - io[0].sigma) <= 1e-6
    assert abs(step_aff - io[0].step_aff) <= 1e-6 and abs(step_cc - io[0].step_cc) <= 1e-6 and 0.0 < step_aff <= 1.0
    # the two transports of the drop-in context -- inputs / outputs in place in pinned mapped memory (default) or copied around the
    # launch -- return the same bits; the latency option (FRP_NMPC_TWIST) the same solution within the tolerances
    env = dict(os.environ)
    env["FRP_NMPC_DROPIN_ZEROCOPY"] = "0"
    r0 = subprocess.run([stub, str(f)], capture_output=True, text=True, env=env)
    assert r0.returncode == 0 and r0.stdout == r.stdout
    env = dict(os.environ)
    env["FRP_NMPC_TWIST"] = "-1"
    rt = subprocess.run([stub, str(f)], capture_output=True, text=True, env=env)
    assert rt.returncode == 0
    t1, t2, tobj_n, tobj_f, tit = rt.stdout.strip().splitlines()[0].split()
    assert t1 == "1" and t2 == "1" and abs(float(tobj_n) - float(obj_n)) < 1e-6 and abs(float(tobj_f) - float(obj_f)) < 1e-6 and tit == it
