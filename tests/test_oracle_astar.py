"""CPU tests of the kinodynamic-A* oracle (oracle/astar_oracle.c): its deterministic elementary functions against libm, and
known answers of the search.  PARITY UNPINNED at the reference level (path_searching needs ROS / Eigen / boost); these tests
pin the oracle to what the algorithm must produce."""
import ctypes
import math

import numpy as np

from forces_resilient_planner_amd import workloads

from . import astar_lib as AL


def _ulps(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / math.ulp(b)


def test_deterministic_functions_are_within_one_ulp_of_libm():
    l = AL.lib()
    rng = np.random.default_rng(5)
    worst = dict(cbrt=0.0, acos=0.0, cos=0.0)
    for x in np.concatenate([rng.uniform(-1e3, 1e3, 20000), rng.uniform(-1e-3, 1e-3, 2000), 10.0 ** rng.uniform(-12, 12, 4000)]):
        worst["cbrt"] = max(worst["cbrt"], _ulps(l.orc_det_cbrt(float(x)), float(np.cbrt(x))))
    for x in np.concatenate([rng.uniform(-1, 1, 20000), [1.0, -1.0, 0.0, 0.5, -0.5, 1e-20]]):
        worst["acos"] = max(worst["acos"], _ulps(l.orc_det_acos(float(x)), math.acos(float(x))))
    for x in rng.uniform(0.0, 2 * math.pi, 30000):  # cubic() evaluates cos on [0, 5 pi / 3]
        worst["cos"] = max(worst["cos"], abs(l.orc_det_cos(float(x)) - math.cos(float(x))) / math.ulp(1.0))
    assert worst["cbrt"] <= 1.0 and worst["acos"] <= 1.0 and worst["cos"] <= 1.0, worst
    assert math.isnan(l.orc_det_acos(1.0000001))


def _plan(world, start, goal, f=(0.0, 0.0, 0.0), v=(0.0, 0.0, 0.0), init=True):
    a = lambda x: np.asarray([x], dtype=float)
    r = AL.plan_batch(world, a(start), a(v), a((0, 0, 0)), a(goal), a((0, 0, 0)), a(f), init=init, nthreads=1)
    return r


def test_empty_world_gives_a_straight_line():
    w = workloads.astar_world(0, "empty")
    r = _plan(w, (-6.0, 0.05, 1.05), (4.0, 0.05, 1.05))
    assert r["status"][0] == 1  # the goal is farther than the horizon: REACH_HORIZON
    n = r["kino_size"][0]
    p = r["kino_path"][0, :n]
    assert n > 50 and np.all(np.abs(p[:, 1] - 0.05) < 1e-12) and np.all(np.abs(p[:, 2] - 1.05) < 1e-12)
    assert np.all(np.diff(p[:, 0]) >= 0.0) and p[-1, 0] - p[0, 0] >= 7.5
    res = r["results"][0]
    # accelerate at +max_acc until the velocity bound, then coast: the primitives' inputs along the path
    inputs = np.array([res.path_input[q][:] for q in range(1, res.n_path)])
    assert np.all(inputs[:, 1] == 0.0) and np.all(inputs[:, 2] == 0.0) and inputs[0, 0] == 3.0
    r2 = _plan(w, (-3.0, 0.05, 1.05), (1.0, 0.05, 1.05))
    assert r2["status"][0] == 2 and r2["results"][0].is_shot_succ == 1  # near the goal: REACH_END with the one-shot polynomial
    p2 = r2["kino_path"][0, :r2["kino_size"][0]]
    assert np.linalg.norm(p2[-1] - np.array([1.0, 0.05, 1.05])) < 0.12  # the last sample is within one Ts of the goal


def test_wall_with_a_gap_is_crossed_through_the_gap():
    w = workloads.astar_world(3, "wall_gap")
    r = _plan(w, (-4.0, w["gap_y"] + 2.5, 1.0), (4.0, w["gap_y"] - 1.0, 1.0))
    assert r["status"][0] in (1, 2, 4)
    p = r["kino_path"][0, :r["kino_size"][0]]
    cross = p[(p[:, 0] > -0.3) & (p[:, 0] < 0.3)]
    assert len(cross) > 0 and np.all(np.abs(cross[:, 1] - w["gap_y"]) < 1.0)
    # every primitive of the path is collision free when replayed under the acceleration it was planned for
    p_ = AL.make_params(w)
    assert AL.lib().orc_astar_replay(ctypes.byref(p_), (ctypes.c_double * 3)(0, 0, 0), ctypes.byref(r["results"][0])) == -1


def test_external_force_changes_the_plan_and_the_unforced_plan_becomes_infeasible():
    w = workloads.astar_world(3, "wall_gap")
    start, goal = (-2.2, w["gap_y"], 1.0), (3.0, w["gap_y"], 1.0)
    f = (0.0, 2.5, 0.0)
    r0 = _plan(w, start, goal)
    rf = _plan(w, start, goal, f=f)
    assert r0["status"][0] != 3 and rf["status"][0] != 3
    p_ = AL.make_params(w)
    F = (ctypes.c_double * 3)(*f)
    # planned without the force, flown with it: a primitive hits the wall; planned with it: feasible under it
    assert AL.lib().orc_astar_replay(ctypes.byref(p_), F, ctypes.byref(r0["results"][0])) >= 1
    assert AL.lib().orc_astar_replay(ctypes.byref(p_), F, ctypes.byref(rf["results"][0])) == -1
    assert AL.lib().orc_astar_replay(ctypes.byref(p_), (ctypes.c_double * 3)(0, 0, 0), ctypes.byref(r0["results"][0])) == -1


def test_blocked_start_retries_and_reports_no_path():
    w = workloads.astar_world(0, "empty", allocate_num=3000)
    w["occ"][:, :, :] = 0
    # a closed box around the start
    to = lambda p, i: int(np.floor((p - w["origin"][i]) / w["resolution"]))
    x0, x1, y0, y1 = to(-1.0, 0), to(1.0, 0), to(-1.0, 1), to(1.0, 1)
    w["occ"][x0:x1 + 1, y0, :] = 1; w["occ"][x0:x1 + 1, y1, :] = 1; w["occ"][x0, y0:y1 + 1, :] = 1; w["occ"][x1, y0:y1 + 1, :] = 1
    w["occ"][x0:x1 + 1, y0:y1 + 1, to(2.0, 2):] = 1
    r = _plan(w, (0.0, 0.0, 1.0), (6.0, 0.0, 1.0))
    assert r["status"][0] == 3 and r["retried"][0] == 1 and r["kino_size"][0] == 0
