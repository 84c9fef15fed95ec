"""-m gpu: the device kinodynamic A* (frp_nmpc_astar_batch, SURVEY 8f row f-4 second half) against its CPU oracle
(oracle/astar_oracle.c).  The two execute the same IEEE operations, so everything is compared exactly: exit status, nodes
created, expansions, the retry, the ids of the path nodes (= the order in which the search created them) and the path samples."""
import numpy as np
import pytest

from forces_resilient_planner_amd import solver, workloads

from . import astar_lib as AL

pytestmark = pytest.mark.gpu


def _run_gpu(world, q, B, K=2048, active=None, planner=None, init=True):
    import torch
    pl = planner or solver.AstarPlanner(world, B, K=K, want_path_nodes=True)
    pl.upload(q["start_pt"], q["start_v"], q["start_a"], q["end_pt"], q["end_v"], q["f_ext"])
    pl.plan(init=init, active=active)
    torch.cuda.synchronize()
    return pl


def _compare(pl, o, B):
    st = pl.status.cpu().numpy(); sz = pl.kino_size.cpu().numpy(); stats = pl.stats.cpu().numpy()
    kp = pl.kino_path.cpu().numpy(); pn = pl.path_nodes.cpu().numpy()
    for b in range(B):
        r = o["results"][b]
        assert st[b] == o["status"][b], (b, st[b], o["status"][b])
        assert stats[b, 0] == r.use_node_num and stats[b, 1] == r.iter_num and stats[b, 2] == o["retried"][b], (b, stats[b], r.use_node_num, r.iter_num)
        if st[b] != solver.ASTAR_NO_PATH:
            n = r.n_path
            assert stats[b, 3] == n
            assert [int(pn[b, i, 10]) for i in range(n)] == [r.path_node[i] for i in range(n)]  # identical node sequence
            for i in range(n):
                assert np.array_equal(pn[b, i, 0:6], np.array(r.path_state[i][:])) and np.array_equal(pn[b, i, 6:9], np.array(r.path_input[i][:]))
            assert sz[b] == o["kino_size"][b]
            assert np.array_equal(kp[b, :sz[b]], o["kino_path"][b, :sz[b]])  # (the required tolerance is 1e-12; the samples are identical)


@pytest.mark.parametrize("chunk", range(6))
def test_device_astar_equals_the_oracle_on_seeded_worlds(chunk):
    """54 seeded worlds (empty / random pillars / wall with a gap), 4 planners each with random starts, goals, initial
    velocities and external accelerations."""
    B = 4
    for wi in range(chunk * 9, chunk * 9 + 9):
        kind = ["empty", "pillars", "wall_gap"][wi % 3]
        w = workloads.astar_world(100 + wi, kind, allocate_num=12000, n_obstacles=30)
        q = workloads.astar_queries(B, 100 + wi)
        o = AL.plan_batch(w, q["start_pt"], q["start_v"], q["start_a"], q["end_pt"], q["end_v"], q["f_ext"], nthreads=4)
        pl = _run_gpu(w, q, B)
        _compare(pl, o, B)


def test_device_astar_known_answers():
    """Empty world: a straight line; wall with a gap: through the gap; an external acceleration the unforced plan cannot
    survive: a different, feasible plan (feasibility replayed by the oracle's collision check)."""
    import ctypes
    w = workloads.astar_world(3, "wall_gap")
    gy = w["gap_y"]
    a = lambda *rows: np.asarray(rows, dtype=float)
    q = dict(start_pt=a((-4.0, gy + 2.5, 1.0), (-2.2, gy, 1.0), (-2.2, gy, 1.0)), start_v=np.zeros((3, 3)), start_a=np.zeros((3, 3)),
             end_pt=a((4.0, gy - 1.0, 1.0), (3.0, gy, 1.0), (3.0, gy, 1.0)), end_v=np.zeros((3, 3)), f_ext=a((0, 0, 0), (0, 0, 0), (0, 2.5, 0)))
    o = AL.plan_batch(w, q["start_pt"], q["start_v"], q["start_a"], q["end_pt"], q["end_v"], q["f_ext"], nthreads=2)
    pl = _run_gpu(w, q, 3)
    _compare(pl, o, 3)
    kp = pl.kino_path.cpu().numpy(); sz = pl.kino_size.cpu().numpy()
    p = kp[0, :sz[0]]
    cross = p[(p[:, 0] > -0.3) & (p[:, 0] < 0.3)]
    assert len(cross) > 0 and np.all(np.abs(cross[:, 1] - gy) < 1.0)
    assert not np.array_equal(kp[1, :sz[1]], kp[2, :sz[2]]) or sz[1] != sz[2]  # the force changes the plan
    we = workloads.astar_world(0, "empty")
    qe = dict(start_pt=a((-6.0, 0.05, 1.05)), start_v=np.zeros((1, 3)), start_a=np.zeros((1, 3)), end_pt=a((4.0, 0.05, 1.05)), end_v=np.zeros((1, 3)), f_ext=np.zeros((1, 3)))
    ple = _run_gpu(we, qe, 1)
    n = int(ple.kino_size.cpu()[0]); pe = ple.kino_path.cpu().numpy()[0, :n]
    assert int(ple.status.cpu()[0]) == solver.ASTAR_REACH_HORIZON and np.all(pe[:, 1] == 0.05) and np.all(pe[:, 2] == 1.05) and np.all(np.diff(pe[:, 0]) >= 0)


def test_inactive_planners_and_failed_searches_keep_their_path():
    import torch
    w = workloads.astar_world(7, "pillars", allocate_num=12000, n_obstacles=20)
    B = 6
    q = workloads.astar_queries(B, 7)
    pl = _run_gpu(w, q, B)
    path0 = pl.kino_path.clone(); size0 = pl.kino_size.clone()
    assert (size0 > 0).all()
    # second round: planners 0, 2 replan towards another goal, planner 4 towards a goal inside a sealed box (NO_PATH), the rest idle
    q2 = {k: v.copy() for k, v in q.items()}
    q2["end_pt"][[0, 2]] += np.array([0.0, 2.0, 0.0])
    to = lambda p, i: int(np.floor((p - w["origin"][i]) / w["resolution"]))
    s = q2["start_pt"][4]
    occ = w["occ"]
    x0, x1, y0, y1 = to(s[0] - 0.8, 0), to(s[0] + 0.8, 0), to(s[1] - 0.8, 1), to(s[1] + 0.8, 1)
    occ[x0:x1 + 1, y0, :] = 1; occ[x0:x1 + 1, y1, :] = 1; occ[x0, y0:y1 + 1, :] = 1; occ[x1, y0:y1 + 1, :] = 1; occ[x0:x1 + 1, y0:y1 + 1, to(s[2] + 0.5, 2):] = 1
    pl.occ.copy_(torch.from_numpy(occ).to(pl.occ.device))
    active = torch.tensor([1, 0, 1, 0, 1, 0], dtype=torch.int32, device=pl.occ.device)
    pl.upload(q2["start_pt"], q2["start_v"], q2["start_a"], q2["end_pt"], q2["end_v"], q2["f_ext"])
    pl.plan(active=active); torch.cuda.synchronize()
    st = pl.status.cpu().numpy()
    assert st[4] == solver.ASTAR_NO_PATH
    for b in (1, 3, 4, 5):
        assert torch.equal(pl.kino_path[b], path0[b]) and int(pl.kino_size[b]) == int(size0[b])
    o = AL.plan_batch(w, q2["start_pt"], q2["start_v"], q2["start_a"], q2["end_pt"], q2["end_v"], q2["f_ext"], nthreads=4)
    replanned = 0
    for b in (0, 2):
        n = int(pl.kino_size[b])
        assert st[b] == o["status"][b]
        if st[b] == solver.ASTAR_NO_PATH:
            assert torch.equal(pl.kino_path[b], path0[b]) and n == int(size0[b])
        else:
            replanned += 1
            assert n == o["kino_size"][b] and np.array_equal(pl.kino_path[b, :n].cpu().numpy(), o["kino_path"][b, :n])
            assert not torch.equal(pl.kino_path[b], path0[b])
    assert replanned >= 1


def test_astar_path_feeds_the_stage_references():
    """kino_path [B][K][3] / kino_size [B] are frp_nmpc_reference's per-planner path inputs (getKinoTraj -> getCurTraj)."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import reference_oracle as RO
    w = workloads.astar_world(11, "pillars", allocate_num=12000, n_obstacles=15)
    B, N = 5, 20
    q = workloads.astar_queries(B, 11)
    pl = _run_gpu(w, q, B, K=1024)
    dev = pl.occ.device
    mpc = np.zeros((B, N + 1, 17)); mpc[:, :, 8:11] = q["start_pt"][:, None, :]; mpc[:, :, 3] = mpc[:, :, 7] = 7.3
    toff = np.linspace(0.0, 0.6, B)
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device=dev); ry = torch.zeros((B, N), dtype=torch.float64, device=dev)
    fl = torch.zeros((B,), dtype=torch.int32, device=dev)
    solver.reference_batch_device(pl.kino_path, torch.from_numpy(toff).to(dev), torch.from_numpy(mpc).to(dev), rp, ry, fl, pl.kino_size)
    torch.cuda.synchronize()
    kp = pl.kino_path.cpu().numpy(); sz = pl.kino_size.cpu().numpy()
    for b in range(B):
        p_, y_, f_ = RO.references_one(kp[b], int(sz[b]), float(toff[b]), mpc[b], N)
        assert np.max(np.abs(rp[b].cpu().numpy() - p_)) < 1e-12 and np.max(np.abs(ry[b].cpu().numpy() - y_)) < 1e-12 and int(fl[b]) == int(f_)


def test_fleet_replans_the_flagged_planners_on_the_device():
    """DeviceFleet.replan = the FSM's REPLAN_TRAJ step: planners whose reference sampling raised kino_replan_ get a new path from
    their plan's next state, the others keep theirs; the new paths equal the oracle's for the same start."""
    import torch
    from forces_resilient_planner_amd import layout as L
    w = workloads.astar_world(21, "pillars", allocate_num=12000, n_obstacles=15)
    B, N = 6, 20
    q = workloads.astar_queries(B, 21)
    fleet = solver.DeviceFleet(B, N, 30, 30, L.MODEL_NORMAL, (7.0, 1.0, 80.0, 12.0, 0.5))
    dev = fleet.solver.device
    pl = solver.AstarPlanner(w, B, K=1024, want_path_nodes=True, device=str(dev))
    # every planner starts with a path from its own start
    pl.upload(q["start_pt"], q["start_v"], q["start_a"], q["end_pt"], q["end_v"], q["f_ext"]); pl.plan(); torch.cuda.synchronize()
    path0 = pl.kino_path.clone(); size0 = pl.kino_size.clone()
    # plans: hover at a point 1.5 m beside the path's start for planners 1 and 4 (-> "hard to follow", replan), on it for the rest
    mpc = np.zeros((B, N + 1, 17)); mpc[:, :, 3] = mpc[:, :, 7] = 7.3
    mpc[:, :, 8:11] = q["start_pt"][:, None, :]
    mpc[[1, 4], :, 9] += 1.5
    rng = np.random.default_rng(3)
    mpc[:, :, 11:14] = rng.uniform(-0.3, 0.3, (B, 1, 3)); mpc[:, :, 14:17] = rng.uniform(-0.2, 0.2, (B, 1, 3))
    fleet.mpc_output.copy_(torch.from_numpy(mpc).to(dev))
    toff = torch.full((B,), 0.3, dtype=torch.float64, device=dev)
    rp = torch.zeros((B, N, 3), dtype=torch.float64, device=dev); ry = torch.zeros((B, N), dtype=torch.float64, device=dev)
    flags = torch.zeros((B,), dtype=torch.int32, device=dev)
    fleet.references(pl.kino_path, toff, rp, ry, flags, pl.kino_size)
    torch.cuda.synchronize()
    assert flags.cpu().tolist() == [0, 1, 0, 0, 1, 0]
    end = torch.from_numpy(q["end_pt"]).to(dev); fext = torch.from_numpy(q["f_ext"]).to(dev)
    fleet.solver.exitflag.fill_(1)   # the last solve succeeded: the search starts from the plan (nmpc_solver.cpp:159-181)
    fleet.solver.exitflag[5] = 0     # ... except planner 5's (not flagged here; its start is checked below all the same)
    side = torch.cuda.Stream(dev)    # everything replan() does is enqueued on the stream it is given
    side.wait_stream(torch.cuda.current_stream(dev))
    ok = fleet.replan(pl, end, fext, flags, time_offset=toff, stream=side)
    torch.cuda.synchronize()
    st = pl.status.cpu().numpy()
    assert np.array_equal(pl._q[2][5].cpu().numpy(), np.zeros(3))  # exit_code != 1: odometry state, zero acceleration (:151-153, :187-189)
    # the oracle from the same starts
    m = mpc[:, 1]; e = m[:, 14:17]
    sr, cr, sp, cp, sy, cy = np.sin(e[:, 0]), np.cos(e[:, 0]), np.sin(e[:, 1]), np.cos(e[:, 1]), np.sin(e[:, 2]), np.cos(e[:, 2])
    acc = np.stack([cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr], 1) * (m[:, 3:4] / 0.74); acc[:, 2] -= 9.81
    up = [t.cpu().numpy() for t in pl._q]
    assert np.allclose(up[2][:5], acc[:5], rtol=0, atol=1e-12) and np.array_equal(up[0], m[:, 8:11])
    o = AL.plan_batch(w, up[0], up[1], up[2], up[3], up[4], up[5], nthreads=4, retry_pt=pl._retry[0].cpu().numpy(), retry_v=pl._retry[1].cpu().numpy())
    for b in range(B):
        n = int(pl.kino_size[b])
        if b in (1, 4) and st[b] != solver.ASTAR_NO_PATH:
            assert bool(ok[b]) and float(toff[b]) == 0.0
            assert st[b] == o["status"][b] and n == o["kino_size"][b] and np.array_equal(pl.kino_path[b, :n].cpu().numpy(), o["kino_path"][b, :n])
        else:
            assert not bool(ok[b]) and float(toff[b]) == 0.3 and torch.equal(pl.kino_path[b], path0[b]) and n == int(size0[b])
    assert bool(ok[1]) or bool(ok[4])


def test_repeated_search_starts_from_the_odometry_state():
    """getKinoPath repeats a failed search from the odometry state, not from the plan's interpolated one (nmpc_solver.cpp:190-193):
    frp_nmpc_astar.retry_pt / retry_vel.  Starts inside an obstacle fail the first search at once; the retry from a free state finds a
    path -- the oracle's, to the bit -- and from the same blocked state it does not."""
    import torch
    w = workloads.astar_world(33, "pillars", allocate_num=12000, n_obstacles=25)
    B = 8
    q = workloads.astar_queries(B, 33)
    occ = w["occ"]; res = w["resolution"]; org = np.array(w["origin"])
    ix, iy, iz = np.argwhere(occ[:, :, 5:15] > 0)[::max(1, int((occ[:, :, 5:15] > 0).sum()) // B)][:B].T
    blocked = org + (np.stack([ix, iy, iz + 5], 1) + 0.5) * res
    pl = solver.AstarPlanner(w, B, K=1024, want_path_nodes=True)
    pl.upload(blocked, np.zeros((B, 3)), q["start_a"], q["end_pt"], q["end_v"], q["f_ext"], retry=(q["start_pt"], q["start_v"]))
    pl.plan(); torch.cuda.synchronize()
    st = pl.status.cpu().numpy(); stats = pl.stats.cpu().numpy()
    o = AL.plan_batch(w, blocked, np.zeros((B, 3)), q["start_a"], q["end_pt"], q["end_v"], q["f_ext"], nthreads=4, cap=1024,
                      retry_pt=q["start_pt"], retry_v=q["start_v"])
    assert np.array_equal(st, o["status"]) and np.array_equal(stats[:, 2], o["retried"])
    assert (stats[:, 2] == 1).all() and (st != solver.ASTAR_NO_PATH).any()
    for b in range(B):
        if st[b] != solver.ASTAR_NO_PATH:
            n = int(pl.kino_size[b])
            assert n == o["kino_size"][b] and np.array_equal(pl.kino_path[b, :n].cpu().numpy(), o["kino_path"][b, :n])
    pl.upload(blocked, np.zeros((B, 3)), q["start_a"], q["end_pt"], q["end_v"], q["f_ext"])  # no retry state: the repeat starts blocked too
    pl.plan(); torch.cuda.synchronize()
    assert (pl.status.cpu().numpy() == solver.ASTAR_NO_PATH).all()
