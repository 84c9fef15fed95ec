"""Seeded synthetic NMPC workloads = BASELINE.json configs[0..4].

Every generator returns a dict of numpy FP64 arrays already in the solver's packed layout
(built through ``adapter.ForcesAdapter.pack`` exactly like forces_normal.cpp:62-136 would):
  xinit [B,9], x0 [B,N,17], params [B,N,10+4M], nfaces [B,N] int32, model, N, M
plus the unpacked ingredients for tests (ref_pos, ref_yaw, f_ext, poly_A/poly_b, E).
Weights are the launch-file values (launch/rotors_sim.launch:56-66): stage (7, 1, 80),
terminal (12, 0.5); final mode: stage (12, 1.5), terminal (15, 0.5).
"""
from __future__ import annotations

import math

import numpy as np

from . import layout as L
from .adapter import ForcesAdapter, init_mpc_output

SEED0 = 20260928
NORMAL_WEIGHTS = (7.0, 1.0, 80.0, 12.0, 0.5)
EGO = np.array([0.27, 0.27, 0.0425])   # ego ellipsoid semi-axes (rotors_sim.launch:67-68)


def _weights(model):
    return (7.0, 1.0, 80.0, 12.0, 0.5) if model == L.MODEL_NORMAL else (12.0, 1.5, 80.0, 15.0, 0.5)


def _rot(eul):
    """R = Rz(yaw) Ry(pitch) Rx(roll) (nonlinear_dynamics.m:21-23), batched over leading dims."""
    r, p, y = eul[..., 0], eul[..., 1], eul[..., 2]
    sr, cr, sp, cp, sy, cy = np.sin(r), np.cos(r), np.sin(p), np.cos(p), np.sin(y), np.cos(y)
    R = np.empty(eul.shape[:-1] + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - cr * sy; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = cp * sy; R[..., 1, 1] = cy * cr + sy * sp * sr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
    return R


def _bbox_faces(seed_pos, heading):
    """DecompROS local bbox (2, 2, 1) about the seed segment [p, p + 0.1 h] with no obstacles
    (nmpc_solver.cpp:311-323, decomp_util/line_segment.h:47-85): 6 faces a_j.x <= b_j.
    seed_pos [...,3], heading [...] -> A [...,6,3], b [...,6]."""
    h = np.stack([np.cos(heading), np.sin(heading), np.zeros_like(heading)], -1)
    l = np.stack([-np.sin(heading), np.cos(heading), np.zeros_like(heading)], -1)
    e3 = np.zeros_like(h); e3[..., 2] = 1.0
    A = np.stack([h, -h, l, -l, e3, -e3], -2)
    p2 = seed_pos + 0.1 * h
    b = np.stack([(h * p2).sum(-1) + 2.0, -(h * seed_pos).sum(-1) + 2.0,
                  (l * seed_pos).sum(-1) + 2.0, -(l * seed_pos).sum(-1) + 2.0,
                  seed_pos[..., 2] + 1.0, -seed_pos[..., 2] + 1.0], -1)
    return A, b


def _random_states(rng, B):
    st = np.zeros((B, 9))
    st[:, 0:2] = rng.uniform(-5, 5, (B, 2))
    st[:, 2] = rng.uniform(0.5, 2.5, B)
    st[:, 3:6] = rng.uniform(-1, 1, (B, 3))
    st[:, 6:8] = rng.uniform(-0.2, 0.2, (B, 2))
    st[:, 8] = rng.uniform(-math.pi, math.pi, B)
    return st


def _line_reference(rng, st, N):
    B = st.shape[0]
    heading = st[:, 8] + rng.uniform(-0.3, 0.3, B)
    speed = rng.uniform(0.5, 2.0, B)
    t = (np.arange(N) + 1) * L.DT
    d = np.stack([np.cos(heading), np.sin(heading), np.zeros(B)], -1)
    ref_pos = st[:, None, 0:3] + speed[:, None, None] * t[None, :, None] * d[:, None, :]
    ref_yaw = np.repeat(heading[:, None], N, 1)
    return ref_pos, ref_yaw, heading


def _finish(ad, mpc_output, f_ext, ref_pos, ref_yaw, E, A, b, nf, model, weights=None):
    ad.set_paras(*(weights if weights is not None else _weights(model)))
    xinit, x0, params, nfaces = ad.pack(mpc_output, f_ext, ref_pos, ref_yaw, E, A, b, nf)
    return dict(xinit=xinit.copy(), x0=x0.copy(), params=params.copy(), nfaces=nfaces.copy(),
                model=model, N=ad.N, M=ad.M, B=ad.B, ref_pos=ref_pos, ref_yaw=ref_yaw, f_ext=f_ext,
                poly_A=A, poly_b=b, E=E, mpc_output=mpc_output)


def config0(model=L.MODEL_NORMAL, f_ext=(0.0, 0.0, 0.0), weights=None):
    """configs[0] 'plumbing' (SURVEY 8d Config 1 / Appendix B): one solve through the N=20 / 30-row ABI.
    SURVEY Appendix B used the normal-mode weights (7, 1, 80, 12, 0.5) for BOTH models: pass
    weights=NORMAL_WEIGHTS to reproduce its B-2 known answer with the final model."""
    N, M = L.N_REF, L.NH_REF
    st = np.array([[0, 0, 1, 0, 0, 0, 0, 0, 0.0]])
    ref_pos = np.zeros((1, N, 3)); ref_pos[0, :, 0] = 0.05 * (np.arange(N) + 1); ref_pos[0, :, 2] = 1.0
    ref_yaw = np.zeros((1, N))
    A1, b1 = _bbox_faces(st[:, 0:3], np.zeros(1))
    A = np.repeat(A1[:, None], N, 1); b = np.repeat(b1[:, None], N, 1)
    E = np.broadcast_to(np.diag(EGO), (1, N, 3, 3)).copy()
    nf = np.full((1, N), 6, dtype=np.int32)
    ad = ForcesAdapter(1, model, N, M)
    return _finish(ad, init_mpc_output(st, N), np.asarray([f_ext], dtype=np.float64), ref_pos, ref_yaw, E, A, b, nf, model, weights)


def config1(B=1024, seed=SEED0 + 2, model=L.MODEL_NORMAL, M=L.NH_REF):
    """configs[1]: N=20, zero external force, box bounds only (all corridor rows padded)."""
    N = L.N_REF
    rng = np.random.default_rng(seed)
    st = _random_states(rng, B)
    ref_pos, ref_yaw, _ = _line_reference(rng, st, N)
    A = np.zeros((B, N, 1, 3)); b = np.zeros((B, N, 1)); nf = np.zeros((B, N), dtype=np.int32)
    E = np.zeros((B, N, 3, 3))
    ad = ForcesAdapter(B, model, N, M)
    return _finish(ad, init_mpc_output(st, N), np.zeros((B, 3)), ref_pos, ref_yaw, E, A, b, nf, model)


def config2(B=4096, seed=SEED0 + 3, model=L.MODEL_NORMAL, M=L.NH_REF):
    """configs[2] (headline): N=20, constant f_ext ~ U[-3,3]^3, 6-face box corridor per stage in the
    path frame, tightened by ||E a_j|| with E = R(eul0) diag(ego) R(eul0)'."""
    N = L.N_REF
    rng = np.random.default_rng(seed)
    st = _random_states(rng, B)
    ref_pos, ref_yaw, heading = _line_reference(rng, st, N)
    f_ext = rng.uniform(-3, 3, (B, 3))
    A, b = _bbox_faces(ref_pos, np.repeat(heading[:, None], N, 1))
    R = _rot(st[:, 6:9])
    E1 = R @ (EGO[None, :, None] * np.swapaxes(R, -1, -2))
    E = np.repeat(E1[:, None], N, 1)
    nf = np.full((B, N), 6, dtype=np.int32)
    ad = ForcesAdapter(B, model, N, M)
    return _finish(ad, init_mpc_output(st, N), f_ext, ref_pos, ref_yaw, E, A, b, nf, model)


def config3(B=16384, seed=SEED0 + 4, model=L.MODEL_NORMAL, N=30, M=15):
    """configs[3]: N=30, per-stage sinusoidal f_ext, per-stage polytopes = 6 bbox faces + U{0..9}
    random tangent planes at 0.6..2.0 m from the stage reference point (<= 15 faces)."""
    rng = np.random.default_rng(seed)
    st = _random_states(rng, B)
    ref_pos, ref_yaw, heading = _line_reference(rng, st, N)
    f0 = rng.uniform(-2, 2, (B, 3)); amp = rng.uniform(0, 1, (B, 3)); ph = rng.uniform(0, 2 * math.pi, (B, 1))
    k = np.arange(N)[None, :, None]
    f_ext = f0[:, None, :] + amp[:, None, :] * np.sin(2 * math.pi * k / N + ph[:, None, :])
    A6, b6 = _bbox_faces(ref_pos, np.repeat(heading[:, None], N, 1))
    nx = rng.integers(0, 10, (B, N))
    nrm = rng.normal(size=(B, N, 9, 3)); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    off = rng.uniform(0.6, 2.0, (B, N, 9))
    bx = (nrm * ref_pos[:, :, None, :]).sum(-1) + off
    A = np.concatenate([A6, nrm], 2); b = np.concatenate([b6, bx], 2)
    nf = (6 + nx).astype(np.int32)
    R = _rot(st[:, 6:9])
    E1 = R @ (EGO[None, :, None] * np.swapaxes(R, -1, -2))
    E = np.repeat(E1[:, None], N, 1)
    ad = ForcesAdapter(B, model, N, M)
    return _finish(ad, init_mpc_output(st, N), f_ext, ref_pos, ref_yaw, E, A, b, nf, model)


def config4_nominal(B=65536, seed=SEED0 + 5, model=L.MODEL_NORMAL, M=L.NH_REF, ticks=20):
    """configs[4]: Monte-Carlo f_ext ~ N(fbar, 0.5^2 I) around ONE nominal problem, N=20; the
    receding-horizon loop (shift warm start, reference advanced one stage per tick) is driven by
    ``receding.run``.  Returns the tick-0 problem plus the long reference (N + ticks points)."""
    N = L.N_REF
    rng = np.random.default_rng(seed)
    st1 = _random_states(rng, 1)
    heading = st1[:, 8] + rng.uniform(-0.3, 0.3, 1)
    speed = rng.uniform(0.8, 1.6, 1)
    t = (np.arange(N + ticks) + 1) * L.DT
    d = np.stack([np.cos(heading), np.sin(heading), np.zeros(1)], -1)
    ref_long = st1[:, None, 0:3] + speed[:, None, None] * t[None, :, None] * d[:, None, :]
    fbar = rng.uniform(-2, 2, 3)
    f_ext = fbar[None, :] + 0.5 * rng.normal(size=(B, 3))
    st = np.repeat(st1, B, 0)
    ref_pos = np.repeat(ref_long[:, :N], B, 0)
    ref_yaw = np.repeat(heading[:, None], N, 1).repeat(B, 0)
    A, b = _bbox_faces(ref_pos, ref_yaw)
    R = _rot(st[:, 6:9])
    E1 = R @ (EGO[None, :, None] * np.swapaxes(R, -1, -2))
    E = np.repeat(E1[:, None], N, 1)
    nf = np.full((B, N), 6, dtype=np.int32)
    ad = ForcesAdapter(B, model, N, M)
    out = _finish(ad, init_mpc_output(st, N), f_ext, ref_pos, ref_yaw, E, A, b, nf, model)
    out.update(ref_long=ref_long, heading=heading, ticks=ticks)
    return out


def _box_corridor(origin, heading, along, lateral, vertical):
    """One box polytope per problem in the frame of `heading` about `origin` [B,3]: along = (lo, hi) extents [B] each,
    lateral / vertical = half widths [B].  Returns A [B,6,3], b [B,6] with a_j.x <= b_j."""
    h = np.stack([np.cos(heading), np.sin(heading), np.zeros_like(heading)], -1)
    l = np.stack([-np.sin(heading), np.cos(heading), np.zeros_like(heading)], -1)
    e3 = np.zeros_like(h); e3[..., 2] = 1.0
    A = np.stack([h, -h, l, -l, e3, -e3], -2)
    b = np.stack([(h * origin).sum(-1) + along[1], -(h * origin).sum(-1) + along[0],
                  (l * origin).sum(-1) + lateral, -(l * origin).sum(-1) + lateral,
                  origin[..., 2] + vertical, -origin[..., 2] + vertical], -1)
    return A, b


HARD_KINDS = ("far", "force", "tight", "replan")


def config_hard(B=128, seed=SEED0 + 41, model=L.MODEL_NORMAL, M=L.NH_REF, replan_solver=None):
    """Hard-but-feasible instances of the configs[2] problem class (N = 20, 6 corridor faces per stage), the family
    VERDICT r03 asked for: what the solver meets when the planner is in trouble rather than cruising.  Problem b is of kind
    HARD_KINDS[b % 4]:
      far     cold start (hover at the state) with the reference line starting 3..5 m away from the state: the velocity
              box (+-2 m/s, mpc_generator_normal.m:39) is active over most of the horizon, the corridor is one long box;
      force   |f_ext| horizontal 6..9 m/s^2 (the planner aborts at 10, nmpc_manage.cpp:404), vertical +-2: 40-degree tilt;
      tight   lateral and vertical corridor faces 5..10 cm beyond the tightening ||E a|| (forces_normal.cpp:124) of the
              reference line, lateral initial velocity and a lateral push;
      replan  a poor warm start: x0 is the converged plan for ANOTHER reference (heading 0.6..1.2 rad away) from the same state,
              as after a replan (nmpc_solver.cpp:154-215).  `replan_solver(w) -> z [B,N,17]` solves the old problems (tests pass
              the oracle, bench tools the device solver); without it the old plan is the hover guess shifted along the old line.
    Feasible by construction in the sense that a trajectory inside the boxes exists; that SLSQP on the reference callbacks
    solves an instance is recorded per instance in tests/golden/solutions_hard.npz (tests/tools/gen_golden.py)."""
    N = L.N_REF
    rng = np.random.default_rng(seed)
    st = _random_states(rng, B)
    kind = np.arange(B) % 4
    heading = st[:, 8] + rng.uniform(-0.3, 0.3, B)
    speed = rng.uniform(0.5, 2.0, B)
    t = (np.arange(N) + 1) * L.DT
    d = np.stack([np.cos(heading), np.sin(heading), np.zeros(B)], -1)
    lat = np.stack([-np.sin(heading), np.cos(heading), np.zeros(B)], -1)
    start = st[:, 0:3].copy()
    # far: the line starts 3..5 m away (random horizontal direction, +-0.4 m in z)
    far = kind == 0
    off_dir = rng.uniform(-math.pi, math.pi, B)
    off_len = rng.uniform(3.0, 5.0, B)
    off = np.stack([off_len * np.cos(off_dir), off_len * np.sin(off_dir), rng.uniform(-0.4, 0.4, B)], -1)
    start[far] += off[far]
    start[:, 2] = np.clip(start[:, 2], 0.6, 4.4)
    ref_pos = start[:, None, :] + speed[:, None, None] * t[None, :, None] * d[:, None, :]
    ref_yaw = np.repeat(heading[:, None], N, 1)
    f_ext = rng.uniform(-3, 3, (B, 3))
    # force: 6..9 m/s^2 horizontally
    frc = kind == 1
    fa = rng.uniform(-math.pi, math.pi, B); fm = rng.uniform(6.0, 9.0, B)
    f_ext[frc] = np.stack([fm * np.cos(fa), fm * np.sin(fa), rng.uniform(-2, 2, B)], -1)[frc]
    # tight: velocity mostly along the line, a lateral component, a lateral push
    tgt = kind == 2
    vlat = rng.uniform(-0.3, 0.3, B); valong = rng.uniform(0.0, 1.0, B)
    st[tgt, 3:6] = (valong[:, None] * d + vlat[:, None] * lat)[tgt]
    st[tgt, 5] = rng.uniform(-0.1, 0.1, B)[tgt]
    f_ext[tgt] = (rng.uniform(1.0, 3.0, B)[:, None] * np.sign(rng.uniform(-1, 1, B))[:, None] * lat + rng.uniform(-1, 1, (B, 1)) * d)[tgt]
    # corridors
    A, b = _bbox_faces(ref_pos, np.repeat(heading[:, None], N, 1))
    R = _rot(st[:, 6:9])
    E1 = R @ (EGO[None, :, None] * np.swapaxes(R, -1, -2))
    E = np.repeat(E1[:, None], N, 1)
    if far.any():  # one long box from 2 m behind the state to 2 m beyond the end of the line, in the frame of the state -> line direction
        to = ref_pos[:, -1, :] - st[:, 0:3]
        hd = np.arctan2(to[:, 1], to[:, 0]); ln = np.hypot(to[:, 0], to[:, 1])
        Af, bf = _box_corridor(st[:, 0:3], hd, (np.full(B, 2.0), ln + 2.0), np.full(B, 2.5), np.full(B, 1.2))
        A[far] = Af[far, None]; b[far] = bf[far, None]
    if tgt.any():
        slack = rng.uniform(0.05, 0.10, (B, 2))
        Ea = np.linalg.norm(np.einsum('bij,bfj->bfi', E1, A[:, 0]), axis=-1)  # tightening per face [B,6]
        for k in range(N):
            # lateral faces 2, 3 and vertical faces 4, 5: tightening + 5..10 cm about the line point of the stage
            b[tgt, k, 2] = ((lat * ref_pos[:, k]).sum(-1) + Ea[:, 2] + slack[:, 0])[tgt]
            b[tgt, k, 3] = (-(lat * ref_pos[:, k]).sum(-1) + Ea[:, 3] + slack[:, 0])[tgt]
            b[tgt, k, 4] = (ref_pos[:, k, 2] + Ea[:, 4] + slack[:, 1])[tgt]
            b[tgt, k, 5] = (-ref_pos[:, k, 2] + Ea[:, 5] + slack[:, 1])[tgt]
    nf = np.full((B, N), 6, dtype=np.int32)
    mpc = init_mpc_output(st, N)
    # replan: the warm start is the plan for the OLD reference (heading 0.6..1.2 rad away, another speed)
    rep = kind == 3
    if rep.any():
        turn = rng.uniform(0.6, 1.2, B) * np.sign(rng.uniform(-1, 1, B))
        h_old = heading - turn; s_old = rng.uniform(0.5, 2.0, B)
        d_old = np.stack([np.cos(h_old), np.sin(h_old), np.zeros(B)], -1)
        ref_old = st[:, None, 0:3] + s_old[:, None, None] * t[None, :, None] * d_old[:, None, :]
        Ao, bo = _bbox_faces(ref_old, np.repeat(h_old[:, None], N, 1))
        ad_o = ForcesAdapter(B, model, N, M)
        w_old = _finish(ad_o, init_mpc_output(st, N), f_ext, ref_old, np.repeat(h_old[:, None], N, 1), E, Ao, bo, nf, model)
        if replan_solver is not None:
            z_old = replan_solver(w_old)
        else:
            z_old = w_old["x0"].copy()
            z_old[:, :, 8:11] = ref_old
            z_old[:, :, 11:14] = (s_old[:, None] * d_old)[:, None, :]
        # the plan deque after the solve (update + duplicate the last row, nmpc_solver.cpp:524-543): row 1.. = the old plan
        mpc[rep, 1:N + 1, :] = z_old[rep]
        mpc[rep, 1, 8:17] = st[rep]  # stage 0 of a plan IS the state (x_0 = xinit)
    ad = ForcesAdapter(B, model, N, M)
    out = _finish(ad, mpc, f_ext, ref_pos, ref_yaw, E, A, b, nf, model)
    out["kind"] = kind
    return out


CONFIGS = {0: config0, 1: config1, 2: config2, 3: config3, 4: config4_nominal}


# ------------------------------------------------------------------ kinodynamic A* worlds (second half)
# search/* and occ_map/* values of plan_manage/launch/advanced_param.xml:56-66, 97-109 and rotors_sim.launch:4-6, 47-48, 67-68
ASTAR_DEFAULTS = dict(max_tau=0.5, init_max_tau=0.5, max_vel=2.0, max_acc=3.0, w_time=10.0, horizon=7.5, lambda_heu=5.0,
                      allocate_num=100000, check_num=15, tie_breaker=1.0 + 1.0 / 10000, resolution=0.1, ego_r=0.27, ego_h=0.0425)


def astar_world(seed=0, kind="pillars", map_size=(20.0, 20.0, 4.0), origin=(-10.0, -10.0, -1.0), n_obstacles=40, **overrides):
    """An occupancy grid occ[x][y][z] (uint8, 1 = occupied) plus the map / search constants of the reference's launch files.
    The A* bounds its states by (origin, map_size / 2) in x, y and (0.1, map_size_z / 2) in z (kinodynamic_astar.cpp:152-154), so
    the map is centred on the origin like the reference's (origin -20, size 40).
    kind: "empty" | "pillars" (random vertical boxes) | "wall_gap" (a wall across x = 0 with one gap)."""
    w = dict(ASTAR_DEFAULTS)
    w.update(overrides)
    res = w["resolution"]
    grid = tuple(int(np.ceil(m / res)) for m in map_size)
    occ = np.zeros(grid, dtype=np.uint8)
    rng = np.random.default_rng(SEED0 + 77 + seed)
    to_idx = lambda p, i: int(np.floor((p - origin[i]) / res))
    if kind == "pillars":
        for _ in range(n_obstacles):
            cx, cy = rng.uniform(-4.5, 4.5), rng.uniform(-7.0, 7.0)
            hx, hy = rng.uniform(0.15, 0.6), rng.uniform(0.15, 0.6)
            z1 = rng.uniform(1.2, 3.0) if rng.random() < 0.3 else 3.0
            occ[max(0, to_idx(cx - hx, 0)):to_idx(cx + hx, 0) + 1, max(0, to_idx(cy - hy, 1)):to_idx(cy + hy, 1) + 1, 0:to_idx(z1, 2)] = 1
    elif kind == "wall_gap":
        gap_y = rng.uniform(-4.0, 4.0); gap_w = rng.uniform(1.2, 2.0)
        occ[to_idx(-0.2, 0):to_idx(0.2, 0) + 1, :, :] = 1
        occ[to_idx(-0.2, 0):to_idx(0.2, 0) + 1, to_idx(gap_y - gap_w / 2, 1):to_idx(gap_y + gap_w / 2, 1) + 1, :] = 0
        w["gap_y"] = gap_y
    elif kind != "empty":
        raise ValueError(kind)
    w.update(occ=occ, origin=tuple(origin), map_size=tuple(map_size), kind=kind)
    return w


def astar_queries(B, seed=0, goal_dist=(4.0, 12.0), f_scale=1.5):
    """B planner queries for a world of astar_world's default size: start on the -x side (in free space by construction of the
    worlds: |x| > 5), goal on the +x side, small initial velocity, external acceleration ~ U[-f_scale, f_scale]^3."""
    rng = np.random.default_rng(SEED0 + 99 + seed)
    start = np.stack([rng.uniform(-8.5, -5.5, B), rng.uniform(-6.0, 6.0, B), rng.uniform(0.6, 1.6, B)], 1)
    d = rng.uniform(goal_dist[0], goal_dist[1], B)
    ang = rng.uniform(-0.5, 0.5, B)
    end = start + np.stack([d * np.cos(ang), d * np.sin(ang), rng.uniform(-0.3, 0.3, B)], 1)
    end[:, 0] = np.clip(end[:, 0], -9.0, 9.0); end[:, 1] = np.clip(end[:, 1], -9.0, 9.0); end[:, 2] = np.clip(end[:, 2], 0.5, 1.8)
    v0 = rng.uniform(-0.5, 0.5, (B, 3)) * np.array([1.0, 1.0, 0.2])
    a0 = np.zeros((B, 3))
    f = rng.uniform(-f_scale, f_scale, (B, 3)) * np.array([1.0, 1.0, 0.3])
    return dict(start_pt=start, start_v=v0, start_a=a0, end_pt=end, end_v=np.zeros((B, 3)), f_ext=f)
