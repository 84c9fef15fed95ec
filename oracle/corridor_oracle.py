"""ORACLE for SURVEY 8f row f-3 (corridor generation / selection) -- TEST INFRASTRUCTURE ONLY.

Only tests/ (including tests/tools/) and __graft_entry__.smoke() may import this file; the product path
(forces_resilient_planner_amd/csrc/frp_corridor.hip) never does.

numpy restatement of what NMPCSolver::getSikangConst does for every stage of the horizon
(src/resilient_planner/plan_manage/src/nmpc_solver.cpp:288-332) on top of the vendored DecompROS headers
(src/ThirdParty/DecompROS/decomp_ros_utils/include/):

    getSikangConst                    nmpc_solver.cpp:288-332   reuse the last polytope while the stage's tube
                                                                 ellipsoid (inflated 1.1x) fits, else decompose
    EllipsoidDecomp3D::dilate         decomp_util/ellipsoid_decomp.h:67-90
    EllipsoidDecomp::get_constraints  decomp_util/ellipsoid_decomp.h:47-56
    DecompBase::set_obs               decomp_util/decomp_base.h:33-38     cloud points inside the local box
    LineSegment::dilate               decomp_util/line_segment.h:31-35
    LineSegment::find_ellipsoid (3D)  decomp_util/line_segment.h:136-211
    DecompBase::find_polyhedron       decomp_util/decomp_base.h:63-83
    LineSegment::add_local_bbox       decomp_util/line_segment.h:47-85
    Ellipsoid::dist / closest_*       decomp_geometry/ellipsoid.h:19-58
    Polyhedron::inside, Hyperplane    decomp_geometry/polyhedron.h:14-60
    LinearConstraint(p0, planes)      decomp_geometry/polyhedron.h:98-118
    vec3_to_rotation                  decomp_geometry/geometric_utils.h:27-35
    epsilon_ = 1e-10                  decomp_basis/data_type.h:129

PARITY UNPINNED: DecompROS is header-only C++ over Eigen, and Eigen is not in this image, so the reference code
cannot be compiled or run here; it ships no test vectors for this path.  The restatement keeps the reference's order
of operations (order-preserving point lists, first-minimum tie breaking, the same thresholds) and the GPU kernel is
tested against it: identical polytope indices and row counts, rows within 1e-9.
"""
import numpy as np

EPS = 1e-10  # data_type.h:129


def _quat_to_rot(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def vec3_to_rotation(v):  # geometric_utils.h:27-35 (zero roll)
    pitch = np.arctan2(-v[2], np.hypot(v[0], v[1]))
    yaw = np.arctan2(v[1], v[0])
    Ry = _quat_to_rot(np.cos(pitch / 2), 0.0, np.sin(pitch / 2), 0.0)
    Rz = _quat_to_rot(np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2))
    return Rz @ Ry


def _dist(C, d, pts):  # ellipsoid.h:19-21
    return np.linalg.norm((pts - d) @ np.linalg.inv(C).T, axis=1)


def _closest(C, d, pts):  # ellipsoid.h:39-50 (first minimum)
    return pts[int(np.argmin(_dist(C, d, pts)))]


def local_bbox_planes(p1, p2, bbox):  # line_segment.h:47-85; list of (point, outward normal)
    if np.linalg.norm(bbox) == 0:
        return []
    dirv = (p2 - p1) / np.linalg.norm(p2 - p1)
    dir_h = np.array([dirv[1], -dirv[0], 0.0])
    if np.linalg.norm(dir_h) == 0:
        dir_h = np.array([-1.0, 0.0, 0.0])
    dir_h = dir_h / np.linalg.norm(dir_h)
    dir_v = np.cross(dirv, dir_h)
    return [(p1 + dir_h * bbox[1], dir_h), (p1 - dir_h * bbox[1], -dir_h),
            (p2 + dirv * bbox[0], dirv), (p1 - dirv * bbox[0], -dirv),
            (p1 + dir_v * bbox[2], dir_v), (p1 - dir_v * bbox[2], -dir_v)]


def find_ellipsoid(p1, p2, obs_, offset_x=0.0):  # line_segment.h:136-211; returns (C, d)
    f = np.linalg.norm(p1 - p2) / 2
    C = f * np.eye(3)
    axes = np.full(3, f)
    C[0, 0] += offset_x
    axes[0] += offset_x
    if axes[0] > 0:
        ratio = axes[1] / axes[0]
        axes = axes * ratio
        C = C * ratio
    Ri = vec3_to_rotation(p2 - p1)
    C = Ri @ C @ Ri.T
    d = (p1 + p2) / 2
    Rf = Ri
    obs = obs_[_dist(C, d, obs_) <= 1] if len(obs_) else obs_
    inside = obs
    while len(inside):
        pw = _closest(C, d, inside)
        p = Ri.T @ (pw - d)
        roll = np.arctan2(p[2], p[1])
        Rf = Ri @ _quat_to_rot(np.cos(roll / 2), np.sin(roll / 2), 0.0, 0.0)
        p = Rf.T @ (pw - d)
        if p[0] < axes[0]:
            axes[1] = abs(p[1]) / np.sqrt(1 - (p[0] / axes[0]) ** 2)
        C = Rf @ np.diag([axes[0], axes[1], axes[1]]) @ Rf.T
        inside = inside[1 - _dist(C, d, inside) > EPS]
    C = Rf @ np.diag(axes) @ Rf.T
    inside = obs[_dist(C, d, obs) <= 1] if len(obs) else obs
    while len(inside):
        pw = _closest(C, d, inside)
        p = Rf.T @ (pw - d)
        dd = 1 - (p[0] / axes[0]) ** 2 - (p[1] / axes[1]) ** 2
        if dd > EPS:
            axes[2] = abs(p[2]) / np.sqrt(dd)
        C = Rf @ np.diag(axes) @ Rf.T
        inside = inside[1 - _dist(C, d, inside) > EPS]
    return C, d


def decompose(p1, p2, cloud, bbox):
    """LineSegment(p1, p2): set_local_bbox, set_obs, dilate(0), then LinearConstraint((p1+p2)/2, planes).
    Returns (A [m,3], b [m])."""
    box = local_bbox_planes(p1, p2, bbox)
    obs_ = cloud
    for (pp, n) in box:  # Polyhedron::inside: rejected if signed_dist > epsilon_ (polyhedron.h:51-58)
        obs_ = obs_[(obs_ - pp) @ n <= EPS]
    C, d = find_ellipsoid(p1, p2, obs_)
    planes = []
    remain = obs_
    Ci = np.linalg.inv(C)
    while len(remain):  # decomp_base.h:63-83
        cp = _closest(C, d, remain)
        n = Ci @ Ci.T @ (cp - d)
        n = n / np.linalg.norm(n)
        planes.append((cp, n))
        remain = remain[(remain - cp) @ n < 0]
    planes += box
    p0 = (p1 + p2) / 2
    A = np.zeros((len(planes), 3)); b = np.zeros(len(planes))
    for i, (pp, n) in enumerate(planes):  # polyhedron.h:98-118
        c = pp @ n
        if n @ p0 - c > 0:
            n, c = -n, -c
        A[i], b[i] = n, c
    return A, b


def corridor_one(ref_pos, ref_yaw, E, cloud, bbox=(2.0, 2.0, 1.0), seed_len=0.1, inflation=1.1):
    """One planner, one tick: the getSikangConst calls of NMPCSolver::setFORCESParams (nmpc_solver.cpp:515).
    ref_pos [N,3], ref_yaw [N], E [N,3,3], cloud [P,3].  Returns (poly_index [N], list of (A, b))."""
    bbox = np.asarray(bbox, float)
    polys, index = [], []
    for i in range(len(ref_pos)):
        reuse = False
        if polys:
            A, b = polys[-1]
            reuse = True
            for j in range(len(b)):
                add = np.linalg.norm(E[i] @ A[j])
                if A[j] @ ref_pos[i] - (b[j] - inflation * add) > 0:
                    reuse = False
                    break
        if not reuse:
            p1 = np.array(ref_pos[i], float)
            p2 = p1 + np.array([seed_len * np.cos(ref_yaw[i]), seed_len * np.sin(ref_yaw[i]), 0.0])
            polys.append(decompose(p1, p2, cloud, bbox))
        index.append(len(polys) - 1)
    return np.array(index, dtype=np.int32), polys
