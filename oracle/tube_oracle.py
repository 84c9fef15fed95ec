"""ORACLE for SURVEY 8f row f-2 (tube / ellipsoid propagation) -- TEST INFRASTRUCTURE ONLY.

Only tests/ (including tests/tools/) and __graft_entry__.smoke() may import this file; the product path
(forces_resilient_planner_amd/csrc/frp_tube.hip) never does.

CPU restatement, in numpy/scipy, of what NMPCSolver::setFORCESParams computes for every stage of the horizon
before each NLP solve (reference: src/resilient_planner/plan_manage/src/nmpc_solver.cpp):

    updateMatrix        :615-699   closed-loop linearisation Phi = A(z_i) + B(z_i) K
    eulerToRot          :554-565   R = Rz Ry Rx
    getDistrEllipsoid   :567-611   disturbance ellipsoid: per channel a Sylvester equation
                                   Phi X + X Phi' = N - exp(-Phi t) N exp(-Phi' t),  N = t w_i^2 d_i d_i',
                                   trace-optimal Minkowski outer ellipsoid, position block of
                                   exp(Phi t) Q exp(Phi' t)
    setFORCESParams     :484-521   ego ellipsoid R ego R', Minkowski sum with the previous stage's disturbance
                                   ellipsoid, E_i = sqrtm(Q) handed to the adapter (forces_normal.cpp:106-118)

The same numerical methods as the reference's Eigen calls are used (complex Schur + triangular Sylvester =
Bartels-Stewart via scipy.linalg.solve_sylvester, Pade expm, general eigendecomposition for sqrtm).

PARITY UNPINNED: Eigen (and its unsupported MatrixFunctions module) is not in this image and nmpc_solver.cpp
needs ROS, so the reference cannot be run here and its tests hold no vectors for this function.  The pin that IS
available: every step is a mathematically defined quantity (the Sylvester solution is unique whenever no two eigenvalues of Phi
sum to zero -- Phi is Hurwitz near zero yaw, where the reference's gain was designed -- and the square root is the
principal one), so any FP64 method must agree to rounding; tests check the GPU kernel -- which uses a
different method (Gauss-Legendre quadrature of the Gramian integral, Jacobi eigen-solver) -- against this
restatement at 1e-9 relative.

Two deliberate deviations from the letter of the reference, both reference defects that make its output depend on
memory garbage / call history and therefore cannot be a batched, stateless function:
  * `double temp` is accumulated into UNINITIALISED (nmpc_solver.cpp:573); it is 0 here, as SURVEY 8f prescribes;
  * At_(5,8) is only ever `+=`-ed (nmpc_solver.cpp:689) on a member matrix that is never cleared, so in the
    reference it sums over every call since construction; the fresh linearisation (d a_z / d yaw = the drag term
    alone) is used here.
"""
import numpy as np
import scipy.linalg as sl

NX, NU, NW = 9, 4, 3  # nmpc_utils.h:202-204

# nmpc_solver.cpp:28-31
KT = np.array([[-2.0, 5.0, 0.0, -1.0, 4.0, 0.0, -8.0, 0.0, 0.0],
               [-5.0, -2.0, 0.0, -4.0, -1.0, 0.0, 0.0, -8.0, 0.0],
               [-2.0, -2.0, 0.0, -1.0, -1.0, 0.0, 0.0, 0.0, -8.0],
               [0.0, 0.0, -8.0, 0.0, 0.0, -6.0, 0.0, 0.0, 0.0]])


def default_consts():
    """ROS parameter defaults (nmpc_solver.cpp:68-74) and nmpc_utils.h:188-189."""
    return dict(mass=0.74, drag=0.33, ego_r=0.27, ego_h=0.0425, noise=(0.5, 0.5, 0.5), epsilon=0.06, Ts=0.05)


def euler_to_rot(e):  # nmpc_solver.cpp:554-565
    cx, sx, cy, sy, cz, sz = np.cos(e[0]), np.sin(e[0]), np.cos(e[1]), np.sin(e[1]), np.cos(e[2]), np.sin(e[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def update_matrix(euler, vel, thrust, c):  # nmpc_solver.cpp:615-699
    roll, pitch, yaw = euler
    v1, v2, v3 = vel
    mass, drag = c["mass"], c["drag"]
    A = np.zeros((NX, NX)); A[0, 3] = A[1, 4] = A[2, 5] = 1.0
    B = np.zeros((NX, NU)); B[6, 0] = B[7, 1] = B[8, 2] = 1.0
    sr, cr, sp, cp, sy, cy = np.sin(roll), np.cos(roll), np.sin(pitch), np.cos(pitch), np.sin(yaw), np.cos(yaw)
    c0 = thrust / mass
    c5, c6, c7, c8, c9 = cp * sp, cp * sr, cp * cr, sp * cr, sp * sr
    c1 = cr * sy - c9 * cy
    c2 = sr * cy - c8 * sy
    c3 = cr * cy + c9 * sy
    c4 = sr * sy + c8 * cy
    A[3, 6], A[4, 6], A[5, 6] = c0 * c1, -c0 * c3, -c0 * c6
    A[3, 7], A[4, 7], A[5, 7] = c0 * c7 * cy, c0 * c7 * sy, -c0 * c8
    A[3, 8], A[4, 8] = c0 * c2, c0 * c4
    R = euler_to_rot(euler)
    A[3:6, 3:6] = R @ np.diag([drag, drag, 0.0]) @ R.T
    t1 = np.array([c6 * c4 - c7 * c1, c3 * c4 + c1 * c2, c6 * c2 - c7 * c3])
    A[3, 6] += drag * (v3 * t1[0] + v2 * t1[1] - 2 * v1 * c4 * c1)
    A[4, 6] += drag * (v1 * t1[1] - v3 * t1[2] - 2 * v2 * c3 * c2)
    A[5, 6] += drag * (v1 * t1[0] - v2 * t1[2] + 2 * v3 * c7 * c6)
    t2 = np.array([cy * (sp * sp - cp * cp + cp * cp * sr * sr) + c9 * c1,
                   2 * c5 * cy * sy - c6 * (cy * c3 + sy * c1),
                   sy * (cp * cp - sp * sp - cp * cp * sr * sr) + c9 * c3])
    A[3, 7] += drag * (v3 * t2[0] - v2 * t2[1] - v1 * 2 * (c5 * cy * cy + c6 * c1 * cy))
    A[4, 7] += -drag * (v3 * t2[2] - v1 * t2[1] - v2 * 2 * (c5 * sy * sy - c6 * c3 * sy))
    A[5, 7] += drag * (v1 * t2[0] - v2 * t2[2] + v3 * 2 * (c5 - c5 * sr * sr))
    t3 = np.array([2 * drag * (c3 * c1 - cp * cp * cy * sy), drag * (c6 * c3 - c5 * sy),
                   drag * (c3 * c3 - c1 * c1 - cp * cp * cy * cy + cp * cp * sy * sy), drag * (c6 * c1 + c5 * cy)])
    A[3, 8] += v1 * t3[0] - v3 * t3[1] - v2 * t3[2]
    A[4, 8] += -v1 * t3[2] - v3 * t3[3] - v2 * t3[0]
    A[5, 8] += -v2 * t3[3] - v1 * t3[1]
    B[3, 3], B[4, 3], B[5, 3] = c4 / mass, -c2 / mass, c7 / mass
    return A + B @ KT, R


def distr_ellipsoid(Phi, t, Q_origin, c):  # nmpc_solver.cpp:567-611; returns (position block, updated Q_origin)
    D = np.zeros((NX, NW)); D[3, 0] = D[4, 1] = D[5, 2] = 1.0  # :24-26
    Em, Ep = sl.expm(-Phi * t), sl.expm(Phi * t)
    temp, temp_Q = 0.0, np.zeros((NX, NX))  # `temp` is uninitialised in the reference (:573); 0 here
    for i in range(NW):
        Nt = t * c["noise"][i] ** 2 * np.outer(D[:, i], D[:, i])
        W = Nt - Em @ Nt @ Em.T
        X = sl.solve_sylvester(Phi, Phi.T, W)  # Phi X + X Phi' = W (complex Schur + triangular solve, :577-596)
        temp += np.sqrt(np.trace(X))
        temp_Q += X / np.sqrt(np.trace(X))
    Qd = temp * temp_Q
    beta = np.sqrt(np.trace(Q_origin) / np.trace(Qd))
    Q_update = (1 + 1 / beta) * Q_origin + (1 + beta) * Qd
    pos = Ep @ Q_update @ Ep.T
    return pos[0:3, 0:3], Q_update


def sqrtm3(Q):  # nmpc_solver.cpp:512-513 (general EigenSolver, V sqrt(L) V^-1, real part)
    lam, V = np.linalg.eig(Q)
    return (V @ np.diag(np.sqrt(lam.astype(complex))) @ np.linalg.inv(V)).real


def tube_one(z, c=None):
    """z [N,17] = the previous plan (mpc_output_, z = [u(4) w(4) x(9)]).  Returns E [N,3,3]
    (ellipsoid_matrices_, nmpc_solver.cpp:484-521)."""
    c = c or default_consts()
    N = z.shape[0]
    Q_init = c["epsilon"] ** 2 * np.eye(NX)
    ego = np.diag([c["ego_r"] ** 2, c["ego_r"] ** 2, c["ego_h"] ** 2])
    E = np.zeros((N, 3, 3))
    Q2 = None
    for i in range(N):
        Phi, R = update_matrix(z[i, 14:17], z[i, 11:14], z[i, 3], c)
        Q1 = R @ ego @ R.T
        if i == 0:
            Q = Q1
        else:
            beta = np.sqrt(np.trace(Q1) / np.trace(Q2))
            Q = (1 + 1 / beta) * Q1 + (1 + beta) * Q2
        E[i] = sqrtm3(Q)
        Q2, Q_init = distr_ellipsoid(Phi, c["Ts"], Q_init, c)
    return E


def tube_batch(z, c=None):
    return np.stack([tube_one(zb, c) for zb in z])
