#!/usr/bin/env python3
"""Numerical study for DESIGN 9.1 (CPU only; test infrastructure -- uses the oracle's stage functions): can the Newton system of an
interior-point iteration be solved from BOTH ends of the horizon at once?

  second half (stages m..N-1): the backward Riccati recursion the kernel runs today, stopped at stage m -> cost-to-go (P_m, p_m)
  first half  (stages 0..m-1): an ARRIVAL-cost recursion in information form, F_{k+1}(s+) = min_{w_k} [l_k(u, w, x) + F_k(w, x)]
                               with u = w+ - d_w and x = A^-1 (x+ - B u - d_x): a 4x4 pivot on w_k and two 13x13 congruences per
                               stage -- the same work as a backward stage.  The pinned x_0 makes the exact recursion rank-deficient
                               for three stages; here it is replaced by a penalty rho |dx_0 - r_0|^2 / 2.
  merge: ds_m = -(Q_m + P_m)^-1 (q_m + p_m); then the first half is back-substituted m-1..0 and the second half forward m..N-1.

The script builds the QP of a real iterate (linearisation and corridor rows from the oracle's stage functions at iterates of
BASELINE configs[2] and of the hard family, barrier terms from its slacks with mu = 1e-3 .. 1: diagonal entries up to ~1e10),
solves it (a) as one dense KKT system in extended precision-free float64 with numpy (truth), (b) by the backward recursion, (c) by
the twisted scheme for several rho, and prints the step errors relative to (a).
   python tools/study/twisted_riccati.py [problems=12]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from forces_resilient_planner_amd import layout as L, workloads as W
import tests.oracle_lib as OL

NU, NS, NXX = 4, 13, 9   # u; s = [w(4); x(9)]


def stage_qp(w, b, z, mu):
    """Per stage: Phi (17x17 over [u, w, x]), phi (17), A (9x9), B (9x4), d (13: residual of s+ = [u; A x + B u] + d)."""
    N, M = w["N"], w["M"]
    lb, ub = L.bounds()
    out = []
    for k in range(N):
        p = w["params"][b, k]
        sc = 0 if k == 0 else (2 if k == N - 1 else 1)
        ev = OL.stage_eval(z[k], p, M, sc, int(w["model"]))
        Phi = np.zeros((17, 17)); phi = ev["gf"].copy()
        # constant cost Hessian by differences of the gradient (the cost is quadratic)
        for i in range(17):
            e = np.zeros(17); e[i] = 1.0
            Phi[:, i] = OL.stage_eval(z[k] + e, p, M, sc, int(w["model"]))["gf"] - ev["gf"]
        Phi = 0.5 * (Phi + Phi.T)
        # bound barriers
        for i in range(17):
            sl, su = max(z[k, i] - lb[i], 1e-6), max(ub[i] - z[k, i], 1e-6)
            Phi[i, i] += mu / sl ** 2 + mu / su ** 2
            phi[i] += -mu / sl + mu / su
        nf = int(w["nfaces"][b, k])
        Aw = p[10:10 + 3 * M].reshape(M, 3)[:nf]; bw = p[10 + 3 * M:10 + 3 * M + nf]
        for j in range(nf):
            s = max(bw[j] + L.HU - Aw[j] @ z[k, 8:11], 1e-6)
            Phi[8:11, 8:11] += (mu / s ** 2) * np.outer(Aw[j], Aw[j])
            phi[8:11] += (mu / s) * Aw[j]
        if k < N - 1:
            J = ev["Jc"].reshape(17, 13).T          # d c / d z, rows: x+(9), then w+(4)
            A = J[0:9, 8:17]; B = J[0:9, 0:4]
            d = np.r_[z[k, 0:4] - z[k + 1, 4:8], ev["c"][0:9] - z[k + 1, 8:17]]   # [d_w; d_x]
        else:
            A = B = d = None
        out.append(dict(Phi=Phi, phi=phi, A=A, B=B, d=d))
    return out


def dense_truth(st, r0):
    """KKT of: min sum 1/2 dz'Phi dz + phi'dz  s.t. dx_0 = r0, ds_{k+1} = [du_k; A dx_k + B du_k] + d_k."""
    N = len(st); n = 17 * N; me = 9 + 13 * (N - 1)
    H = np.zeros((n, n)); g = np.zeros(n); C = np.zeros((me, n)); c = np.zeros(me)
    for k in range(N):
        H[17 * k:17 * k + 17, 17 * k:17 * k + 17] = st[k]["Phi"]; g[17 * k:17 * k + 17] = st[k]["phi"]
    C[0:9, 8:17] = np.eye(9); c[0:9] = r0
    for k in range(N - 1):
        r = 9 + 13 * k
        # w+ = u + d_w
        C[r:r + 4, 17 * (k + 1) + 4:17 * (k + 1) + 8] = np.eye(4); C[r:r + 4, 17 * k:17 * k + 4] = -np.eye(4); c[r:r + 4] = st[k]["d"][0:4]
        C[r + 4:r + 13, 17 * (k + 1) + 8:17 * (k + 1) + 17] = np.eye(9)
        C[r + 4:r + 13, 17 * k + 8:17 * k + 17] = -st[k]["A"]; C[r + 4:r + 13, 17 * k:17 * k + 4] = -st[k]["B"]; c[r + 4:r + 13] = st[k]["d"][4:13]
    K = np.block([[H, C.T], [C, np.zeros((me, me))]])
    # equilibrate: the barrier terms span ten orders of magnitude
    sc = 1.0 / np.sqrt(np.maximum(np.abs(np.diag(K)), 1.0))
    sol = np.linalg.solve(K * sc[:, None] * sc[None, :], np.r_[-g, c] * sc) * sc
    sol = sol + np.linalg.solve(K * sc[:, None] * sc[None, :], (np.r_[-g, c] - K @ sol) * sc) * sc   # one refinement step
    return sol[:n].reshape(N, 17)


def backward(st, k_from, k_to):
    """Backward recursion over stages k_from-1 .. k_to; returns (P, p) at k_to over s = [w; x] and per-stage gains for the forward pass."""
    N = len(st)
    P = np.zeros((13, 13)); p = np.zeros(13)
    gains = {}
    for k in range(k_from - 1, k_to - 1, -1):
        Phi, phi = st[k]["Phi"], st[k]["phi"]
        Hf = Phi.copy(); gf = phi.copy()                       # over [u(0:4), w(4:8), x(8:17)]
        if k < N - 1 and k + 1 <= k_from - 1 or (k + 1 == k_from and k_from < N) :
            A, B, d = st[k]["A"], st[k]["B"], st[k]["d"]
            T = np.zeros((13, 17)); T[0:4, 0:4] = np.eye(4); T[4:13, 0:4] = B; T[4:13, 8:17] = A   # s+ = T [u; w; x] + d
            Hf += T.T @ P @ T; gf += T.T @ (P @ d + p)
        # eliminate u
        Huu = Hf[0:4, 0:4]; Hus = Hf[0:4, 4:17]
        Kg = np.linalg.solve(Huu, np.c_[Hus, gf[0:4]])
        gains[k] = Kg
        P = Hf[4:17, 4:17] - Hus.T @ Kg[:, :13]; p = gf[4:17] - Hus.T @ Kg[:, 13]
        P = 0.5 * (P + P.T)
    return P, p, gains


def forward(st, gains, ds, k_from, k_to):
    """du_k = -K [ds_k; 1], ds_{k+1} = T [du; ds] + d for k = k_from .. k_to-1; returns dz rows."""
    out = {}
    for k in range(k_from, k_to):
        Kg = gains[k]
        du = -(Kg[:, :13] @ ds + Kg[:, 13])
        out[k] = np.r_[du, ds]
        if st[k]["A"] is not None:
            ds = np.r_[du, st[k]["A"] @ ds[4:13] + st[k]["B"] @ du] + st[k]["d"]
    return out


def arrival(st, m, r0, rho):
    """Arrival-cost recursion over stages 0..m-1 in information form; returns (Q_m, q_m) and what the back-substitution needs."""
    Q = np.zeros((13, 13)); q = np.zeros(13)
    Q[4:13, 4:13] = rho * np.eye(9); q[4:13] = -rho * r0
    keep = {}
    for k in range(m):
        Phi, phi, A, B, d = st[k]["Phi"], st[k]["phi"], st[k]["A"], st[k]["B"], st[k]["d"]
        Hf = Phi.copy(); gf = phi.copy()
        Hf[4:17, 4:17] += Q; gf[4:17] += q
        # eliminate w_k (4x4 pivot)
        Hww = Hf[4:8, 4:8]; idx = np.r_[0:4, 8:17]
        Hwr = Hf[4:8][:, idx]
        Kw = np.linalg.solve(Hww, np.c_[Hwr, gf[4:8]])          # w = -Kw [u; x; 1]
        G = Hf[np.ix_(idx, idx)] - Hwr.T @ Kw[:, :13]; gg = gf[idx] - Hwr.T @ Kw[:, 13]
        # [u; x] = T~ [w+; x+] + t~ :  u = w+ - d_w,  x = A^-1 (x+ - B u - d_x)
        Ai = np.linalg.inv(A)
        Tt = np.zeros((13, 13)); Tt[0:4, 0:4] = np.eye(4); Tt[4:13, 0:4] = -Ai @ B; Tt[4:13, 4:13] = Ai
        tt = np.r_[-d[0:4], Ai @ (B @ d[0:4] - d[4:13])]
        Q = Tt.T @ G @ Tt; q = Tt.T @ (G @ tt + gg)
        Q = 0.5 * (Q + Q.T)
        keep[k] = (Kw, Tt, tt)
    return Q, q, keep


def twisted(st, m, r0, rho):
    N = len(st)
    P, p, gains = backward(st, N, m)
    Q, q, keep = arrival(st, m, r0, rho)
    ds = -np.linalg.solve(Q + P, q + p)
    dz = forward(st, gains, ds, m, N)
    sp = ds
    for k in range(m - 1, -1, -1):
        Kw, Tt, tt = keep[k]
        ux = Tt @ sp + tt
        wk = -(Kw[:, :13] @ ux + Kw[:, 13])
        dz[k] = np.r_[ux[0:4], wk, ux[4:13]]
        sp = np.r_[wk, ux[4:13]]
    return np.array([dz[k] for k in range(N)])


def riccati_full(st, r0):
    N = len(st)
    P, p, gains = backward(st, N, 0)
    # stage 0: dx_0 = r0, dw_0 minimises
    dw = -np.linalg.solve(P[0:4, 0:4], P[0:4, 4:13] @ r0 + p[0:4])
    dz = forward(st, gains, np.r_[dw, r0], 0, N)
    return np.array([dz[k] for k in range(N)])


if __name__ == "__main__":
    nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rows = []
    for name, w in (("configs[2]", W.config2(nprob)), ("hard", W.config_hard(nprob, replan_solver=lambda wo: OL.solve_batch(wo)[0]))):
        for b in range(nprob):
            for maxit, mu in ((0, 1.0), (2, 1e-1), (4, 1e-3)):
                if maxit:
                    z, fl, info = OL.solve_one(w["xinit"][b], w["x0"][b], w["params"][b], w["nfaces"][b], w["N"], w["M"], int(w["model"]), OL.default_options(maxit=maxit))
                else:
                    z = w["x0"][b].copy()
                st = stage_qp(w, b, z, mu)
                r0 = w["xinit"][b] - z[0, 8:17]
                ref = dense_truth(st, r0)
                scale = np.max(np.abs(ref)) + 1e-12
                e_r = np.max(np.abs(riccati_full(st, r0) - ref)) / scale
                errs = [np.max(np.abs(twisted(st, 10, r0, rho) - ref)) / scale for rho in (1e8, 1e10, 1e12, 1e14, 1e16)]
                cond = max(np.max(np.diag(s_["Phi"])) for s_ in st)
                rows.append((name, b, maxit, mu, cond, e_r, *errs))
    print("workload  prob it  mu     max Phi_ii   | backward Riccati | twisted rho=1e8   1e10      1e12      1e14      1e16   (max |dz - dz_dense| / max |dz_dense|)")
    for r in rows:
        print("%-10s %3d %2d %7.0e %10.2e | %10.2e       | %9.2e %9.2e %9.2e %9.2e %9.2e" % r)
    a = np.array([r[5:] for r in rows])
    print("median:", " ".join("%.2e" % x for x in np.median(a, 0)), "  worst:", " ".join("%.2e" % x for x in np.max(a, 0)))
