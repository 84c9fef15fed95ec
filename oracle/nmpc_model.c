/*
 * nmpc_model.c -- TEST INFRASTRUCTURE (see nmpc_oracle.h).
 *
 * Hand-written FP64 restatement, with analytic first derivatives, of the reference NLP's stage
 * functions.  Specification followed (all under src/resilient_planner/plan_manage/):
 *   dynamics   matlab_code/dynamics/nonlinear_dynamics.m:21-40, dynamics/transit.m:7-8 (RK2 = Heun),
 *              constants as generated in solver/normal/FORCESNLPsolver_normal_casadi.c:238-259,358
 *   cost       matlab_code/mpc/mpc_objective1.m:19-48, mpc/normal/mpc_objective_normal.m:17-40,
 *              mpc/normal/mpc_objectiveN_normal.m:20-46, mpc/final/mpc_objectiveN_final.m:20-52
 *   corridor   matlab_code/mpc/mpc_corridorconst.m:5-10
 *   dispatch   solver/normal/FORCESNLPsolver_normal_casadi2forces.c:85,143,201
 */
#include <math.h>
#include <string.h>
#include "nmpc_oracle.h"

#define DT 0.05                 /* setup.m:37 */
#define MASS 0.745319           /* setup.m:17 */
#define GRAV 9.81               /* setup.m:18 */
#define DRAG 0.33               /* nonlinear_dynamics.m:27 */
#define HALF_PI 1.5707963267948966

/* body z axis zB(roll,pitch,yaw) = third column of R = Rz Ry Rx (nonlinear_dynamics.m:21-25)
 * and its partial derivatives wrt roll, pitch, yaw (columns of dzB, row-major 3x3). */
static void body_z(const double *eul, double *zb, double *dzb)
{
    const double sr = sin(eul[0]), cr = cos(eul[0]);
    const double sp = sin(eul[1]), cp = cos(eul[1]);
    const double sy = sin(eul[2]), cy = cos(eul[2]);
    zb[0] = cy * sp * cr + sy * sr;
    zb[1] = sy * sp * cr - cy * sr;
    zb[2] = cp * cr;
    if (dzb) {
        /* d/droll */
        dzb[0 * 3 + 0] = -cy * sp * sr + sy * cr;
        dzb[1 * 3 + 0] = -sy * sp * sr - cy * cr;
        dzb[2 * 3 + 0] = -cp * sr;
        /* d/dpitch */
        dzb[0 * 3 + 1] = cy * cp * cr;
        dzb[1 * 3 + 1] = sy * cp * cr;
        dzb[2 * 3 + 1] = -sp * cr;
        /* d/dyaw */
        dzb[0 * 3 + 2] = -sy * sp * cr + cy * sr;
        dzb[1 * 3 + 2] = cy * sp * cr + sy * sr;
        dzb[2 * 3 + 2] = 0.0;
    }
}

/* acc = zB*T/m + f_ext - g e3 - R diag(d,d,0) R' v  (nonlinear_dynamics.m:27-32).
 * R diag(d,d,0) R' = d (I - zB zB') because R is orthonormal, so
 * acc = a zB - d v + f_ext - g e3 with a = T/m + d (zB.v).
 * Fvv = dacc/dv (3x3), Fve = dacc/deuler (3x3), gT = dacc/dT (3) -- row-major. */
static void accel(const double *v, const double *eul, double T, const double *fext,
                  double *acc, double *Fvv, double *Fve, double *gT)
{
    double zb[3], dzb[9];
    body_z(eul, zb, Fvv ? dzb : 0);
    const double zv = zb[0] * v[0] + zb[1] * v[1] + zb[2] * v[2];
    const double a = T / MASS + DRAG * zv;
    acc[0] = a * zb[0] - DRAG * v[0] + fext[0];
    acc[1] = a * zb[1] - DRAG * v[1] + fext[1];
    acc[2] = a * zb[2] - DRAG * v[2] + fext[2] - GRAV;
    if (!Fvv) return;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            Fvv[i * 3 + j] = DRAG * zb[i] * zb[j] - (i == j ? DRAG : 0.0);
        }
        gT[i] = zb[i] / MASS;
    }
    for (int j = 0; j < 3; j++) {
        const double dzv = dzb[0 * 3 + j] * v[0] + dzb[1 * 3 + j] * v[1] + dzb[2 * 3 + j] * v[2];
        for (int i = 0; i < 3; i++) Fve[i * 3 + j] = a * dzb[i * 3 + j] + DRAG * zb[i] * dzv;
    }
}

/* One Heun step x+ = x + dt/2 (k1 + k2), k2 = f(x + dt k1) (transit.m:7; verified numerically
 * against casadi_f1 in tests).  x = [p v eul], u = [rates(3) T].
 * Ax = dx+/dx (9x9 row-major), Bx = dx+/du (9x4 row-major); either may be NULL. */
void orc_rk2(const double *x, const double *u, const double *fext, double *xn, double *Ax, double *Bx)
{
    const double *p = x, *v = x + 3, *e = x + 6;
    const double T = u[3];
    double a1[3], a2[3], F1vv[9], F1ve[9], g1[3], F2vv[9], F2ve[9], g2[3];
    double vt[3], et[3];
    const int jac = (Ax != 0) || (Bx != 0);
    accel(v, e, T, fext, a1, jac ? F1vv : 0, F1ve, g1);
    for (int i = 0; i < 3; i++) {
        vt[i] = v[i] + DT * a1[i];
        et[i] = e[i] + DT * u[i];
    }
    accel(vt, et, T, fext, a2, jac ? F2vv : 0, F2ve, g2);
    for (int i = 0; i < 3; i++) {
        xn[i] = p[i] + 0.5 * DT * (v[i] + vt[i]);
        xn[3 + i] = v[i] + 0.5 * DT * (a1[i] + a2[i]);
        xn[6 + i] = e[i] + DT * u[i];
    }
    if (!jac) return;
    /* chain rule pieces: d(vt)/dv = I + dt F1vv, d(vt)/de = dt F1ve, d(vt)/dT = dt g1,
     * d(et)/de = I, d(et)/drates = dt I. */
    double D2v[9], D2e[9], D2T[3], D2w[9]; /* d acc2 / d(v, e, T, rates) */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double sv = F2vv[i * 3 + j], se = F2ve[i * 3 + j];
            for (int l = 0; l < 3; l++) {
                sv += DT * F2vv[i * 3 + l] * F1vv[l * 3 + j];
                se += DT * F2vv[i * 3 + l] * F1ve[l * 3 + j];
            }
            D2v[i * 3 + j] = sv;
            D2e[i * 3 + j] = se;
            D2w[i * 3 + j] = DT * F2ve[i * 3 + j];
        }
        double sT = g2[i];
        for (int l = 0; l < 3; l++) sT += DT * F2vv[i * 3 + l] * g1[l];
        D2T[i] = sT;
    }
    if (Ax) {
        memset(Ax, 0, 81 * sizeof(double));
        for (int i = 0; i < 3; i++) {
            Ax[i * 9 + i] = 1.0;
            Ax[(6 + i) * 9 + 6 + i] = 1.0;
            for (int j = 0; j < 3; j++) {
                Ax[i * 9 + 3 + j] = (i == j ? DT : 0.0) + 0.5 * DT * DT * F1vv[i * 3 + j];
                Ax[i * 9 + 6 + j] = 0.5 * DT * DT * F1ve[i * 3 + j];
                Ax[(3 + i) * 9 + 3 + j] = (i == j ? 1.0 : 0.0) + 0.5 * DT * (F1vv[i * 3 + j] + D2v[i * 3 + j]);
                Ax[(3 + i) * 9 + 6 + j] = 0.5 * DT * (F1ve[i * 3 + j] + D2e[i * 3 + j]);
            }
        }
    }
    if (Bx) {
        memset(Bx, 0, 36 * sizeof(double));
        for (int i = 0; i < 3; i++) {
            Bx[i * 4 + 3] = 0.5 * DT * DT * g1[i];
            Bx[(3 + i) * 4 + 3] = 0.5 * DT * (g1[i] + D2T[i]);
            Bx[(6 + i) * 4 + i] = DT;
            for (int j = 0; j < 3; j++) Bx[(3 + i) * 4 + j] = 0.5 * DT * D2w[i * 3 + j];
        }
    }
}

/* ---- second derivatives (exact Lagrangian Hessian of the dynamics) -------------------------------
 * For a fixed contraction vector gam:  phi_gam(v, e, T) = gam . acc(v, e, T)
 *   = a (gam.zB) - d gam.v + const,  a = T/m + d (zB.v).
 * Its Hessian wrt (T, v, e) has only the blocks (T,e), (v,e), (e,e).  Also returns
 * gv = d phi / d v = d zB (gam.zB) - d gam. */
static void phi_hess(const double *gam, const double *v, const double *e, double T,
                     double *hTe, double *Hve, double *Hee, double *gv)
{
    const double sr = sin(e[0]), cr = cos(e[0]), sp = sin(e[1]), cp = cos(e[1]), sy = sin(e[2]), cy = cos(e[2]);
    const double zb[3] = {cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr};
    double D1[3][3], D2[3][3][3];
    D1[0][0] = -cy * sp * sr + sy * cr; D1[0][1] = -sy * sp * sr - cy * cr; D1[0][2] = -cp * sr;
    D1[1][0] = cy * cp * cr;            D1[1][1] = sy * cp * cr;            D1[1][2] = -sp * cr;
    D1[2][0] = -sy * sp * cr + cy * sr; D1[2][1] = cy * sp * cr + sy * sr;  D1[2][2] = 0.0;
    const double rr[3] = {-zb[0], -zb[1], -zb[2]};
    const double rp[3] = {-cy * cp * sr, -sy * cp * sr, sp * sr};
    const double ry[3] = {sy * sp * sr + cy * cr, -cy * sp * sr + sy * cr, 0.0};
    const double pp[3] = {-cy * sp * cr, -sy * sp * cr, -cp * cr};
    const double py[3] = {-sy * cp * cr, cy * cp * cr, 0.0};
    const double yy[3] = {-zb[0], -zb[1], 0.0};
    for (int c = 0; c < 3; c++) {
        D2[0][0][c] = rr[c]; D2[0][1][c] = D2[1][0][c] = rp[c]; D2[0][2][c] = D2[2][0][c] = ry[c];
        D2[1][1][c] = pp[c]; D2[1][2][c] = D2[2][1][c] = py[c]; D2[2][2][c] = yy[c];
    }
    double s = 0, zv = 0, sj[3], aj[3];
    for (int c = 0; c < 3; c++) { s += gam[c] * zb[c]; zv += zb[c] * v[c]; }
    const double a = T / MASS + DRAG * zv;
    for (int j = 0; j < 3; j++) {
        sj[j] = 0; aj[j] = 0;
        for (int c = 0; c < 3; c++) { sj[j] += gam[c] * D1[j][c]; aj[j] += DRAG * D1[j][c] * v[c]; }
        hTe[j] = sj[j] / MASS;
    }
    for (int i = 0; i < 3; i++) {
        gv[i] = DRAG * zb[i] * s - DRAG * gam[i];
        for (int j = 0; j < 3; j++) Hve[i * 3 + j] = DRAG * (D1[j][i] * s + zb[i] * sj[j]);
    }
    for (int j = 0; j < 3; j++)
        for (int l = 0; l < 3; l++) {
            double sjl = 0, ajl = 0;
            for (int c = 0; c < 3; c++) { sjl += gam[c] * D2[j][l][c]; ajl += DRAG * D2[j][l][c] * v[c]; }
            Hee[j * 3 + l] = ajl * s + aj[j] * sj[l] + aj[l] * sj[j] + a * sjl;
        }
}

/* Hessian of y_x' x+(x, u) wrt zt = (rates(3), T, v(3), e(3)) -- 10 x 10 symmetric, row-major.
 * Only the position and velocity rows of x+ are non-linear: y enters through yp = y[0:3], yv = y[3:6].
 * x+_p = p + dt v + dt^2/2 acc1,  x+_v = v + dt/2 (acc1 + acc2),  acc2 = acc(v + dt acc1, e + dt w, T). */
void orc_rk2_hess(const double *x, const double *u, const double *fext, const double *yp, const double *yv, double *H)
{
    const double *v = x + 3, *e = x + 6;
    const double T = u[3];
    double a1[3], F1vv[9], F1ve[9], g1[3], vt[3], et[3];
    accel(v, e, T, fext, a1, F1vv, F1ve, g1);
    for (int i = 0; i < 3; i++) { vt[i] = v[i] + DT * a1[i]; et[i] = e[i] + DT * u[i]; }
    double alpha[3], beta[3], gam1[3], gvt[3];
    for (int i = 0; i < 3; i++) { alpha[i] = 0.5 * DT * DT * yp[i] + 0.5 * DT * yv[i]; beta[i] = 0.5 * DT * yv[i]; }
    double h2Te[3], H2ve[9], H2ee[9];
    phi_hess(beta, vt, et, T, h2Te, H2ve, H2ee, gvt);
    for (int i = 0; i < 3; i++) gam1[i] = alpha[i] + DT * gvt[i];
    double h1Te[3], H1ve[9], H1ee[9], dummy[3];
    phi_hess(gam1, v, e, T, h1Te, H1ve, H1ee, dummy);
    memset(H, 0, 100 * sizeof(double));
    /* term 1: Hessian of phi_gam1 at (T, v, e), variable indices T=3, v=4..6, e=7..9 */
    for (int j = 0; j < 3; j++) {
        H[3 * 10 + 7 + j] += h1Te[j]; H[(7 + j) * 10 + 3] += h1Te[j];
        for (int i = 0; i < 3; i++) { H[(4 + i) * 10 + 7 + j] += H1ve[i * 3 + j]; H[(7 + j) * 10 + 4 + i] += H1ve[i * 3 + j]; }
        for (int l = 0; l < 3; l++) H[(7 + j) * 10 + 7 + l] += H1ee[j * 3 + l];
    }
    /* term 2: J' H2 J with J = d(T, vt, et)/d zt (7 x 10) */
    double J[7 * 10], M2[7 * 7], W[7 * 10];
    memset(J, 0, sizeof J); memset(M2, 0, sizeof M2);
    J[0 * 10 + 3] = 1.0;
    for (int i = 0; i < 3; i++) {
        J[(1 + i) * 10 + 3] = DT * g1[i];
        for (int j = 0; j < 3; j++) {
            J[(1 + i) * 10 + 4 + j] = (i == j ? 1.0 : 0.0) + DT * F1vv[i * 3 + j];
            J[(1 + i) * 10 + 7 + j] = DT * F1ve[i * 3 + j];
        }
        J[(4 + i) * 10 + i] = DT;
        J[(4 + i) * 10 + 7 + i] = 1.0;
    }
    for (int j = 0; j < 3; j++) {
        M2[0 * 7 + 4 + j] = M2[(4 + j) * 7 + 0] = h2Te[j];
        for (int i = 0; i < 3; i++) M2[(1 + i) * 7 + 4 + j] = M2[(4 + j) * 7 + 1 + i] = H2ve[i * 3 + j];
        for (int l = 0; l < 3; l++) M2[(4 + j) * 7 + 4 + l] = H2ee[j * 3 + l];
    }
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 10; j++) {
            double acc = 0;
            for (int l = 0; l < 7; l++) acc += M2[i * 7 + l] * J[l * 10 + j];
            W[i * 10 + j] = acc;
        }
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double acc = 0;
            for (int l = 0; l < 7; l++) acc += J[l * 10 + i] * W[l * 10 + j];
            H[i * 10 + j] += acc;
        }
}

/* Constant (Gauss-Newton == exact, the cost is quadratic) stage cost pieces:
 * f = 1/2 z'Hz + q'z + const with H diagonal except the (u_i, w_i) couplings.
 * hd[17] = diag(H), hc = the u_i/w_i off-diagonal entry (-2 w_rate), q[17] linear term. */
void orc_cost_quadratic(const double *p, int stage_class, int model, double *hd, double *hc, double *q, double *cst)
{
    const double w_wp = p[6], w_in = p[7], w_rate = p[8], yaw_ref = p[9];
    const double kin = 2.0 * w_in / (HALF_PI * HALF_PI);
    memset(hd, 0, 17 * sizeof(double));
    memset(q, 0, 17 * sizeof(double));
    for (int i = 0; i < 4; i++) {
        hd[i] = 2.0 * w_rate + (i < 3 ? kin : 0.0);
        hd[4 + i] = 2.0 * w_rate;
    }
    if (stage_class == ORC_STAGE_FIRST)
        for (int i = 0; i < 3; i++) hd[4 + i] += 20.0 * w_in; /* mpc_objective1.m:38-41 */
    for (int i = 0; i < 3; i++) {
        hd[8 + i] = 2.0 * w_wp;
        q[8 + i] = -2.0 * w_wp * p[i];
    }
    hd[16] = 24.0 * w_wp;
    q[16] = -24.0 * w_wp * yaw_ref;
    if (stage_class == ORC_STAGE_LAST && model == ORC_MODEL_FINAL)
        for (int i = 0; i < 3; i++) hd[11 + i] = 40.0 * w_wp; /* mpc_objectiveN_final.m:26 */
    *hc = -2.0 * w_rate;
    if (cst) *cst = w_wp * (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 12.0 * w_wp * yaw_ref * yaw_ref;
}

void orc_stage_eval(const double *z, const double *p, int M, int stage_class, int model,
                    double *f, double *gf, double *c, double *Jc, double *h, double *Jh)
{
    if (f || gf) {
        const double w_wp = p[6], w_in = p[7], w_rate = p[8], yaw_ref = p[9];
        double cost = 0.0, g[17];
        memset(g, 0, sizeof g);
        for (int i = 0; i < 3; i++) {
            const double e = p[i] - z[8 + i];
            cost += w_wp * e * e;
            g[8 + i] = -2.0 * w_wp * e;
        }
        {
            const double e = yaw_ref - z[16];
            cost += 12.0 * w_wp * e * e;
            g[16] = -24.0 * w_wp * e;
        }
        for (int i = 0; i < 3; i++) {
            const double n = z[i] / HALF_PI;
            cost += w_in * n * n;
            g[i] += 2.0 * w_in * n / HALF_PI;
        }
        for (int i = 0; i < 4; i++) {
            const double e = z[i] - z[4 + i];
            cost += w_rate * e * e;
            g[i] += 2.0 * w_rate * e;
            g[4 + i] -= 2.0 * w_rate * e;
        }
        if (stage_class == ORC_STAGE_FIRST) {
            for (int i = 0; i < 3; i++) {
                cost += 10.0 * w_in * z[4 + i] * z[4 + i];
                g[4 + i] += 20.0 * w_in * z[4 + i];
            }
        }
        if (stage_class == ORC_STAGE_LAST && model == ORC_MODEL_FINAL) {
            for (int i = 0; i < 3; i++) {
                cost += 20.0 * w_wp * z[11 + i] * z[11 + i];
                g[11 + i] += 40.0 * w_wp * z[11 + i];
            }
        }
        if (f) *f = cost;
        if (gf) memcpy(gf, g, sizeof g);
    }
    if ((c || Jc) && stage_class != ORC_STAGE_LAST) {
        double xn[9], Ax[81], Bx[36];
        orc_rk2(z + 8, z, p + 3, xn, Jc ? Ax : 0, Jc ? Bx : 0);
        if (c) {
            memcpy(c, xn, 9 * sizeof(double));
            for (int i = 0; i < 4; i++) c[9 + i] = z[i]; /* input carry rows (mpc_generator_normal.m:4-5) */
        }
        if (Jc) {
            memset(Jc, 0, 13 * 17 * sizeof(double));
            for (int i = 0; i < 9; i++) {
                for (int j = 0; j < 4; j++) Jc[j * 13 + i] = Bx[i * 4 + j];
                for (int j = 0; j < 9; j++) Jc[(8 + j) * 13 + i] = Ax[i * 9 + j];
            }
            for (int i = 0; i < 4; i++) Jc[i * 13 + 9 + i] = 1.0;
        }
    }
    if (h) {
        const double *A = p + ORC_NPRE, *b = p + ORC_NPRE + 3 * M;
        for (int j = 0; j < M; j++)
            h[j] = A[3 * j] * z[8] + A[3 * j + 1] * z[9] + A[3 * j + 2] * z[10] - b[j];
    }
    if (Jh) {
        const double *A = p + ORC_NPRE;
        memset(Jh, 0, (size_t)M * 17 * sizeof(double));
        for (int j = 0; j < M; j++)
            for (int l = 0; l < 3; l++) Jh[(8 + l) * M + j] = A[3 * j + l];
    }
}

void orc_bounds(double *lb, double *ub)
{
    const double r = HALF_PI;                    /* deg2rad(90), setup.m:27-29 */
    const double tmax = 2.0 * GRAV * MASS, tmin = 0.5 * GRAV * MASS; /* setup.m:30-31 */
    const double PI = 3.14159265358979323846;
    const double u[17] = {r, r, r, tmax, r, r, r, tmax, 20.0, 20.0, 5.0, 2.0, 2.0, 2.0, 0.4 * PI, 0.4 * PI, 2.0 * PI};
    const double l[17] = {-r, -r, -r, tmin, -r, -r, -r, tmin, -20.0, -20.0, 0.0, -2.0, -2.0, -2.0, -0.4 * PI, -0.4 * PI, -2.0 * PI};
    memcpy(lb, l, sizeof l);
    memcpy(ub, u, sizeof u);
}


/* ---- timing anchor: the reference's OWN model callback (oracle/_ref, FORCESNLPsolver_*_casadi2forces,
 * casadi2forces.c:42-245 -- the only piece of the reference's hot path that can run without the ForcesPro licence)
 * against this file's restatement of the same stage evaluation, both called `reps` times on one thread.
 * fn = the callback's address (extfunc signature, FORCESNLPsolver_normal.h:321).  Returns ns per call. */
#include <time.h>
typedef void (*orc_extfunc)(double *x, double *y, double *lambda, double *params, double *pobj, double *g, double *c,
                            double *Jeq, double *h, double *Jineq, double *H, int stage, int iterations, int threadID);
static double now_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e9 * (double)t.tv_sec + (double)t.tv_nsec; }
void orc_time_callback(void *fn, int model, int reps, double *ns_reference, double *ns_port)
{
    double z[17], p[130], y[13] = {0}, lam[64] = {0}, f, gf[17], c[13], Jc[221], h[30], Jh[510];
    for (int i = 0; i < 17; i++) z[i] = 0.1 * (i + 1) - 0.8;
    z[3] = 7.0; z[7] = 7.2;
    for (int i = 0; i < 130; i++) p[i] = 0.0;
    p[0] = 1.0; p[1] = 2.0; p[2] = 1.5; p[3] = 0.5; p[4] = -0.3; p[5] = 0.2; p[6] = 7.0; p[7] = 1.0; p[8] = 80.0; p[9] = 0.3;
    for (int j = 0; j < 6; j++) { p[10 + 3 * j + j % 3] = (j & 1) ? -1.0 : 1.0; p[100 + j] = 2.0; }
    volatile double sink = 0.0;
    if (fn && ns_reference) {
        orc_extfunc cb = (orc_extfunc)fn;
        const double t0 = now_ns();
        for (int r = 0; r < reps; r++) { z[0] += 1e-9; cb(z, y, lam, p, &f, gf, c, Jc, h, Jh, 0, 3, 0, 0); sink += f + c[5] + Jc[40]; }
        *ns_reference = (now_ns() - t0) / reps;
    }
    if (ns_port) {
        const double t0 = now_ns();
        for (int r = 0; r < reps; r++) { z[0] += 1e-9; orc_stage_eval(z, p, 30, 1, model, &f, gf, c, Jc, h, Jh); sink += f + c[5] + Jc[40]; }
        *ns_port = (now_ns() - t0) / reps;
    }
    (void)sink;
}
