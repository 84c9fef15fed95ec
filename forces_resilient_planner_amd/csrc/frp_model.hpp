// frp_model.hpp -- quadrotor NMPC stage model of the resilient planner, written for gfx950.
//
// What it evaluates (reference locations, all under src/resilient_planner/plan_manage/):
//   dynamics  : matlab_code/dynamics/nonlinear_dynamics.m:21-40 discretised by one Heun (RK2) step,
//               dynamics/transit.m:7-8; generated form solver/normal/FORCESNLPsolver_normal_casadi.c:235-1695
//   cost      : matlab_code/mpc/mpc_objective1.m:19-48, mpc/normal/mpc_objective_normal.m:17-40,
//               mpc/normal/mpc_objectiveN_normal.m:20-46, mpc/final/mpc_objectiveN_final.m:20-52
//   corridor  : matlab_code/mpc/mpc_corridorconst.m:5-10
//   bounds    : matlab_code/mpc/normal/mpc_generator_normal.m:33-46
// The reference evaluates these through CasADi-generated straight-line code; here they are derived
// by hand so that only the structurally non-zero Jacobian blocks are ever formed:
//   x+ = [p; v; e]+ ,  dp+/dv = Apv, dp+/de = Ape, dv+/dv = Avv, dv+/de = Ave,
//   dp+/dT = BpT, dv+/dT = BvT, dv+/drates = Bvw, de+/de = I, de+/drates = dt I, dp+/dp = I.
#pragma once
#include <hip/hip_runtime.h>

namespace frp {

constexpr int NU = 4, NW = 4, NX = 9, NZ = 17, NS = 13, NPRE = 10;
constexpr double DT = 0.05;                 // setup.m:37
constexpr double MASS = 0.745319;           // setup.m:17
constexpr double GRAV = 9.81;               // setup.m:18
constexpr double DRAG = 0.33;               // nonlinear_dynamics.m:27
constexpr double HALF_PI = 1.5707963267948966;
constexpr double PI = 3.14159265358979323846;
constexpr double HU = 1e-5;                 // mpc_generator_normal.m:14

enum StageClass { STAGE_FIRST = 0, STAGE_MID = 1, STAGE_LAST = 2 };
enum Model { MODEL_NORMAL = 0, MODEL_FINAL = 1 };

__host__ __device__ inline int stage_class(int k, int N) { return k == 0 ? STAGE_FIRST : (k == N - 1 ? STAGE_LAST : STAGE_MID); }

__host__ __device__ inline double lower_bound(int i)
{
    switch (i) {
    case 0: case 1: case 2: case 4: case 5: case 6: return -HALF_PI;
    case 3: case 7: return 0.5 * GRAV * MASS;
    case 8: case 9: return -20.0;
    case 10: return 0.0;
    case 11: case 12: case 13: return -2.0;
    case 14: case 15: return -0.4 * PI;
    default: return -2.0 * PI;
    }
}
__host__ __device__ inline double upper_bound(int i)
{
    switch (i) {
    case 0: case 1: case 2: case 4: case 5: case 6: return HALF_PI;
    case 3: case 7: return 2.0 * GRAV * MASS;
    case 8: case 9: return 20.0;
    case 10: return 5.0;
    case 11: case 12: case 13: return 2.0;
    case 14: case 15: return 0.4 * PI;
    default: return 2.0 * PI;
    }
}

// Compact linearisation of one RK2 step (51 doubles, the structural non-zeros only).
struct Lin {
    double Apv[9], Ape[9], Avv[9], Ave[9]; // row-major 3x3
    double BpT[3], BvT[3], Bvw[9];
};

struct AccJac {
    double Fvv[9], Fve[9], gT[3];
};

// sines / cosines of (roll, pitch, yaw): computed once per evaluation point and shared by the dynamics, its
// Jacobian and its Hessian (FP64 sincos is by far the most expensive scalar operation of the model)
struct Trig {
    double sr, cr, sp, cp, sy, cy;
};
// sin and cos of one angle, branch-free, for the device: the attitude angles of the planner are bounded (|roll|, |pitch|
// <= 0.4 pi, |yaw| <= 2 pi, mpc_generator_normal.m:33-46) and an interior-point iterate leaves the box by a few percent at
// most, so the two-constant Cody-Waite reduction by pi/2 (exact to ~n * 1e-33, n = quadrant count) followed by the
// fdlibm minimax kernels on [-pi/4, pi/4] is accurate to 1 ulp for |x| < 1e9 (the product n * pi/2_hi is formed exactly
// inside the fma; checked against libm on 4 M points up to 1e5).  A diverging iterate beyond that only loses accuracy,
// it ends in the solver's divergence guard.  The library sincos costs ~2.5x the instructions and carries the Payne-Hanek
// large-argument path behind divergent branches.
__device__ __forceinline__ void sincos_bounded(double x, double *sn, double *cs)
{
    const double n = rint(x * 6.36619772367581382433e-01); // x * 2 / pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    // __kernel_sin / __kernel_cos of fdlibm (Sun Microsystems, freely distributable minimax coefficients)
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}
__host__ __device__ inline Trig make_trig(const double e[3])
{
    Trig t;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos_bounded(e[0], &t.sr, &t.cr);
    sincos_bounded(e[1], &t.sp, &t.cp);
    sincos_bounded(e[2], &t.sy, &t.cy);
#else
    sincos(e[0], &t.sr, &t.cr);
    sincos(e[1], &t.sp, &t.cp);
    sincos(e[2], &t.sy, &t.cy);
#endif
    return t;
}

// acc = zB T/m + f_ext - g e3 - R diag(d,d,0) R' v with R diag(d,d,0) R' = d (I - zB zB')
template <bool JAC>
__host__ __device__ inline void accel_t(const double v[3], const Trig &tg, double T, const double fext[3],
                                        double acc[3], AccJac *J)
{
    const double sr = tg.sr, cr = tg.cr, sp = tg.sp, cp = tg.cp, sy = tg.sy, cy = tg.cy;
    const double zb0 = cy * sp * cr + sy * sr;
    const double zb1 = sy * sp * cr - cy * sr;
    const double zb2 = cp * cr;
    const double zv = zb0 * v[0] + zb1 * v[1] + zb2 * v[2];
    const double a = T * (1.0 / MASS) + DRAG * zv;
    acc[0] = a * zb0 - DRAG * v[0] + fext[0];
    acc[1] = a * zb1 - DRAG * v[1] + fext[1];
    acc[2] = a * zb2 - DRAG * v[2] + fext[2] - GRAV;
    if (JAC) {
        const double zb[3] = {zb0, zb1, zb2};
        // columns: d/droll, d/dpitch, d/dyaw
        const double dz[9] = {-cy * sp * sr + sy * cr, cy * cp * cr, -sy * sp * cr + cy * sr,
                              -sy * sp * sr - cy * cr, sy * cp * cr, cy * sp * cr + sy * sr,
                              -cp * sr, -sp * cr, 0.0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) J->Fvv[i * 3 + j] = DRAG * zb[i] * zb[j] - (i == j ? DRAG : 0.0);
            J->gT[i] = zb[i] * (1.0 / MASS);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double dzv = dz[j] * v[0] + dz[3 + j] * v[1] + dz[6 + j] * v[2];
#pragma unroll
            for (int i = 0; i < 3; i++) J->Fve[i * 3 + j] = a * dz[i * 3 + j] + DRAG * zb[i] * dzv;
        }
    }
}

template <bool JAC>
__host__ __device__ inline void accel(const double v[3], const double e[3], double T, const double fext[3],
                                      double acc[3], AccJac *J)
{
    accel_t<JAC>(v, make_trig(e), T, fext, acc, J);
}

// x = [p v e] (9), u = [rates(3) T]; xn = x + dt/2 (k1 + k2), k2 = f(x + dt k1)
template <bool JAC>
__host__ __device__ inline void rk2(const double x[9], const double u[4], const double fext[3], double xn[9], Lin *L)
{
    AccJac J1, J2;
    double a1[3], a2[3], vt[3], et[3];
    accel<JAC>(x + 3, x + 6, u[3], fext, a1, &J1);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        vt[i] = x[3 + i] + DT * a1[i];
        et[i] = x[6 + i] + DT * u[i];
    }
    accel<JAC>(vt, et, u[3], fext, a2, &J2);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        xn[i] = x[i] + 0.5 * DT * (x[3 + i] + vt[i]);
        xn[3 + i] = x[3 + i] + 0.5 * DT * (a1[i] + a2[i]);
        xn[6 + i] = et[i];
    }
    if (JAC) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double sT = J2.gT[i];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double sv = J2.Fvv[i * 3 + j], se = J2.Fve[i * 3 + j];
#pragma unroll
                for (int l = 0; l < 3; l++) {
                    sv += DT * J2.Fvv[i * 3 + l] * J1.Fvv[l * 3 + j];
                    se += DT * J2.Fvv[i * 3 + l] * J1.Fve[l * 3 + j];
                }
                L->Apv[i * 3 + j] = (i == j ? DT : 0.0) + 0.5 * DT * DT * J1.Fvv[i * 3 + j];
                L->Ape[i * 3 + j] = 0.5 * DT * DT * J1.Fve[i * 3 + j];
                L->Avv[i * 3 + j] = (i == j ? 1.0 : 0.0) + 0.5 * DT * (J1.Fvv[i * 3 + j] + sv);
                L->Ave[i * 3 + j] = 0.5 * DT * (J1.Fve[i * 3 + j] + se);
                L->Bvw[i * 3 + j] = 0.5 * DT * DT * J2.Fve[i * 3 + j];
                sT += DT * J2.Fvv[i * 3 + j] * J1.gT[j];
            }
            L->BpT[i] = 0.5 * DT * DT * J1.gT[i];
            L->BvT[i] = 0.5 * DT * (J1.gT[i] + sT);
        }
    }
}

// ---- second derivatives: exact Lagrangian Hessian of the dynamics ---------------------------------
// For a contraction vector gam: phi_gam(v, e, T) = gam . acc(v, e, T) = a (gam.zB) - d gam.v + const,
// a = T/m + d (zB.v).  Its Hessian wrt (T, v, e) has only the (T,e), (v,e), (e,e) blocks.
struct PhiHess {
    double hTe[3], Hve[9], Hee[9], gv[3]; // gv = d phi / d v
};

__host__ __device__ inline void phi_hess(const double gam[3], const double v[3], const Trig &tg, double T, PhiHess *o)
{
    const double sr = tg.sr, cr = tg.cr, sp = tg.sp, cp = tg.cp, sy = tg.sy, cy = tg.cy;
    const double zb[3] = {cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr};
    // D1[j] = d zB / d e_j ; D2[j][l] = d2 zB / d e_j d e_l
    const double D1[3][3] = {{-cy * sp * sr + sy * cr, -sy * sp * sr - cy * cr, -cp * sr},
                             {cy * cp * cr, sy * cp * cr, -sp * cr},
                             {-sy * sp * cr + cy * sr, cy * sp * cr + sy * sr, 0.0}};
    const double rr[3] = {-zb[0], -zb[1], -zb[2]};
    const double rp[3] = {-cy * cp * sr, -sy * cp * sr, sp * sr};
    const double ry[3] = {sy * sp * sr + cy * cr, -cy * sp * sr + sy * cr, 0.0};
    const double pp[3] = {-cy * sp * cr, -sy * sp * cr, -cp * cr};
    const double py[3] = {-sy * cp * cr, cy * cp * cr, 0.0};
    const double yy[3] = {-zb[0], -zb[1], 0.0};
    const double *D2[3][3] = {{rr, rp, ry}, {rp, pp, py}, {ry, py, yy}};
    double s = 0.0, zv = 0.0, sj[3], aj[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { s += gam[c] * zb[c]; zv += zb[c] * v[c]; }
    const double a = T * (1.0 / MASS) + DRAG * zv;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        sj[j] = gam[0] * D1[j][0] + gam[1] * D1[j][1] + gam[2] * D1[j][2];
        aj[j] = DRAG * (D1[j][0] * v[0] + D1[j][1] * v[1] + D1[j][2] * v[2]);
        o->hTe[j] = sj[j] * (1.0 / MASS);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o->gv[i] = DRAG * zb[i] * s - DRAG * gam[i];
#pragma unroll
        for (int j = 0; j < 3; j++) o->Hve[i * 3 + j] = DRAG * (D1[j][i] * s + zb[i] * sj[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const double *d2 = D2[j][l];
            const double sjl = gam[0] * d2[0] + gam[1] * d2[1] + gam[2] * d2[2];
            const double ajl = DRAG * (d2[0] * v[0] + d2[1] * v[1] + d2[2] * v[2]);
            o->Hee[j * 3 + l] = ajl * s + aj[j] * sj[l] + aj[l] * sj[j] + a * sjl;
        }
}

// Hessian of y_x' x+(x,u) wrt (rates(3), T, v(3), e(3)), 10 x 10 symmetric; only the position and
// velocity rows of x+ are non-linear (multipliers yp, yv).  Sink(i, j, value) receives every entry of
// the upper triangle (i <= j) exactly once.
//   x+_p = p + dt v + dt^2/2 acc1,  x+_v = v + dt/2 (acc1 + acc2),  acc2 = acc(v + dt acc1, e + dt w, T).
// Core with everything the RK2 evaluation already has: J1 = Jacobian of acc at (v, e), vt = v + dt acc1,
// tg1 / tg2 = trig of e and of e + dt rates.
template <typename Sink>
__host__ __device__ inline void rk2_hessian_core(const double v[3], double T, const AccJac &J1, const double vt[3],
                                                 const Trig &tg1, const Trig &tg2, const double yp[3], const double yv[3], Sink sink)
{
    double alpha[3], beta[3], gam1[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        alpha[i] = 0.5 * DT * DT * yp[i] + 0.5 * DT * yv[i];
        beta[i] = 0.5 * DT * yv[i];
    }
    PhiHess h2, h1;
    phi_hess(beta, vt, tg2, T, &h2);
#pragma unroll
    for (int i = 0; i < 3; i++) gam1[i] = alpha[i] + DT * h2.gv[i];
    phi_hess(gam1, v, tg1, T, &h1);
    // K = Jv' H2ve (+ h2Te on the T row): rows T, v(3), e(3); columns = e~ index
    double KT[3], Kv[9], Ke[9];
#pragma unroll
    for (int l = 0; l < 3; l++) {
        KT[l] = h2.hTe[l] + DT * (J1.gT[0] * h2.Hve[0 * 3 + l] + J1.gT[1] * h2.Hve[1 * 3 + l] + J1.gT[2] * h2.Hve[2 * 3 + l]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double kv = h2.Hve[i * 3 + l], ke = 0.0;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                kv += DT * J1.Fvv[m * 3 + i] * h2.Hve[m * 3 + l];
                ke += DT * J1.Fve[m * 3 + i] * h2.Hve[m * 3 + l];
            }
            Kv[i * 3 + l] = kv;
            Ke[i * 3 + l] = ke;
        }
    }
    // variable order: w0 w1 w2 (0..2), T (3), v (4..6), e (7..9)
#pragma unroll
    for (int j = 0; j < 3; j++) {
#pragma unroll
        for (int l = j; l < 3; l++) sink(j, l, DT * DT * h2.Hee[j * 3 + l]);                  // (w,w)
        sink(j, 3, DT * KT[j]);                                                              // (w,T)
#pragma unroll
        for (int i = 0; i < 3; i++) sink(j, 4 + i, DT * Kv[i * 3 + j]);                      // (w,v)
#pragma unroll
        for (int l = 0; l < 3; l++) sink(j, 7 + l, DT * (Ke[l * 3 + j] + h2.Hee[j * 3 + l])); // (w,e)
    }
    sink(3, 3, 0.0);
#pragma unroll
    for (int i = 0; i < 3; i++) sink(3, 4 + i, 0.0);                                         // (T,v)
#pragma unroll
    for (int l = 0; l < 3; l++) sink(3, 7 + l, h1.hTe[l] + KT[l]);                            // (T,e)
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = i; j < 3; j++) sink(4 + i, 4 + j, 0.0);                                 // (v,v)
#pragma unroll
        for (int l = 0; l < 3; l++) sink(4 + i, 7 + l, h1.Hve[i * 3 + l] + Kv[i * 3 + l]);    // (v,e)
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int l = j; l < 3; l++)
            sink(7 + j, 7 + l, h1.Hee[j * 3 + l] + Ke[j * 3 + l] + Ke[l * 3 + j] + h2.Hee[j * 3 + l]); // (e,e)
}

template <typename Sink>
__host__ __device__ inline void rk2_hessian(const double x[9], const double u[4], const double fext[3],
                                            const double yp[3], const double yv[3], Sink sink)
{
    const double *v = x + 3, *e = x + 6;
    AccJac J1;
    double a1[3], vt[3], et[3];
    const Trig tg1 = make_trig(e);
    accel_t<true>(v, tg1, u[3], fext, a1, &J1);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        vt[i] = v[i] + DT * a1[i];
        et[i] = e[i] + DT * u[i];
    }
    rk2_hessian_core(v, u[3], J1, vt, tg1, make_trig(et), yp, yv, sink);
}

// Dense entries of Ax = dx+/dx (9x9) and Bx = dx+/du (9x4) from the compact form.
__host__ __device__ inline double lin_A(const double *c /*Lin as 51 doubles*/, int i, int j)
{
    const int bi = i / 3, bj = j / 3, ii = i % 3, jj = j % 3;
    if (bi == 0) return bj == 0 ? (ii == jj ? 1.0 : 0.0) : c[(bj == 1 ? 0 : 9) + ii * 3 + jj];
    if (bi == 1) return bj == 0 ? 0.0 : c[(bj == 1 ? 18 : 27) + ii * 3 + jj];
    return (bj == 2 && ii == jj) ? 1.0 : 0.0;
}
__host__ __device__ inline double lin_B(const double *c, int i, int j)
{
    const int bi = i / 3, ii = i % 3;
    if (j == 3) return bi == 0 ? c[36 + ii] : (bi == 1 ? c[39 + ii] : 0.0);
    if (bi == 1) return c[42 + ii * 3 + j];
    if (bi == 2) return ii == j ? DT : 0.0;
    return 0.0;
}

// Stage cost f = 1/2 z'Hz + q'z + const: H = diag(hd) + hc on the (u_i, w_i) pairs.  p = 10 leading params.
struct CostQ {
    double w_wp, w_in, w_rate, yaw_ref, ref[3];
    int sc, model;
    __host__ __device__ inline double hd(int i) const
    {
        if (i < 3) return 2.0 * w_rate + 2.0 * w_in / (HALF_PI * HALF_PI);
        if (i == 3) return 2.0 * w_rate;
        if (i < 8) return 2.0 * w_rate + ((sc == STAGE_FIRST && i < 7) ? 20.0 * w_in : 0.0); // mpc_objective1.m:38-41
        if (i < 11) return 2.0 * w_wp;
        if (i < 14) return (sc == STAGE_LAST && model == MODEL_FINAL) ? 40.0 * w_wp : 0.0;    // mpc_objectiveN_final.m:26
        if (i == 16) return 24.0 * w_wp;
        return 0.0;
    }
    __host__ __device__ inline double hc() const { return -2.0 * w_rate; }
    __host__ __device__ inline double q(int i) const
    {
        if (i >= 8 && i < 11) return -2.0 * w_wp * ref[i - 8];
        if (i == 16) return -24.0 * w_wp * yaw_ref;
        return 0.0;
    }
};

__host__ __device__ inline CostQ make_cost(const double *p, int sc, int model)
{
    CostQ c;
    c.ref[0] = p[0]; c.ref[1] = p[1]; c.ref[2] = p[2];
    c.w_wp = p[6]; c.w_in = p[7]; c.w_rate = p[8]; c.yaw_ref = p[9];
    c.sc = sc; c.model = model;
    return c;
}

// f and gradient written in the reference's expression order (for the batched callback kernel)
__host__ __device__ inline double stage_cost(const double *z, const double *p, int sc, int model, double *g /*17 or null*/)
{
    const double w_wp = p[6], w_in = p[7], w_rate = p[8], yaw_ref = p[9];
    double cost = 0.0, gl[NZ];
#pragma unroll
    for (int i = 0; i < NZ; i++) gl[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double e = p[i] - z[8 + i];
        cost += w_wp * e * e;
        gl[8 + i] = -2.0 * w_wp * e;
    }
    {
        const double e = yaw_ref - z[16];
        cost += 12.0 * w_wp * e * e;
        gl[16] = -24.0 * w_wp * e;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double n = z[i] / HALF_PI;
        cost += w_in * n * n;
        gl[i] += 2.0 * w_in * n / HALF_PI;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double e = z[i] - z[4 + i];
        cost += w_rate * e * e;
        gl[i] += 2.0 * w_rate * e;
        gl[4 + i] -= 2.0 * w_rate * e;
    }
    if (sc == STAGE_FIRST) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            cost += 10.0 * w_in * z[4 + i] * z[4 + i];
            gl[4 + i] += 20.0 * w_in * z[4 + i];
        }
    }
    if (sc == STAGE_LAST && model == MODEL_FINAL) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            cost += 20.0 * w_wp * z[11 + i] * z[11 + i];
            gl[11 + i] += 40.0 * w_wp * z[11 + i];
        }
    }
    if (g) {
#pragma unroll
        for (int i = 0; i < NZ; i++) g[i] = gl[i];
    }
    return cost;
}

} // namespace frp
