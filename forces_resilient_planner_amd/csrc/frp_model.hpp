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

// acc = zB T/m + f_ext - g e3 - R diag(d,d,0) R' v with R diag(d,d,0) R' = d (I - zB zB')
template <bool JAC>
__host__ __device__ inline void accel(const double v[3], const double e[3], double T, const double fext[3],
                                      double acc[3], AccJac *J)
{
    double sr, cr, sp, cp, sy, cy;
    sincos(e[0], &sr, &cr);
    sincos(e[1], &sp, &cp);
    sincos(e[2], &sy, &cy);
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
