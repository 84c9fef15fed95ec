// frp_tube.hip -- SURVEY 8f row f-2: the tube (ego + disturbance ellipsoid) propagation that
// NMPCSolver::setFORCESParams runs for every stage before each NLP solve, batched on the device.  Its output, the
// per-stage matrices E_i, is the `ellipsoid` input of frp_nmpc_pack_batch (f-1), so plan -> tube -> pack -> solve
// stays in HBM.
//
// Reference (src/resilient_planner/plan_manage/src/nmpc_solver.cpp):
//   updateMatrix :615-699, eulerToRot :554-565, getDistrEllipsoid :567-611, setFORCESParams :484-521,
//   constants :11-31 (A/B/D pattern, K), :68-99 (mass, drag, ego size, noise bound), nmpc_utils.h:188-189.
//
// What the reference does per stage with Eigen -- two 9x9 complex Schur forms, three triangular Sylvester solves,
// four Pade matrix exponentials, a general 3x3 eigendecomposition -- is restated for a GPU lane through the
// quantities those calls define:
//   * the Sylvester solution of  Phi X + X Phi' = N - e^{-Phi t} N e^{-Phi' t},  N = t w^2 d d',  is the Gramian
//     X = t w^2 int_0^t (e^{-Phi s} d)(e^{-Phi s} d)' ds  (differentiate the integrand; the Gramian always solves
//     the equation, and the solution is unique unless two eigenvalues of Phi sum to zero).  The integrand is
//     entire and ||Phi|| t < 2, so 8-point Gauss-Legendre is exact to rounding; the vectors e^{-Phi s_j} d are
//     stepped node to node with a 14-term Taylor series (||Phi|| ds < 0.4);
//   * only rows 0..2 of e^{Phi t} are used (the position block): three Taylor-stepped vectors e^{Phi' t} e_j;
//   * Phi is never formed densely: rows 0..2 are [0 I 0], rows 6..8 are the constant gain rows, so a product with
//     Phi or Phi' is 21 variable + 15 constant multiply-adds;
//   * sqrtm of the (symmetric positive definite) 3x3 sum by cyclic Jacobi, E = V sqrt(L) V'.
// Thread (stage k, channel i) owns disturbance channel i's Gramian and row i of e^{Phi t}; 21 stages per 64-lane
// wave, channels combined with wave shuffles (deterministic order).  The only sequential part -- the 9x9
// Minkowski recursion Q <- (1+1/beta) Q + (1+beta) Qd over the stages -- runs on 45 lanes afterwards.
//
// FP64 VALU-bound, not HBM-bound: 8*17*N bytes in and 72*N bytes out per problem against ~0.9 MFLOP.
// Deliberate deviations from two reference defects (uninitialised `temp` :573 -> 0; At_(5,8) accumulated across
// calls :689 -> fresh value) are documented in oracle/tube_oracle.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/frp_nmpc.h"

namespace frp {

#ifndef FRP_TB_GSTEPS
#define FRP_TB_GSTEPS 2
#define FRP_TB_GTERMS 24
#endif
constexpr int TB_NZ = 17, TB_SYM = 45, TB_STAGES_PER_WAVE = 21, TB_TAYLOR = 14, TB_GSTEPS = FRP_TB_GSTEPS, TB_GTERMS = FRP_TB_GTERMS;

// K rows 0..2 (nmpc_solver.cpp:28-30); row 3 = [0 0 -8 0 0 -6 0 0 0] (:31) is folded into PhiS::b8 / m3.
#define TB_K(a, j) (tb_gain[(a) * 9 + (j)])
__device__ constexpr double tb_gain[27] = {-2.0, 5.0, 0.0, -1.0, 4.0, 0.0, -8.0, 0.0, 0.0,
                                           -5.0, -2.0, 0.0, -4.0, -1.0, 0.0, 0.0, -8.0, 0.0,
                                           -2.0, -2.0, 0.0, -1.0, -1.0, 0.0, 0.0, 0.0, -8.0};
__device__ constexpr double tb_glx[8] = {-0.9602898564975362, -0.7966664774136267, -0.525532409916329, -0.18343464249564978,
                                         0.18343464249564978, 0.525532409916329,   0.7966664774136267, 0.9602898564975362};
__device__ constexpr double tb_glw[8] = {0.10122853629037669, 0.22238103445337434, 0.31370664587788705, 0.36268378337836177,
                                         0.36268378337836177, 0.31370664587788705, 0.22238103445337434, 0.10122853629037669};

struct PhiS {          // the variable rows 3..5 of Phi = A + B K
    double b8[3];      // column 2:  -8 * B(3+a, 3)
    double m3[3][3];   // columns 3..5: R drag R' with -6 * B(3+a, 3) added to column 5
    double m6[3][3];   // columns 6..8: d a / d (roll, pitch, yaw)
};

__device__ __forceinline__ int sym_index(int m, int n) { return m * 9 - (m * (m - 1)) / 2 + (n - m); } // m <= n

// updateMatrix (:615-699) + eulerToRot (:554-565); R out row-major
__device__ void build_phi(const double *z, double mass, double drag, PhiS &P, double R[9])
{
    const double thrust = z[3], v1 = z[11], v2 = z[12], v3 = z[13], roll = z[14], pitch = z[15], yaw = z[16];
    double sr, cr, sp, cp, sy, cy;
    sincos(roll, &sr, &cr); sincos(pitch, &sp, &cp); sincos(yaw, &sy, &cy);
    const double c0 = thrust / mass;
    const double c5 = cp * sp, c6 = cp * sr, c7 = cp * cr, c8 = sp * cr, c9 = sp * sr;
    const double c1 = cr * sy - c9 * cy, c2 = sr * cy - c8 * sy, c3 = cr * cy + c9 * sy, c4 = sr * sy + c8 * cy;
    // R = Rz Ry Rx
    R[0] = cy * cp; R[1] = cy * c9 - sy * cr; R[2] = cy * c8 + sy * sr;
    R[3] = sy * cp; R[4] = sy * c9 + cy * cr; R[5] = sy * c8 - cy * sr;
    R[6] = -sp;     R[7] = c6;                R[8] = c7;
    const double t10 = c6 * c4 - c7 * c1, t11 = c3 * c4 + c1 * c2, t12 = c6 * c2 - c7 * c3;
    P.m6[0][0] = c0 * c1 + drag * (v3 * t10 + v2 * t11 - 2 * v1 * c4 * c1);
    P.m6[1][0] = -c0 * c3 + drag * (v1 * t11 - v3 * t12 - 2 * v2 * c3 * c2);
    P.m6[2][0] = -c0 * c6 + drag * (v1 * t10 - v2 * t12 + 2 * v3 * c7 * c6);
    const double sr2 = sr * sr, cp2 = cp * cp, sp2 = sp * sp;
    const double t20 = cy * (sp2 - cp2 + cp2 * sr2) + c9 * c1;
    const double t21 = 2 * c5 * cy * sy - c6 * (cy * c3 + sy * c1);
    const double t22 = sy * (cp2 - sp2 - cp2 * sr2) + c9 * c3;
    P.m6[0][1] = c0 * c7 * cy + drag * (v3 * t20 - v2 * t21 - v1 * 2 * (c5 * cy * cy + c6 * c1 * cy));
    P.m6[1][1] = c0 * c7 * sy - drag * (v3 * t22 - v1 * t21 - v2 * 2 * (c5 * sy * sy - c6 * c3 * sy));
    P.m6[2][1] = -c0 * c8 + drag * (v1 * t20 - v2 * t22 + v3 * 2 * (c5 - c5 * sr2));
    const double t30 = 2 * drag * (c3 * c1 - cp2 * cy * sy), t31 = drag * (c6 * c3 - c5 * sy);
    const double t32 = drag * (c3 * c3 - c1 * c1 - cp2 * cy * cy + cp2 * sy * sy), t33 = drag * (c6 * c1 + c5 * cy);
    P.m6[0][2] = c0 * c2 + v1 * t30 - v3 * t31 - v2 * t32;
    P.m6[1][2] = c0 * c4 - v1 * t32 - v3 * t33 - v2 * t30;
    P.m6[2][2] = -v2 * t33 - v1 * t31;
    // R diag(drag, drag, 0) R'
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) P.m3[a][c] = drag * (R[3 * a] * R[3 * c] + R[3 * a + 1] * R[3 * c + 1]);
    const double bt[3] = {c4 / mass, -c2 / mass, c7 / mass}; // Bt_(3..5, 3) (:692-694)
#pragma unroll
    for (int a = 0; a < 3; ++a) { P.b8[a] = -8.0 * bt[a]; P.m3[a][2] += -6.0 * bt[a]; }
}

__device__ __forceinline__ void phi_mul(const PhiS &P, const double v[9], double o[9])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = v[3 + a];
        double s = P.b8[a] * v[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) s += P.m3[a][c] * v[3 + c] + P.m6[a][c] * v[6 + c];
        o[3 + a] = s;
        double g = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) if (TB_K(a, j) != 0.0) g += TB_K(a, j) * v[j];
        o[6 + a] = g;
    }
}

__device__ __forceinline__ void phiT_mul(const PhiS &P, const double v[9], double o[9])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s0 = 0.0, s3 = v[c], s6 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (TB_K(a, c) != 0.0) s0 += TB_K(a, c) * v[6 + a];
            s3 += P.m3[a][c] * v[3 + a];
            if (TB_K(a, 3 + c) != 0.0) s3 += TB_K(a, 3 + c) * v[6 + a];
            s6 += P.m6[a][c] * v[3 + a];
            if (TB_K(a, 6 + c) != 0.0) s6 += TB_K(a, 6 + c) * v[6 + a];
        }
        if (c == 2) s0 += P.b8[0] * v[3] + P.b8[1] * v[4] + P.b8[2] * v[5];
        o[c] = s0; o[3 + c] = s3; o[6 + c] = s6;
    }
}

// v <- exp(h Phi) v  (TRANSPOSED: exp(h Phi') v), |h| ||Phi|| < 0.5: the TB_TAYLOR-term series in Horner form,
// v + h Phi (v + h/2 Phi (v + h/3 Phi (...))) -- one product with Phi and nine multiply-adds per term (round 5; the term-by-term sum
// cost a scaling and an addition per entry on top: 0.20 -> 0.18 ms per 4096 planners, same values to rounding)
template <bool TRANSPOSED, int TERMS = TB_TAYLOR>
__device__ __forceinline__ void expm_step(const PhiS &P, double h, double v[9])
{
    double y[9], nxt[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) y[j] = v[j];
    for (int n = TERMS; n >= 1; --n) {
        if (TRANSPOSED) phiT_mul(P, y, nxt); else phi_mul(P, y, nxt);
        const double f = h / (double)n;
#pragma unroll
        for (int j = 0; j < 9; ++j) y[j] = __builtin_fma(f, nxt[j], v[j]);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = y[j];
}

// principal square root of a symmetric positive definite 3x3 (q = xx xy xz yy yz zz), cyclic Jacobi; out row-major
__device__ void sqrt_sym3(const double q[6], double E[9])
{
    double a00 = q[0], a01 = q[1], a02 = q[2], a11 = q[3], a12 = q[4], a22 = q[5];
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#define TB_ROT(app, aqq, apq, arp, arq, p, q_)                                                   \
    if (apq != 0.0) {                                                                            \
        const double th = (aqq - app) / (2.0 * apq);                                             \
        const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));                   \
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                     \
        app -= t * apq; aqq += t * apq; apq = 0.0;                                               \
        const double rp = arp, rq = arq;                                                         \
        arp = c * rp - s * rq; arq = s * rp + c * rq;                                            \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                          \
            const double vp = V[i][p], vq = V[i][q_];                                            \
            V[i][p] = c * vp - s * vq; V[i][q_] = s * vp + c * vq;                               \
        }                                                                                        \
    }
    for (int sweep = 0; sweep < 8; ++sweep) {
        TB_ROT(a00, a11, a01, a02, a12, 0, 1)
        TB_ROT(a00, a22, a02, a01, a12, 0, 2)
        TB_ROT(a11, a22, a12, a01, a02, 1, 2)
    }
#undef TB_ROT
    const double l[3] = {sqrt(fmax(a00, 0.0)), sqrt(fmax(a11, 0.0)), sqrt(fmax(a22, 0.0))};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[3 * i + j] = l[0] * V[i][0] * V[j][0] + l[1] * V[i][1] * V[j][1] + l[2] * V[i][2] * V[j][2];
}

// LDS per stage: Qd (45) | G rows 0..2 of exp(Phi t) (27) | Q1 (6) | Q2 (6) | tr Qd (1)
constexpr int TS_QD = 0, TS_G = 45, TS_Q1 = 72, TS_Q2 = 78, TS_TR = 84, TS_STRIDE = 85;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tube_kernel(frp_nmpc_tube p)
{
    extern __shared__ double sm[];
    double *s_qo = sm + (size_t)p.N * TS_STRIDE; // 45: Q_origin of the running stage; afterwards 9 N outputs
    double *s_tmp = s_qo + (9 * p.N > TB_SYM ? 9 * p.N : TB_SYM); // 54: the products of the recursion's position block
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = wave * TB_STAGES_PER_WAVE + lane / 3, ch = lane % 3;
    const bool live = lane < 3 * TB_STAGES_PER_WAVE && k < p.N;
    const double *zb = p.mpc_output + (size_t)b * (p.N + 1) * TB_NZ;
    const double t = p.Ts;

    // ---- per (stage, channel): row ch of exp(Phi t), then the Gramian of channel ch ----------------------------
    double *st = sm + (size_t)(live ? k : 0) * TS_STRIDE;
    double X[TB_SYM];
#pragma unroll
    for (int e = 0; e < TB_SYM; ++e) X[e] = 0.0;
    double rootTr = 0.0;
    if (live) {
        PhiS P;
        {
            double z[TB_NZ], R[9];
#pragma unroll
            for (int j = 0; j < TB_NZ; ++j) z[j] = zb[k * TB_NZ + j];
            build_phi(z, p.mass, p.drag, P, R);
            if (ch == 0) { // ego_size_ (:90-92), Q1 = R ego R' (:503)
                const double er = p.ego_r * p.ego_r, eh = p.ego_h * p.ego_h;
                int e = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = a; c < 3; ++c)
                        st[TS_Q1 + e++] = er * (R[3 * a] * R[3 * c] + R[3 * a + 1] * R[3 * c + 1]) + eh * R[3 * a + 2] * R[3 * c + 2];
            }
        }
        double v[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = (j == ch) ? 1.0 : 0.0;
        // (rows of exp(Phi t): TB_GSTEPS equal steps.  Eight 14-term steps covered |h| ||Phi|| < 0.5 to rounding; two 24-term steps cover the
        // same range of ||Phi|| t (< 4: 2^25 / 25! = 2e-18) with 48 products instead of 112, at exp(2) ~ 7 roundings of cancellation instead of 2)
        for (int n = 0; n < TB_GSTEPS; ++n) expm_step<true, TB_GTERMS>(P, t / TB_GSTEPS, v);
#pragma unroll
        for (int j = 0; j < 9; ++j) { st[TS_G + 9 * ch + j] = v[j]; v[j] = (j == 3 + ch) ? 1.0 : 0.0; }
        double s_prev = 0.0;
        for (int n = 0; n < 8; ++n) {
            const double s = 0.5 * t * (1.0 + tb_glx[n]);
            expm_step<false>(P, -(s - s_prev), v);
            s_prev = s;
            const double wgt = 0.5 * t * tb_glw[n];
            int e = 0;
#pragma unroll
            for (int m = 0; m < 9; ++m)
#pragma unroll
                for (int nn = m; nn < 9; ++nn) X[e++] += wgt * v[m] * v[nn];
        }
        const double scale = t * p.noise[ch] * p.noise[ch]; // N = t w^2 d d' (:591)
        double tr = 0.0;
#pragma unroll
        for (int m = 0; m < 9; ++m) tr += X[sym_index(m, m)];
        rootTr = sqrt(scale * tr);
        const double nrm = scale / rootTr;                  // X / sqrt(trace X) (:598)
#pragma unroll
        for (int e = 0; e < TB_SYM; ++e) X[e] *= nrm;
    }
    // combine the three channels of a stage (lanes 3s, 3s+1, 3s+2 of one wave): Qd = (sum sqrt tr)(sum X/sqrt tr)
    const int base = lane - ch;
    const double temp = __shfl(rootTr, base) + __shfl(rootTr, base + 1) + __shfl(rootTr, base + 2);
#pragma unroll
    for (int e = 0; e < TB_SYM; ++e) {
        const double q = temp * (__shfl(X[e], base) + __shfl(X[e], base + 1) + __shfl(X[e], base + 2));
        if (live && e % 3 == ch) st[TS_QD + e] = q;
    }
    if (live && ch == 0) st[TS_TR] = temp * temp; // tr Qd = temp * sum_i tr(X_i)/sqrt(tr X_i) = temp^2
    __syncthreads();

    // ---- the stage recursion of getDistrEllipsoid's Q_origin (:603-608), 45 lanes ------------------------------
    int em = 0, en = 0; // (row, col) of packed entry tid
    if (tid < TB_SYM) {
        int e = tid;
        while (e >= 9 - em) { e -= 9 - em; ++em; }
        en = em + e;
    }
    double qo = (tid < TB_SYM && em == en) ? p.epsilon * p.epsilon : 0.0; // Q_init (:487)
    double trQo = 9.0 * p.epsilon * p.epsilon;
    for (int s = 0; s < p.N; ++s) {
        const double *ss = sm + (size_t)s * TS_STRIDE;
        const double trd = ss[TS_TR];
        const double beta = sqrt(trQo / trd);
        const double ca = 1.0 + 1.0 / beta, cb = 1.0 + beta;
        if (tid < TB_SYM) { qo = ca * qo + cb * ss[TS_QD + tid]; s_qo[tid] = qo; }
        trQo = ca * trQo + cb * trd;
        __syncthreads();
        // position block of exp(Phi t) Q exp(Phi' t) (:605, :609): entry (a, c) = sum_m G[a][m] (sum_n Q[m][n] G[c][n]).  54 lanes take one
        // (entry, m) each and six add them up -- on six lanes alone the 9 x 9 sums were 250 instructions of every stage of this
        // sequential recursion, a quarter of the kernel (round 5: 0.188 -> 0.17 ms per 4096 planners)
        if (tid < 54) {
            const int pr = tid / 9, m = tid - 9 * pr;
            const int a = pr < 3 ? 0 : (pr < 5 ? 1 : 2), c = pr < 3 ? pr : (pr < 5 ? pr - 2 : 2);
            double row = 0.0;
#pragma unroll
            for (int n = 0; n < 9; ++n) row += s_qo[m <= n ? sym_index(m, n) : sym_index(n, m)] * ss[TS_G + 9 * c + n];
            s_tmp[tid] = ss[TS_G + 9 * a + m] * row;
        }
        __syncthreads();
        if (tid < 6) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 9; ++m) acc += s_tmp[9 * tid + m];
            sm[(size_t)s * TS_STRIDE + TS_Q2 + tid] = acc;
        }
        __syncthreads();
    }

    // ---- per stage: Minkowski sum of ego and previous disturbance ellipsoid, square root (:503-513) -------------
    __syncthreads(); // s_qo is free now; the outputs leave through it so the global store is one contiguous run
    double *s_out = s_qo; // reuse: 9 N doubles (launcher sizes it)
    if (tid < p.N) {
        const double *ss = sm + (size_t)tid * TS_STRIDE;
        double q[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) q[e] = ss[TS_Q1 + e];
        if (tid > 0) {
            const double *q2 = ss - TS_STRIDE + TS_Q2;
            const double beta = sqrt((q[0] + q[3] + q[5]) / (q2[0] + q2[3] + q2[5]));
#pragma unroll
            for (int e = 0; e < 6; ++e) q[e] = (1.0 + 1.0 / beta) * q[e] + (1.0 + beta) * q2[e];
        }
        double E[9];
        sqrt_sym3(q, E);
#pragma unroll
        for (int j = 0; j < 9; ++j) s_out[9 * tid + j] = E[j];
    }
    __syncthreads();
    for (int e = tid; e < 9 * p.N; e += blockDim.x) p.ellipsoid[(size_t)b * 9 * p.N + e] = s_out[e];
}

} // namespace frp

extern "C" int frp_nmpc_tube_batch(const frp_nmpc_tube *p, void *stream)
{
    if (!p || p->B <= 0 || p->N < 1 || p->N > 64 || !p->mpc_output || !p->ellipsoid) return FRP_ERR_ARG;
    if (!(p->mass > 0.0) || !(p->Ts > 0.0) || !(p->epsilon > 0.0) || !(p->ego_r > 0.0) || !(p->ego_h > 0.0)) return FRP_ERR_ARG;
    for (int i = 0; i < 3; ++i) if (!(p->noise[i] > 0.0)) return FRP_ERR_ARG;
    const int waves = (p->N + frp::TB_STAGES_PER_WAVE - 1) / frp::TB_STAGES_PER_WAVE;
    const int tail = 9 * p->N > frp::TB_SYM ? 9 * p->N : frp::TB_SYM;
    const size_t lds = ((size_t)p->N * frp::TS_STRIDE + tail + 54) * sizeof(double);
    hipLaunchKernelGGL(frp::tube_kernel, dim3((unsigned)p->B), dim3(64 * waves), lds, static_cast<hipStream_t>(stream), *p);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}
