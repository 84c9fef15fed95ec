// frp_kernels.hip -- gfx950 kernels of the batched NMPC solver (hand-written HIP, FP64).
//
// nmpc_ipm_kernel: one 64-lane wavefront == one workgroup == one NMPC problem, resident for the whole
// interior-point solve (no host round trips, per-problem early exit; the hardware dispatcher
// refills the slot with the next problem).  It replaces the reference's closed NLP solver
//   FORCESNLPsolver_{normal,final}_solve  (FORCESNLPsolver_normal.h:323; forces_normal.cpp:139)
// and its model callback FORCESNLPsolver_*_casadi2forces (casadi2forces.c:42-245).
//
// Iteration (same as the CPU oracle so both can be compared iterate by iterate):
//   primal-dual interior point, Mehrotra predictor-corrector, exact (constant) cost Hessian,
//   Newton KKT system solved by a Riccati recursion over the stage chain with
//   state s = [w; x] (13) and control u (4):   s_{k+1} = [u_k; A_k x_k + B_k u_k] + d_k.
// Work distribution inside the wavefront:
//   * stage-parallel phases (model evaluation, residuals, barrier terms, step lengths): lane == stage,
//     operands in [item][stage] arrays so that the 64 lanes read consecutive doubles;
//   * the serial Riccati chain: the 64 lanes share every small dense product through LDS
//     (P_{k+1}, [A|B], Q blocks staged in LDS; ~7 KB per problem), per-stage factors streamed
//     to/from a 208-double HBM record with coalesced wave loads, next stage prefetched.
#include <hip/hip_runtime.h>
#include <math.h>
#include "frp_model.hpp"
#include "../../include/frp_nmpc.h"
#include "frp_kernels.h"

namespace frp {

// occupancy target: waves per SIMD the register allocator must leave room for (propagates to the
// non-inlined phase functions)
#ifndef FRP_WAVES_PER_EU
#define FRP_WAVES_PER_EU 4
#endif

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// One wavefront per workgroup.  LDS operations of one wave are executed in issue order, so lanes can hand
// data to each other through LDS with only a COMPILER ordering fence: WSYNC() emits no instruction and,
// unlike __syncthreads(), does not drain the vector-memory counter -- prefetched global loads and
// streamed stores stay in flight across it.  FULLSYNC() (= __syncthreads()) is used at phase boundaries
// where lanes exchange data through global memory.
#define WSYNC()                                                   \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
#define FULLSYNC() __syncthreads()

// ------------------------------------------------------------------ LDS layout (doubles)
constexpr int L_P0 = 0;                 // P buffer 0 (13x13)
constexpr int L_P1 = L_P0 + 169;        // P buffer 1
constexpr int L_AB = L_P1 + 169;        // [A | B] 9 x 13 row-major
constexpr int L_D = L_AB + 117;         // d (13)
constexpr int L_PA = L_D + 13;          // Pxx [A|B]  9 x 13
constexpr int L_QUU = L_PA + 117;       // 4 x 4
constexpr int L_QUS = L_QUU + 16;       // 4 x 13  (cols 0..3 = Quw, 4..12 = Qux)
constexpr int L_Q = L_QUS + 52;         // q (17)
constexpr int L_OUT = L_Q + 17;         // [Kb 52 | R 16 | Pd 13 | kb 4 | p 13] = record part 2 (98)
constexpr int L_KB = L_OUT;
constexpr int L_R = L_OUT + 52;
constexpr int L_PD = L_OUT + 68;
constexpr int L_KV = L_OUT + 81;
constexpr int L_PV = L_OUT + 85;
constexpr int L_PHI = L_OUT + 98;       // [PhiD 17 | PhiPos 9 | phi 17] = record part 3 (43)
constexpr int L_PHID = L_PHI;
constexpr int L_PHIPOS = L_PHI + 17;
constexpr int L_PHIV = L_PHI + 26;
constexpr int L_HC = L_PHI + 43;        // hc of the stage being processed
constexpr int L_S0 = L_PHI + 44;        // stage-0 solve: [Rw 16 | Pwx 36]
constexpr int L_DS = L_S0 + 52;         // ds (13)
constexpr int L_DSN = L_DS + 13;        // next ds (13)
constexpr int L_DU = L_DSN + 13;        // du (4)
constexpr int L_YX = L_DU + 4;          // costate y_x (9) x 2
constexpr int L_TOTAL = L_YX + 18;

// ------------------------------------------------------------------ small device pieces
__device__ __forceinline__ int lin_dst(int t)
{
    // destination (offset from L_AB) of compact-linearisation entry t (0..50) / d entry (51..63)
    if (t < 9) return (t / 3) * 13 + 3 + t % 3;                     // Apv -> A[i][3+j]
    if (t < 18) return ((t - 9) / 3) * 13 + 6 + (t - 9) % 3;        // Ape -> A[i][6+j]
    if (t < 27) return (3 + (t - 18) / 3) * 13 + 3 + (t - 18) % 3;  // Avv
    if (t < 36) return (3 + (t - 27) / 3) * 13 + 6 + (t - 27) % 3;  // Ave
    if (t < 39) return (t - 36) * 13 + 12;                          // BpT -> B[i][3]
    if (t < 42) return (3 + t - 39) * 13 + 12;                      // BvT -> B[3+i][3]
    if (t < 51) return (3 + (t - 42) / 3) * 13 + 9 + (t - 42) % 3;  // Bvw -> B[3+i][j]
    return 117 + (t - 51);                                          // d
}

// symmetric positive definite 4x4 inverse via LDL'; returns false if a pivot is not positive
__device__ __forceinline__ bool spd4_inverse(const double *a /*row-major 4x4*/, double *r /*16*/)
{
    const double a00 = a[0], a10 = a[4], a11 = a[5], a20 = a[8], a21 = a[9], a22 = a[10];
    const double a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    const double d0 = a00;
    if (!(d0 > 0.0)) return false;
    const double i0 = 1.0 / d0;
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = a11 - l10 * a10;
    if (!(d1 > 0.0)) return false;
    const double i1 = 1.0 / d1;
    const double t21 = a21 - l20 * a10, t31 = a31 - l30 * a10;
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = a22 - l20 * a20 - l21 * t21;
    if (!(d2 > 0.0)) return false;
    const double i2 = 1.0 / d2;
    const double t32 = a32 - l30 * a20 - l31 * t21;
    const double l32 = t32 * i2;
    const double d3 = a33 - l30 * a30 - l31 * t31 - l32 * t32;
    if (!(d3 > 0.0)) return false;
    const double i3 = 1.0 / d3;
    // inverse of unit lower L: m = L^-1
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -l20 - l21 * m10;
    const double m31 = -l31 - l32 * m21;
    const double m30 = -l30 - l31 * m10 - l32 * m20;
    // R = m' D^-1 m
    const double r33 = i3;
    const double r32 = m32 * i3, r31 = m31 * i3, r30 = m30 * i3;
    const double r22 = i2 + m32 * r32;
    const double r21 = m21 * i2 + m32 * r31;
    const double r20 = m20 * i2 + m32 * r30;
    const double r11 = i1 + m21 * m21 * i2 + m31 * r31;
    const double r10 = m10 * i1 + m21 * m20 * i2 + m31 * r30;
    const double r00 = i0 + m10 * m10 * i1 + m20 * m20 * i2 + m30 * r30;
    r[0] = r00; r[1] = r10; r[2] = r20; r[3] = r30;
    r[4] = r10; r[5] = r11; r[6] = r21; r[7] = r31;
    r[8] = r20; r[9] = r21; r[10] = r22; r[11] = r32;
    r[12] = r30; r[13] = r31; r[14] = r32; r[15] = r33;
    return true;
}


// Explicit global address space: inside non-inlined device functions a plain double* is a GENERIC
// pointer and compiles to flat_load/flat_store, which also count on lgkmcnt and therefore make every
// LDS wait drain the in-flight prefetches.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double cgdouble;

struct WsView {
    gdouble *rec, *z, *y, *dz, *s, *lam, *corr, *face;
};

__host__ __device__ inline int padded_stages(int N) { return N <= 32 ? 32 : 64; }

__host__ __device__ inline size_t ws_doubles_per_problem(int N, int MF)
{
    const size_t mcf = 34 + MF, NPs = padded_stages(N);
    return (size_t)N * REC_STRIDE + NPs * (17 + 13 + 17 + 3 * mcf + 4 * (size_t)MF);
}

// The phases below are separate NON-inlined device functions on purpose: as one monolithic kernel
// body the compiler hoists hundreds of loop-invariant values (addresses, constants, index maths)
// across the solve loop and needs > 700 registers; as functions each phase is register-allocated on
// its own.  Their arguments are wave-uniform; readfirstlane restores that knowledge (SGPRs, scalar
// address arithmetic, uniform branches) on the callee side.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ WsView uni(WsView w)
{
    w.rec = uni(w.rec); w.z = uni(w.z); w.y = uni(w.y); w.dz = uni(w.dz);
    w.s = uni(w.s); w.lam = uni(w.lam); w.corr = uni(w.corr); w.face = uni(w.face);
    return w;
}

__shared__ double sm[L_TOTAL];

struct EvalOut {
    double eq, in, rs, rc, gap, obj;
};

// Staging area of the element-wise phases (aliases the Riccati working set, which is dead then):
// gm[17][NP] multiplier part of the stationarity residual, gf[6][NP] corridor sums for pos entries.
__shared__ double sm_big[23 * 64]; // only referenced (hence only allocated) by the NP = 64 instantiation
template <int NP>
__device__ __forceinline__ double *stage_area() { return NP == 32 ? sm : sm_big; }
static_assert(23 * 32 <= L_PHI, "element-wise staging must fit below the persistent LDS fields");

__device__ __forceinline__ void init_ab_constants(int lane)
{
    // constant entries of [A|B]: identity blocks of A, dt*I in B's euler rows (the variable entries are
    // scattered over them stage by stage)
    for (int t = lane; t < 117; t += 64) {
        const int i = t / 13, j = t % 13;
        double v = 0.0;
        if (j < 9) v = (i == j) ? 1.0 : 0.0;
        else if (i >= 6 && (j - 9) == (i - 6)) v = DT;
        sm[L_AB + t] = v;
    }
}

__device__ __forceinline__ double xhalf_sum(double v) { return v + __shfl_xor(v, 32); }

// ------------------------------------------------------------------ E: evaluate
// part 1 (lane == stage): model + linearisation -> record, equality residuals, M'y -> LDS
// part 2 (lane == (row pair, stage), all 64 lanes): corridor rows, then bounds: residual norms,
//        barrier Hessian / affine rhs -> record
template <int NP>
__device__ __noinline__ EvalOut phase_eval(WsView w, cgdouble *pbase, int np, cgdouble *xinit, int N, int MF, int nfk, int model)
{
    w = uni(w); pbase = uni(pbase); np = uni(np); xinit = uni(xinit); N = uni(N); MF = uni(MF); model = uni(model);
    FULLSYNC(); // phase boundary: other lanes' global writes of the previous phase are visible
    constexpr int H = 64 / NP;
    const int lane = threadIdx.x;
    double *stg = stage_area<NP>();
    double l_eq = 0, l_in = 0, l_rs = 0, l_rc = 0, l_gap = 0, l_obj = 0;
    if (lane < N) {
        const int k = lane;
        cgdouble *pk = pbase + (size_t)k * np;
        double p10[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
        const int sc_k = stage_class(k, N);
        double zk[NZ];
#pragma unroll
        for (int i = 0; i < NZ; i++) zk[i] = w.z[i * NP + k];
        l_obj = stage_cost(zk, p10, sc_k, model, nullptr);
        gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) l_eq = fmax(l_eq, fabs(xinit[i] - zk[8 + i]));
        }
        // gm = M' y_{k+1} - [0; y_k]  (multiplier part of the stationarity residual)
        double gm[NZ];
#pragma unroll
        for (int i = 0; i < 4; i++) gm[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; i++) gm[4 + i] = -w.y[i * NP + k];
        if (k < N - 1) {
            double yn[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) yn[i] = w.y[i * NP + k + 1];
            const double *yw = yn, *yp = yn + 4, *yv = yn + 7, *ye = yn + 10;
            // one Heun step with its linearisation streamed out entry by entry (record + M'y)
            AccJac J1, J2;
            double a1[3], a2[3], vt[3], et[3];
            accel<true>(zk + 11, zk + 14, zk[3], p10 + 3, a1, &J1);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                vt[i] = zk[11 + i] + DT * a1[i];
                et[i] = zk[14 + i] + DT * zk[i];
            }
            accel<true>(vt, et, zk[3], p10 + 3, a2, &J2);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double d = zk[i] - w.z[(4 + i) * NP + k + 1];
                rec[REC_D + i] = d;
                l_eq = fmax(l_eq, fabs(d));
                gm[i] += yw[i];
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double xp = zk[8 + i] + 0.5 * DT * (zk[11 + i] + vt[i]);
                const double xv = zk[11 + i] + 0.5 * DT * (a1[i] + a2[i]);
                const double dp = xp - w.z[(8 + i) * NP + k + 1];
                const double dv = xv - w.z[(11 + i) * NP + k + 1];
                const double de = et[i] - w.z[(14 + i) * NP + k + 1];
                rec[REC_D + 4 + i] = dp; rec[REC_D + 7 + i] = dv; rec[REC_D + 10 + i] = de;
                l_eq = fmax(l_eq, fmax(fabs(dp), fmax(fabs(dv), fabs(de))));
                gm[i] += DT * ye[i];
                gm[8 + i] += yp[i];
                gm[14 + i] += ye[i];
            }
            double gT = 0.0;
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
                    const double apv = (i == j ? DT : 0.0) + 0.5 * DT * DT * J1.Fvv[i * 3 + j];
                    const double ape = 0.5 * DT * DT * J1.Fve[i * 3 + j];
                    const double avv = (i == j ? 1.0 : 0.0) + 0.5 * DT * (J1.Fvv[i * 3 + j] + sv);
                    const double ave = 0.5 * DT * (J1.Fve[i * 3 + j] + se);
                    const double bvw = 0.5 * DT * DT * J2.Fve[i * 3 + j];
                    rec[REC_LIN + i * 3 + j] = apv;
                    rec[REC_LIN + 9 + i * 3 + j] = ape;
                    rec[REC_LIN + 18 + i * 3 + j] = avv;
                    rec[REC_LIN + 27 + i * 3 + j] = ave;
                    rec[REC_LIN + 42 + i * 3 + j] = bvw;
                    gm[j] += bvw * yv[i];
                    gm[11 + j] += apv * yp[i] + avv * yv[i];
                    gm[14 + j] += ape * yp[i] + ave * yv[i];
                    sT += DT * J2.Fvv[i * 3 + j] * J1.gT[j];
                }
                const double bpt = 0.5 * DT * DT * J1.gT[i];
                const double bvt = 0.5 * DT * (J1.gT[i] + sT);
                rec[REC_LIN + 36 + i] = bpt;
                rec[REC_LIN + 39 + i] = bvt;
                gT += bpt * yp[i] + bvt * yv[i];
            }
            gm[3] += gT;
        }
#pragma unroll
        for (int i = 0; i < NZ; i++) stg[i * NP + k] = gm[i];
    }
    // ---- part 2: all 64 lanes, lane = (half, stage k); rows handled in pairs
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N;
    // corridor rows: sums over the faces of a stage (pos entries 8..10 only)
    {
        double gp0 = 0, gp1 = 0, gp2 = 0, fp0 = 0, fp1 = 0, fp2 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
        if (kact) {
            const double z8 = w.z[8 * NP + k], z9 = w.z[9 * NP + k], z10 = w.z[10 * NP + k];
            for (int j = half; j < nfk; j += H) {
                const double a0 = w.face[(3 * j) * NP + k], a1 = w.face[(3 * j + 1) * NP + k], a2 = w.face[(3 * j + 2) * NP + k];
                const double hj = a0 * z8 + a1 * z9 + a2 * z10 - w.face[(3 * MF + j) * NP + k] - HU;
                const double sc = w.s[(34 + j) * NP + k], lc = w.lam[(34 + j) * NP + k];
                const double rc = hj + sc;
                l_in = fmax(l_in, fmax(hj, fabs(rc)));
                l_rc = fmax(l_rc, sc * lc);
                l_gap += sc * lc;
                gp0 += a0 * lc; gp1 += a1 * lc; gp2 += a2 * lc;
                const double sg = lc * (1.0 / sc), t = sg * rc;
                fp0 += a0 * t; fp1 += a1 * t; fp2 += a2 * t;
                p0 += sg * a0 * a0; p1 += sg * a0 * a1; p2 += sg * a0 * a2;
                p3 += sg * a1 * a1; p4 += sg * a1 * a2; p5 += sg * a2 * a2;
            }
        }
        if (H == 2) {
            gp0 = xhalf_sum(gp0); gp1 = xhalf_sum(gp1); gp2 = xhalf_sum(gp2);
            fp0 = xhalf_sum(fp0); fp1 = xhalf_sum(fp1); fp2 = xhalf_sum(fp2);
            p0 = xhalf_sum(p0); p1 = xhalf_sum(p1); p2 = xhalf_sum(p2);
            p3 = xhalf_sum(p3); p4 = xhalf_sum(p4); p5 = xhalf_sum(p5);
        }
        if (kact && half == 0) {
            gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
            rec[REC_PHIPOS + 0] = p0; rec[REC_PHIPOS + 1] = p1; rec[REC_PHIPOS + 2] = p2;
            rec[REC_PHIPOS + 3] = p1; rec[REC_PHIPOS + 4] = p3; rec[REC_PHIPOS + 5] = p4;
            rec[REC_PHIPOS + 6] = p2; rec[REC_PHIPOS + 7] = p4; rec[REC_PHIPOS + 8] = p5;
            stg[(17 + 0) * NP + k] = gp0; stg[(17 + 1) * NP + k] = gp1; stg[(17 + 2) * NP + k] = gp2;
            stg[(17 + 3) * NP + k] = fp0; stg[(17 + 4) * NP + k] = fp1; stg[(17 + 5) * NP + k] = fp2;
        }
    }
    WSYNC();
    // bounds: residuals, barrier Hessian / gradient, finished entry by entry
    if (kact) {
        cgdouble *pk = pbase + (size_t)k * np;
        double pc[NPRE];
        pc[0] = pk[0]; pc[1] = pk[1]; pc[2] = pk[2]; pc[6] = pk[6]; pc[7] = pk[7]; pc[8] = pk[8]; pc[9] = pk[9];
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
        constexpr int R = (NZ + H - 1) / H;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
            if (H == 2 && i1 >= NZ && half) continue;
            const int i = half ? i1 : i0;
            const double hd = half ? cq.hd(i1 < NZ ? i1 : i0) : cq.hd(i0);
            const double qi = half ? cq.q(i1 < NZ ? i1 : i0) : cq.q(i0);
            const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
            const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
            const double zi = w.z[i * NP + k];
            double cg = hd * zi + qi; // cost gradient
            if (i0 < 8) cg += cq.hc() * w.z[(i < 4 ? i + 4 : i - 4) * NP + k];
            const double sl = w.s[i * NP + k], su = w.s[(17 + i) * NP + k];
            const double ll = w.lam[i * NP + k], lu = w.lam[(17 + i) * NP + k];
            const double vl = lb - zi, vu = zi - ub;
            const double rl = vl + sl, ru = vu + su;
            l_in = fmax(l_in, fmax(fmax(vl, vu), fmax(fabs(rl), fabs(ru))));
            l_rc = fmax(l_rc, fmax(sl * ll, su * lu));
            l_gap += sl * ll + su * lu;
            const double sgl = ll * (1.0 / sl), sgu = lu * (1.0 / su);
            double gi = cg + stg[i * NP + k] + lu - ll;
            double ph = cg + sgu * ru - sgl * rl;
            if (i0 + H > 8 && i0 < 11) {
                if (i >= 8 && i < 11) { gi += stg[(17 + i - 8) * NP + k]; ph += stg[(17 + 3 + i - 8) * NP + k]; }
            }
            rec[REC_PHID + i] = hd + sgl + sgu;
            rec[REC_PHI + i] = ph;
            l_rs = fmax(l_rs, fabs(gi));
        }
    }
    FULLSYNC();
    EvalOut o;
    o.eq = l_eq; o.in = l_in; o.rs = l_rs; o.rc = l_rc; o.gap = l_gap; o.obj = l_obj;
    return o;
}

// ------------------------------------------------------------------ backward Riccati sweep
// pass 0: factorisation + vector part; pass 1: vector part only (new phi).  Ends with the stage-0
// solve (ds_0 left in LDS).  Returns non-zero when a pivot block is not positive definite.
template <int NP>
__device__ __noinline__ int sweep_backward(WsView w, cgdouble *xinit, int N, int pass)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); pass = uni(pass);
    FULLSYNC(); // phase boundary: other lanes' global writes of the previous phase are visible
    const int lane = threadIdx.x;
    const int my_dst = L_AB + lin_dst(lane);
    bool fact_fail = false;
    init_ab_constants(lane);
    WSYNC();
    {
        int cur = 0; // buffer holding P_{k+1}
        cgdouble *r0 = w.rec + (size_t)(N - 1) * REC_STRIDE;
        double pre_lin = r0[lane];                                           // LIN + D
        double pre_phi = (lane < 44) ? r0[REC_PHID + lane] : 0.0;            // PhiD | PhiPos | phi
        double pre_out0 = 0.0, pre_out1 = 0.0;                               // Kb|R|Pd (pass 1)
        if (pass == 1) {
            pre_out0 = r0[REC_KB + lane];
            pre_out1 = (lane < 17) ? r0[REC_KB + 64 + lane] : 0.0;
        }
        for (int kk = N - 1; kk >= 0; kk--) {
            gdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
            const bool last = (kk == N - 1);
            sm[my_dst] = pre_lin;
            if (lane < 44) sm[L_PHI + lane] = pre_phi;
            if (pass == 1) {
                sm[L_OUT + lane] = pre_out0;               // Kb(52) R(12 of 16)
                if (lane < 17) sm[L_OUT + 64 + lane] = pre_out1; // R tail, Pd
            }
            if (kk > 0) {
                cgdouble *rn = rec - REC_STRIDE;
                pre_lin = rn[lane];
                pre_phi = (lane < 44) ? rn[REC_PHID + lane] : 0.0;
                if (pass == 1) {
                    pre_out0 = rn[REC_KB + lane];
                    pre_out1 = (lane < 17) ? rn[REC_KB + 64 + lane] : 0.0;
                }
            }
            WSYNC();
            const double *Pn = sm + (cur ? L_P1 : L_P0);
            double *Pk = sm + (cur ? L_P0 : L_P1);
            const double *AB = sm + L_AB;
            if (pass == 0 && !last) {
                // PA = Pxx [A|B] (9 x 13), Pd = P d
                for (int t = lane; t < 117; t += 64) {
                    const int i = t / 13, j = t % 13;
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < 9; l++) acc += Pn[(4 + i) * 13 + 4 + l] * AB[l * 13 + j];
                    sm[L_PA + t] = acc;
                }
                if (lane < 13) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 13; j++) acc += Pn[lane * 13 + j] * sm[L_D + j];
                    sm[L_PD + lane] = acc;
                }
                WSYNC();
            }
            // q = phi + M'(Pd + p_{k+1})
            if (lane < 17) {
                double acc = sm[L_PHIV + lane];
                if (!last) {
                    if (lane < 4) {
                        acc += sm[L_PD + lane] + sm[L_PV + lane];
#pragma unroll
                        for (int i = 0; i < 9; i++) acc += AB[i * 13 + 9 + lane] * (sm[L_PD + 4 + i] + sm[L_PV + 4 + i]);
                    } else if (lane >= 8) {
#pragma unroll
                        for (int i = 0; i < 9; i++) acc += AB[i * 13 + lane - 8] * (sm[L_PD + 4 + i] + sm[L_PV + 4 + i]);
                    }
                }
                sm[L_Q + lane] = acc;
            }
            if (pass == 0) {
                // Qxx -> Pk[4+i][4+j]; Qww -> diag; Qwx = 0
                for (int t = lane; t < 169; t += 64) {
                    const int i = t / 13, j = t % 13;
                    double acc = 0.0;
                    if (i >= 4 && j >= 4) {
                        const int ii = i - 4, jj = j - 4;
                        if (ii == jj) acc = sm[L_PHID + 8 + ii];
                        if (ii < 3 && jj < 3) acc += sm[L_PHIPOS + ii * 3 + jj];
                        if (!last) {
#pragma unroll
                            for (int l = 0; l < 9; l++) acc += AB[l * 13 + ii] * sm[L_PA + l * 13 + jj];
                        }
                    } else if (i == j) acc = sm[L_PHID + 4 + i];
                    Pk[t] = acc;
                }
                // Qus (4 x 13): Quw = hc I, Qux = T A with T = Pwx + (Pxx B)'
                if (lane < 52) {
                    const int i = lane / 13, j = lane % 13;
                    double acc = 0.0;
                    if (j < 4) acc = (i == j) ? sm[L_HC] : 0.0;
                    else if (!last) {
#pragma unroll
                        for (int l = 0; l < 9; l++) acc += (Pn[i * 13 + 4 + l] + sm[L_PA + l * 13 + 9 + i]) * AB[l * 13 + j - 4];
                    }
                    sm[L_QUS + lane] = acc;
                }
                // Quu
                if (lane < 16) {
                    const int i = lane / 4, j = lane % 4;
                    double acc = (i == j) ? sm[L_PHID + i] : 0.0;
                    if (!last) {
                        acc += Pn[i * 13 + j];
#pragma unroll
                        for (int l = 0; l < 9; l++)
                            acc += (Pn[i * 13 + 4 + l] + sm[L_PA + l * 13 + 9 + i]) * AB[l * 13 + 9 + j] + AB[l * 13 + 9 + i] * Pn[(4 + l) * 13 + j];
                    }
                    sm[L_QUU + lane] = acc;
                }
                WSYNC();
                {
                    double R[16];
                    const bool ok = spd4_inverse(sm + L_QUU, R);
                    if (!ok) fact_fail = true;
#pragma unroll
                    for (int t = 0; t < 16; t++) sm[L_R + t] = ok ? R[t] : 0.0;
                }
                WSYNC();
                // Kb = R Qus
                if (lane < 52) {
                    const int i = lane / 13, j = lane % 13;
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < 4; l++) acc += sm[L_R + i * 4 + l] * sm[L_QUS + l * 13 + j];
                    sm[L_KB + lane] = acc;
                }
                WSYNC();
                // P_k = Qss - Qus' Kb
                for (int t = lane; t < 169; t += 64) {
                    const int i = t / 13, j = t % 13;
                    double acc = Pk[t];
#pragma unroll
                    for (int l = 0; l < 4; l++) acc -= sm[L_QUS + l * 13 + i] * sm[L_KB + l * 13 + j];
                    Pk[t] = acc;
                }
            } else {
                WSYNC();
            }
            // kb = R q_u ; p_k = q_s - Kb' q_u
            double kbv = 0.0, pv = 0.0;
            if (lane < 4) {
#pragma unroll
                for (int l = 0; l < 4; l++) kbv += sm[L_R + lane * 4 + l] * sm[L_Q + l];
            } else if (lane < 17) {
                pv = sm[L_Q + lane];
#pragma unroll
                for (int l = 0; l < 4; l++) pv -= sm[L_KB + l * 13 + lane - 4] * sm[L_Q + l];
            }
            WSYNC(); // everyone has read p_{k+1} (L_PV) before it is overwritten
            if (lane < 4) sm[L_KV + lane] = kbv;
            else if (lane < 17) sm[L_PV + lane - 4] = pv;
            WSYNC();
            // stream the factors of stage kk to HBM
            if (pass == 0) {
                rec[REC_KB + lane] = sm[L_OUT + lane];
                if (lane < 34) rec[REC_KB + 64 + lane] = sm[L_OUT + 64 + lane];
            } else if (lane < 17) {
                rec[REC_KV + lane] = sm[L_KV + lane]; // kb (4) + p (13)
            }
            cur ^= 1;
        }
        // stage 0: dw = -Pww^-1 (Pwx dx + p_w), dx = xinit - x_0
        const double *P0 = sm + (cur ? L_P1 : L_P0);
        if (pass == 0) {
            double Rw[16], Pww[16];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) Pww[i * 4 + j] = P0[i * 13 + j];
            const bool ok = spd4_inverse(Pww, Rw);
            if (!ok) fact_fail = true;
            WSYNC();
#pragma unroll
            for (int t = 0; t < 16; t++) sm[L_S0 + t] = ok ? Rw[t] : 0.0;
            if (lane < 36) sm[L_S0 + 16 + lane] = P0[(lane / 9) * 13 + 4 + lane % 9];
        }
        if (lane < 9) sm[L_DS + 4 + lane] = xinit[lane] - w.z[(8 + lane) * NP + 0];
        WSYNC();
        if (lane < 4) {
            double acc = sm[L_PV + lane];
#pragma unroll
            for (int j = 0; j < 9; j++) acc += sm[L_S0 + 16 + lane * 9 + j] * sm[L_DS + 4 + j];
            sm[L_DU + lane] = acc; // temporary: rhs
        }
        WSYNC();
        if (lane < 4) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < 4; l++) acc -= sm[L_S0 + lane * 4 + l] * sm[L_DU + l];
            sm[L_DS + lane] = acc;
        }
        WSYNC();
    }

    FULLSYNC();
    return fact_fail ? 1 : 0;
}

// ------------------------------------------------------------------ forward sweep: dz for all stages
template <int NP>
__device__ __noinline__ void sweep_forward(WsView w, int N)
{
    w = uni(w); N = uni(N);
    FULLSYNC(); // phase boundary: other lanes' global writes of the previous phase are visible
    const int lane = threadIdx.x;
    const int my_dst = L_AB + lin_dst(lane);
    init_ab_constants(lane);
    WSYNC();
    {
        cgdouble *r0 = w.rec;
        double pre_lin = r0[lane];
        double pre_kb = (lane < 52) ? r0[REC_KB + lane] : ((lane < 56) ? r0[REC_KV + lane - 52] : 0.0);
        for (int kk = 0; kk < N; kk++) {
            cgdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
            sm[my_dst] = pre_lin;
            if (lane < 52) sm[L_KB + lane] = pre_kb;
            else if (lane < 56) sm[L_KV + lane - 52] = pre_kb;
            if (kk < N - 1) {
                cgdouble *rn = rec + REC_STRIDE;
                pre_lin = rn[lane];
                pre_kb = (lane < 52) ? rn[REC_KB + lane] : ((lane < 56) ? rn[REC_KV + lane - 52] : 0.0);
            }
            WSYNC();
            if (lane < 4) {
                double a0 = sm[L_KV + lane], a1 = 0.0;
#pragma unroll
                for (int j = 0; j < 12; j += 2) {
                    a0 += sm[L_KB + lane * 13 + j] * sm[L_DS + j];
                    a1 += sm[L_KB + lane * 13 + j + 1] * sm[L_DS + j + 1];
                }
                a0 += sm[L_KB + lane * 13 + 12] * sm[L_DS + 12];
                const double du = -(a0 + a1);
                sm[L_DU + lane] = du;
                w.dz[lane * NP + kk] = du;
            } else if (lane < 17) {
                w.dz[lane * NP + kk] = sm[L_DS + lane - 4];
            }
            WSYNC();
            if (kk < N - 1) {
                double v = 0.0;
                if (lane < 4) v = sm[L_DU + lane] + sm[L_D + lane];
                else if (lane < 13) {
                    const int i = lane - 4;
                    double a0 = sm[L_D + lane], a1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        a0 += sm[L_AB + i * 13 + j] * sm[L_DS + 4 + j];
                        a1 += sm[L_AB + i * 13 + j + 1] * sm[L_DS + 4 + j + 1];
                    }
                    a0 += sm[L_AB + i * 13 + 8] * sm[L_DS + 12];
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        a0 += sm[L_AB + i * 13 + 9 + j] * sm[L_DU + j];
                        a1 += sm[L_AB + i * 13 + 10 + j] * sm[L_DU + j + 1];
                    }
                    v = a0 + a1;
                }
                WSYNC();
                if (lane < 13) sm[L_DS + lane] = v;
            }
        }
                WSYNC();
    }
}

struct SlackOut {
    double ap, ad, sigma, smu;
};

// ------------------------------------------------------------------ slack / multiplier steps
// All 64 lanes, lane = (half, stage k), constraints handled in pairs (flattened [row][stage] arrays).
// pass 0 (affine): step lengths, mu_aff -> sigma, second-order term, corrector rhs phi -> record
// pass 1 (corrector): fraction-to-boundary step lengths, update z, s, lambda
// Per constraint:  ds = -(G z - g + s) - G dz,  dl = (-(s l - smu + corr) - l ds) / s.
// Ratios -ds/s and -dl/l are formed with ONE reciprocal u = 1/(s l) per constraint.
template <int NP>
__device__ __noinline__ SlackOut phase_slack(WsView w, cgdouble *pbase, int np, int N, int MF, int nfk, int model, int pass,
                                             double smu, double mu, int mtot, double ftb, double tol_comp)
{
    w = uni(w); pbase = uni(pbase); np = uni(np); N = uni(N); MF = uni(MF); model = uni(model); pass = uni(pass);
    smu = uni(smu); mu = uni(mu); mtot = uni(mtot); ftb = uni(ftb); tol_comp = uni(tol_comp);
    FULLSYNC(); // phase boundary: dz of the forward sweep is visible
    constexpr int H = 64 / NP;
    constexpr int R = (NZ + H - 1) / H;
    const int lane = threadIdx.x;
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N;
    double *stg = stage_area<NP>();
    double sigma = 0.0;

    // one constraint: returns ds, dl and the two ratios
    auto cstep = [&](int c, double gdz, double viol, double &ds, double &dl, double &rp, double &rd, double &s, double &l) {
        s = w.s[c * NP + k];
        l = w.lam[c * NP + k];
        const double u = 1.0 / (s * l);
        const double sinv = u * l, linv = u * s;
        ds = -(viol + s) - gdz;
        const double rc = s * l - smu + (pass ? w.corr[c * NP + k] : 0.0);
        dl = (-rc - l * ds) * sinv;
        rp = -ds * sinv;
        rd = -dl * linv;
    };

    // ---- A: largest ratios -> step lengths
    double m_p = 0.0, m_d = 0.0;
    double z8 = 0, z9 = 0, z10 = 0, d8 = 0, d9 = 0, d10 = 0;
    if (kact) {
        z8 = w.z[8 * NP + k]; z9 = w.z[9 * NP + k]; z10 = w.z[10 * NP + k];
        d8 = w.dz[8 * NP + k]; d9 = w.dz[9 * NP + k]; d10 = w.dz[10 * NP + k];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
            if (H == 2 && i1 >= NZ && half) continue;
            const int i = half ? i1 : i0;
            const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
            const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
            const double zi = w.z[i * NP + k], dzi = w.dz[i * NP + k];
            double ds, dl, rp, rd, sv, lv;
            cstep(i, -dzi, lb - zi, ds, dl, rp, rd, sv, lv);
            m_p = fmax(m_p, rp); m_d = fmax(m_d, rd);
            cstep(17 + i, dzi, zi - ub, ds, dl, rp, rd, sv, lv);
            m_p = fmax(m_p, rp); m_d = fmax(m_d, rd);
        }
        for (int j = half; j < nfk; j += H) {
            const double a0 = w.face[(3 * j) * NP + k], a1 = w.face[(3 * j + 1) * NP + k], a2 = w.face[(3 * j + 2) * NP + k];
            double ds, dl, rp, rd, sv, lv;
            cstep(34 + j, a0 * d8 + a1 * d9 + a2 * d10, a0 * z8 + a1 * z9 + a2 * z10 - w.face[(3 * MF + j) * NP + k] - HU,
                  ds, dl, rp, rd, sv, lv);
            m_p = fmax(m_p, rp); m_d = fmax(m_d, rd);
        }
    }
    m_p = wave_max(m_p); m_d = wave_max(m_d);
    const double lim = pass ? ftb : 1.0;
    const double ap = (m_p > lim) ? lim / m_p : 1.0;
    const double ad = (m_d > lim) ? lim / m_d : 1.0;

    // ---- B: corridor rows: affine complementarity + second-order term (pass 0) / update (pass 1)
    double l_gapaff = 0.0;
    if (pass == 0) {
        if (kact) {
            for (int j = half; j < nfk; j += H) {
                const double a0 = w.face[(3 * j) * NP + k], a1 = w.face[(3 * j + 1) * NP + k], a2 = w.face[(3 * j + 2) * NP + k];
                double ds, dl, rp, rd, sv, lv;
                cstep(34 + j, a0 * d8 + a1 * d9 + a2 * d10, a0 * z8 + a1 * z9 + a2 * z10 - w.face[(3 * MF + j) * NP + k] - HU,
                      ds, dl, rp, rd, sv, lv);
                l_gapaff += (sv + ap * ds) * (lv + ad * dl);
                w.corr[(34 + j) * NP + k] = ds * dl;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
                if (H == 2 && i1 >= NZ && half) continue;
                const int i = half ? i1 : i0;
                const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
                const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
                const double zi = w.z[i * NP + k], dzi = w.dz[i * NP + k];
                double ds, dl, rp, rd, sv, lv;
                cstep(i, -dzi, lb - zi, ds, dl, rp, rd, sv, lv);
                l_gapaff += (sv + ap * ds) * (lv + ad * dl);
                w.corr[i * NP + k] = ds * dl;
                cstep(17 + i, dzi, zi - ub, ds, dl, rp, rd, sv, lv);
                l_gapaff += (sv + ap * ds) * (lv + ad * dl);
                w.corr[(17 + i) * NP + k] = ds * dl;
            }
        }
        const double mu_aff = wave_sum(l_gapaff) / (double)mtot;
        sigma = mu_aff / mu;
        sigma = sigma * sigma * sigma;
        if (sigma > 1.0) sigma = 1.0;
        smu = sigma * mu;
        if (smu < MU_FLOOR_FRAC * tol_comp) smu = MU_FLOOR_FRAC * tol_comp;
        FULLSYNC(); // corr written above is re-read below by the same lanes; also orders LDS staging
        // corrector rhs: phi = grad f + G'(Sigma r_in + (smu - corr)/s)
        {
            double fp0 = 0, fp1 = 0, fp2 = 0;
            if (kact) {
                for (int j = half; j < nfk; j += H) {
                    const double a0 = w.face[(3 * j) * NP + k], a1 = w.face[(3 * j + 1) * NP + k], a2 = w.face[(3 * j + 2) * NP + k];
                    const double hj = a0 * z8 + a1 * z9 + a2 * z10 - w.face[(3 * MF + j) * NP + k] - HU;
                    const double sc = w.s[(34 + j) * NP + k], lc = w.lam[(34 + j) * NP + k];
                    const double t = (lc * (hj + sc) + smu - w.corr[(34 + j) * NP + k]) * (1.0 / sc);
                    fp0 += a0 * t; fp1 += a1 * t; fp2 += a2 * t;
                }
            }
            if (H == 2) { fp0 = xhalf_sum(fp0); fp1 = xhalf_sum(fp1); fp2 = xhalf_sum(fp2); }
            if (kact && half == 0) { stg[0 * NP + k] = fp0; stg[1 * NP + k] = fp1; stg[2 * NP + k] = fp2; }
        }
        WSYNC();
        if (kact) {
            cgdouble *pk = pbase + (size_t)k * np;
            double pc[NPRE];
            pc[0] = pk[0]; pc[1] = pk[1]; pc[2] = pk[2]; pc[6] = pk[6]; pc[7] = pk[7]; pc[8] = pk[8]; pc[9] = pk[9];
            const CostQ cq = make_cost(pc, stage_class(k, N), model);
            gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
                if (H == 2 && i1 >= NZ && half) continue;
                const int i = half ? i1 : i0;
                const double hd = half ? cq.hd(i1 < NZ ? i1 : i0) : cq.hd(i0);
                const double qi = half ? cq.q(i1 < NZ ? i1 : i0) : cq.q(i0);
                const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
                const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
                const double zi = w.z[i * NP + k];
                double ph = hd * zi + qi;
                if (i0 < 8) ph += cq.hc() * w.z[(i < 4 ? i + 4 : i - 4) * NP + k];
                const double sl = w.s[i * NP + k], su = w.s[(17 + i) * NP + k];
                const double ll = w.lam[i * NP + k], lu = w.lam[(17 + i) * NP + k];
                const double rl = lb - zi + sl, ru = zi - ub + su;
                const double tl = (ll * rl + smu - w.corr[i * NP + k]) * (1.0 / sl);
                const double tu = (lu * ru + smu - w.corr[(17 + i) * NP + k]) * (1.0 / su);
                ph += tu - tl;
                if (i0 + H > 8 && i0 < 11) {
                    if (i >= 8 && i < 11) ph += stg[(i - 8) * NP + k];
                }
                rec[REC_PHI + i] = ph;
            }
        }
    } else if (kact) {
        for (int j = half; j < nfk; j += H) {
            const double a0 = w.face[(3 * j) * NP + k], a1 = w.face[(3 * j + 1) * NP + k], a2 = w.face[(3 * j + 2) * NP + k];
            double ds, dl, rp, rd, sv, lv;
            cstep(34 + j, a0 * d8 + a1 * d9 + a2 * d10, a0 * z8 + a1 * z9 + a2 * z10 - w.face[(3 * MF + j) * NP + k] - HU,
                  ds, dl, rp, rd, sv, lv);
            w.s[(34 + j) * NP + k] = sv + ap * ds;
            w.lam[(34 + j) * NP + k] = lv + ad * dl;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
            if (H == 2 && i1 >= NZ && half) continue;
            const int i = half ? i1 : i0;
            const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
            const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
            const double zi = w.z[i * NP + k], dzi = w.dz[i * NP + k];
            double ds, dl, rp, rd, sv, lv;
            cstep(i, -dzi, lb - zi, ds, dl, rp, rd, sv, lv);
            w.s[i * NP + k] = sv + ap * ds;
            w.lam[i * NP + k] = lv + ad * dl;
            cstep(17 + i, dzi, zi - ub, ds, dl, rp, rd, sv, lv);
            w.s[(17 + i) * NP + k] = sv + ap * ds;
            w.lam[(17 + i) * NP + k] = lv + ad * dl;
            w.z[i * NP + k] = zi + ap * dzi;
        }
    }
    FULLSYNC();
    SlackOut o;
    o.ap = ap; o.ad = ad; o.sigma = sigma; o.smu = smu;
    return o;
}

// ------------------------------------------------------------------ costate sweep: y <- y + ap (y+ - y)
// y+_k = (Phi_k dz_k + phi_k)_s + [0; A_k' y+_{k+1,x}]
template <int NP>
__device__ __noinline__ void sweep_costate(WsView w, cgdouble *pk, int N, double ap)
{
    w = uni(w); N = uni(N); ap = uni(ap);
    FULLSYNC(); // phase boundary: other lanes' global writes of the previous phase are visible
    const int lane = threadIdx.x, k = lane;
    const bool act = lane < N;
    const int my_dst = L_AB + lin_dst(lane);
    const double hc_k = act ? -2.0 * pk[8] : 0.0;
    init_ab_constants(lane);
    WSYNC();
    {
        // w-part is stage-parallel
        if (act) {
            cgdouble *rec = w.rec + (size_t)k * REC_STRIDE;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double yw = rec[REC_PHID + 4 + i] * w.dz[(4 + i) * NP + k] + hc_k * w.dz[i * NP + k] + rec[REC_PHI + 4 + i];
                const double yo = w.y[i * NP + k];
                w.y[i * NP + k] = yo + ap * (yw - yo);
            }
        }
        int cy = 0;
        cgdouble *r0 = w.rec + (size_t)(N - 1) * REC_STRIDE;
        double pre_lin = r0[lane];
        double pre_phi = (lane < 44) ? r0[REC_PHID + lane] : 0.0;
        double pre_dz = (lane < 9) ? w.dz[(8 + lane) * NP + N - 1] : 0.0;
        for (int kk = N - 1; kk >= 0; kk--) {
            sm[my_dst] = pre_lin;
            if (lane < 44) sm[L_PHI + lane] = pre_phi;
            if (lane < 9) sm[L_DS + 4 + lane] = pre_dz;
            if (kk > 0) {
                cgdouble *rn = w.rec + (size_t)(kk - 1) * REC_STRIDE;
                pre_lin = rn[lane];
                pre_phi = (lane < 44) ? rn[REC_PHID + lane] : 0.0;
                pre_dz = (lane < 9) ? w.dz[(8 + lane) * NP + kk - 1] : 0.0;
            }
            WSYNC();
            if (lane < 9) {
                double acc = sm[L_PHID + 8 + lane] * sm[L_DS + 4 + lane] + sm[L_PHIV + 8 + lane];
                if (lane < 3) {
#pragma unroll
                    for (int j = 0; j < 3; j++) acc += sm[L_PHIPOS + lane * 3 + j] * sm[L_DS + 4 + j];
                }
                if (kk < N - 1) {
                    const double *yn = sm + L_YX + (cy ? 9 : 0);
#pragma unroll
                    for (int i = 0; i < 9; i++) acc += sm[L_AB + i * 13 + lane] * yn[i];
                }
                sm[L_YX + (cy ? 0 : 9) + lane] = acc;
                const double yo = w.y[(4 + lane) * NP + kk];
                w.y[(4 + lane) * NP + kk] = yo + ap * (acc - yo);
            }
            cy ^= 1;
            WSYNC();
        }
                WSYNC();
    }
}

// ------------------------------------------------------------------ the solver kernel
template <int NP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRP_WAVES_PER_EU, FRP_WAVES_PER_EU))) void nmpc_ipm_kernel(KernelArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N, M = a.M, MF = a.MF, np = NPRE + 4 * M;
    const int mcf = 34 + MF;
    const bool act = lane < N; // lane == stage in the stage-parallel phases
    const int k = lane;

    WsView w;
    {
        gdouble *base = (gdouble *)(a.ws + (size_t)b * ws_doubles_per_problem(N, MF));
        w.rec = base;
        w.z = w.rec + (size_t)N * REC_STRIDE;
        w.y = w.z + 17 * NP;
        w.dz = w.y + 13 * NP;
        w.s = w.dz + 17 * NP;
        w.lam = w.s + (size_t)mcf * NP;
        w.corr = w.lam + (size_t)mcf * NP;
        w.face = w.corr + (size_t)mcf * NP;
    }
    cgdouble *xinit = (cgdouble *)(a.xinit + (size_t)b * 9);
    cgdouble *pk = (cgdouble *)(a.params + ((size_t)b * N + (act ? k : 0)) * np);

    // ---------------------------------------------------------------- init (lane == stage)
    int nf = 0;
    int bad_param = 0;
    double smin = 1e300;
    if (act) {
        if (a.nfaces) nf = a.nfaces[(size_t)b * N + k];
        else { // trailing all-zero rows are padding (forces_normal.cpp:127-135)
            nf = M;
            while (nf > 0) {
                cgdouble *r = pk + NPRE + 3 * (nf - 1);
                if (r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0 && pk[NPRE + 3 * M + nf - 1] >= -HU) nf--;
                else break;
            }
        }
        if (nf > MF || nf < 0) { bad_param = 1; nf = 0; }
        double zk[NZ];
        const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
#pragma unroll
        for (int i = 0; i < NZ; i++) {
            zk[i] = z0[i];
            w.z[i * NP + k] = zk[i];
        }
#pragma unroll
        for (int i = 0; i < NZ; i++) {
            const double sl = zk[i] - lower_bound(i), su = upper_bound(i) - zk[i];
            w.s[i * NP + k] = sl;
            w.s[(17 + i) * NP + k] = su;
            smin = fmin(smin, fmin(sl, su));
        }
        for (int j = 0; j < nf; j++) {
            const double a0 = pk[NPRE + 3 * j], a1 = pk[NPRE + 3 * j + 1], a2 = pk[NPRE + 3 * j + 2];
            const double bj = pk[NPRE + 3 * M + j];
            w.face[(3 * j) * NP + k] = a0;
            w.face[(3 * j + 1) * NP + k] = a1;
            w.face[(3 * j + 2) * NP + k] = a2;
            w.face[(3 * MF + j) * NP + k] = bj;
            const double sc = -(a0 * zk[8] + a1 * zk[9] + a2 * zk[10] - bj - HU);
            w.s[(34 + j) * NP + k] = sc;
            smin = fmin(smin, sc);
        }
#pragma unroll
        for (int i = 0; i < NS; i++) w.y[i * NP + k] = 0.0;
        w.rec[(size_t)k * REC_STRIDE + REC_HC] = -2.0 * pk[8]; // hc of this stage's cost (constant)
    }
    smin = wave_min(smin);
    const int mtot = (int)wave_sum(act ? (double)(34 + nf) : 0.0);
    if (wave_max((double)bad_param) > 0.0) {
        if (lane == 0) { a.exitflag[b] = FRP_EXIT_PARAM_VALUE; a.iters[b] = 0; }
        if (act) {
            const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
            for (int i = 0; i < NZ; i++) a.z[((size_t)b * N + k) * NZ + i] = z0[i];
        }
        return;
    }
    {
        // infeasible-start initialisation: uniform slack shift (see oracle/nmpc_ipm.c)
        const double shift = (smin >= S_MIN) ? 0.0 : (S_MIN - smin) + fmax(0.0, -smin);
        if (act) {
            for (int i = 0; i < 34 + nf; i++) {
                const double s = w.s[i * NP + k] + shift;
                w.s[i * NP + k] = s;
                w.lam[i * NP + k] = a.mu0 / s;
            }
        }
    }
    cgdouble *pbase = (cgdouble *)(a.params + (size_t)b * N * np);
    const int nfk = __shfl(nf, lane % NP); // face count of stage k = lane % NP for the (half, stage) lane mapping
    FULLSYNC();

    int flag = FRP_EXIT_MAXIT, it = 0;
    double res_eq = 0, res_in = 0, rs = 0, rcomp = 0, pobj = 0, mu = 0, sigma = 0, step_cc = 0;

#ifdef FRP_PROFILE
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tc0, tc1;
#define TICK() tc0 = clock64()
#define TOCK(i) do { tc1 = clock64(); tph[i] += tc1 - tc0; tc0 = tc1; } while (0)
#else
#define TICK()
#define TOCK(i)
#endif
    for (it = 0;; it++) {
        TICK();
        const EvalOut e = phase_eval<NP>(w, pbase, np, xinit, N, MF, nfk, a.model);
        res_eq = wave_max(e.eq); res_in = wave_max(e.in); rs = wave_max(e.rs); rcomp = wave_max(e.rc);
        pobj = wave_sum(e.obj);
        mu = wave_sum(e.gap) / (double)mtot;
        if (!(res_eq == res_eq) || !(rs == rs) || !(pobj == pobj)) { flag = FRP_EXIT_BADFUNCEVAL; break; }
        if (res_eq <= a.tol_eq && res_in <= a.tol_ineq && rs <= a.tol_stat && rcomp <= a.tol_comp) { flag = FRP_EXIT_OPTIMAL; break; }
        if (it >= a.maxit) { flag = FRP_EXIT_MAXIT; break; }
        if (mu > DIVERGE_MU * fmax(1.0, a.mu0) || rs > DIVERGE_RS) { flag = FRP_EXIT_NOPROGRESS; break; }
        WSYNC();
        TOCK(0);

        // predictor (affine) solve
        if (sweep_backward<NP>(w, xinit, N, 0)) { flag = FRP_EXIT_FACTORIZATION; break; }
        TOCK(1);
        sweep_forward<NP>(w, N);
        TOCK(2);
        const SlackOut s0 = phase_slack<NP>(w, pbase, np, N, MF, nfk, a.model, 0, 0.0, mu, mtot, a.ftb, a.tol_comp);
        sigma = s0.sigma;
        TOCK(3);
        // corrector solve (same factorisation, new rhs)
        sweep_backward<NP>(w, xinit, N, 1);
        TOCK(4);
        sweep_forward<NP>(w, N);
        TOCK(2);
        const SlackOut s1 = phase_slack<NP>(w, pbase, np, N, MF, nfk, a.model, 1, s0.smu, mu, mtot, a.ftb, a.tol_comp);
        step_cc = s1.ap;
        TOCK(3);
        sweep_costate<NP>(w, pk, N, s1.ap);
        TOCK(5);
    }

    // ---------------------------------------------------------------- outputs
    if (act) {
        double *zo = a.z + ((size_t)b * N + k) * NZ;
#pragma unroll
        for (int i = 0; i < NZ; i++) zo[i] = w.z[i * NP + k];
    }
    if (lane == 0) {
        a.exitflag[b] = flag;
        a.iters[b] = it;
        if (a.info) {
            double *o = a.info + (size_t)b * FRP_INFO_STRIDE;
            o[0] = res_eq; o[1] = res_in; o[2] = rs; o[3] = rcomp; o[4] = pobj; o[5] = mu; o[6] = step_cc; o[7] = sigma;
#ifdef FRP_PROFILE
            for (int i = 0; i < 6; i++) o[i] = (double)tph[i]; // cycles: eval, factor, forward(x2), slack(x2), backvec, costate
#endif
        }
    }
}

// ------------------------------------------------------------------ batched model callback
// One thread per (problem, stage): the reference's extfunc for B*N stage points (casadi2forces.c:42-245).
__global__ __launch_bounds__(256) void stage_eval_kernel(int B, int N, int M, int model, const double *__restrict__ z,
                                                          const double *__restrict__ params, double *__restrict__ f,
                                                          double *__restrict__ gf, double *__restrict__ c,
                                                          double *__restrict__ Jc, double *__restrict__ h)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * N) return;
    const int k = (int)(t % N);
    const int np = NPRE + 4 * M;
    const double *zk = z + t * NZ, *pk = params + t * np;
    double zl[NZ], p10[NPRE];
#pragma unroll
    for (int i = 0; i < NZ; i++) zl[i] = zk[i];
#pragma unroll
    for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
    const int sc = stage_class(k, N);
    if (f || gf) {
        double g[NZ];
        const double cost = stage_cost(zl, p10, sc, model, g);
        if (f) f[t] = cost;
        if (gf) {
#pragma unroll
            for (int i = 0; i < NZ; i++) gf[t * NZ + i] = g[i];
        }
    }
    if (c || Jc) {
        if (sc != STAGE_LAST) {
            Lin L;
            double xn[9];
            rk2<true>(zl + 8, zl, p10 + 3, xn, &L);
            if (c) {
#pragma unroll
                for (int i = 0; i < 9; i++) c[t * 13 + i] = xn[i];
#pragma unroll
                for (int i = 0; i < 4; i++) c[t * 13 + 9 + i] = zl[i];
            }
            if (Jc) {
                double *J = Jc + t * 221;
                const double *Lc = reinterpret_cast<const double *>(&L);
                for (int col = 0; col < 17; col++) {
#pragma unroll
                    for (int row = 0; row < 13; row++) {
                        double v = 0.0;
                        if (row < 9) {
                            if (col < 4) v = lin_B(Lc, row, col);
                            else if (col >= 8) v = lin_A(Lc, row, col - 8);
                        } else if (col == row - 9) v = 1.0;
                        J[col * 13 + row] = v;
                    }
                }
            }
        } else {
            if (c) for (int i = 0; i < 13; i++) c[t * 13 + i] = 0.0;
            if (Jc) for (int i = 0; i < 221; i++) Jc[t * 221 + i] = 0.0;
        }
    }
    if (h) {
        const double *A = pk + NPRE, *bb = pk + NPRE + 3 * M;
        for (int j = 0; j < M; j++) h[t * M + j] = A[3 * j] * zl[8] + A[3 * j + 1] * zl[9] + A[3 * j + 2] * zl[10] - bb[j];
    }
}

// ------------------------------------------------------------------ launchers
size_t ws_bytes(int B, int N, int MF) { return (size_t)B * ws_doubles_per_problem(N, MF) * sizeof(double); }

hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream)
{
    if (padded_stages(a.N) == 32) hipLaunchKernelGGL(nmpc_ipm_kernel<32>, dim3(a.B), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(nmpc_ipm_kernel<64>, dim3(a.B), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream)
{
    const size_t total = (size_t)B * N;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(stage_eval_kernel, dim3(blocks), dim3(256), 0, stream, B, N, M, model, z, params, f, gf, c, Jc, h);
    return hipGetLastError();
}

} // namespace frp
