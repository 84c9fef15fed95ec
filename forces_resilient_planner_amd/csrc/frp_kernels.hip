// frp_kernels.hip -- gfx950 kernels of the batched NMPC solver (hand-written HIP, FP64).
//
// nmpc_ipm_kernel: one 64-lane wavefront == one workgroup == one NMPC problem, resident for the whole
// interior-point solve (no host round trips, per-problem early exit; the hardware dispatcher
// refills the slot with the next problem).  It replaces the reference's closed NLP solver
//   FORCESNLPsolver_{normal,final}_solve  (FORCESNLPsolver_normal.h:323; forces_normal.cpp:139)
// and its model callback FORCESNLPsolver_*_casadi2forces (casadi2forces.c:42-245).
//
// Iteration (same as the CPU oracle so both can be compared iterate by iterate):
//   primal-dual interior point, Mehrotra predictor-corrector, stage Hessian = exact cost Hessian +
//   exact Hessian of the RK2 dynamics (Gauss-Newton fallback when the reduced Hessian is indefinite),
//   Newton KKT system solved by a Riccati recursion over the stage chain with
//   state s = [w; x] (13) and control u (4):   s_{k+1} = [u_k; A_k x_k + B_k u_k] + d_k.
//
// Work distribution inside the wavefront
//   * element-wise phases (residuals, barrier terms, step lengths, updates): all 64 lanes,
//     lane = (row pair, stage), operands in [row][stage] arrays -> coalesced 64-lane accesses;
//   * model evaluation (RK2 step, Jacobian, exact Hessian): lane == stage;
//   * the serial Riccati sweeps: every 13x13(+1) block lives in REGISTERS as a 16x16 FP64 tile in the
//     v_mfma_f64_16x16x4_f64 accumulator layout (lane (g,c), register r <-> element [4r+g][c]).  In that
//     layout D = X'Y is four MFMAs with A := X, B := Y register for register, so the whole recursion
//       X = P M,  G = M'X + Phi,  T = R G_u,  S = G - G_u' T,  P <- S (+ w blocks)
//     runs on the matrix pipe with no cross-lane data movement; the right-hand side rides along as
//     column 13 of the tiles.  Per stage the sweeps stream one 64-lane record row from/to HBM.
#include <hip/hip_runtime.h>
#include <math.h>
#include "frp_model.hpp"
#include "../../include/frp_nmpc.h"
#include "frp_kernels.h"

namespace frp {

// occupancy target: waves per SIMD the register allocator must leave room for (propagates to the
// non-inlined phase functions)
#ifndef FRP_WAVES_PER_EU
#define FRP_WAVES_PER_EU 2
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
// sum over the 16 lanes of one row group (lanes with equal lane >> 4)
__device__ __forceinline__ double row16_sum(double v)
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double lane_bcast(double v, int src) // wave-uniform src lane
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
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
// [0, 736): staging.  Element-wise phases: gm[17][NP] + corridor sums[6][NP] (NP = 32).
//           Riccati sweeps: the E part of the current stage record + constants + T' + R.
// [736, ...): fields that live across phases of one iteration.
constexpr int S_E = 0;                    // E part of the stage record (248)
constexpr int S_ZERO = 248, S_ONE = 249, S_DTC = 250;
constexpr int S_T = 256;                  // T' (64)
constexpr int S_R = 320;                  // 4x4 inverse handed from uniform registers to lanes (16)
constexpr int S_STAGING = 23 * 32;        // 736
constexpr int S_RW = S_STAGING;           // stage-0 solve: Pww^-1 (16)
constexpr int S_PWX = S_RW + 16;          // stage-0 solve: Pwx (4 x 9)
constexpr int S_DS0 = S_PWX + 36;         // ds_0 = [dw_0; dx_0] (13, padded 16)
constexpr int L_TOTAL = S_DS0 + 16;

// symmetric positive definite 4x4 inverse via LDL'; returns false if a pivot is not positive
__device__ __forceinline__ bool spd4_inverse(const double *a /*row-major 4x4, lower part used*/, double *r /*16*/)
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
    gdouble *rec, *z, *y, *dz, *s, *lam, *corr, *face, *step;
};

__host__ __device__ inline int padded_stages(int N) { return N <= 32 ? 32 : 64; }

__host__ __device__ inline size_t ws_doubles_per_problem(int N, int MF)
{
    const size_t mcf = 34 + MF, NPs = padded_stages(N);
    return (size_t)N * REC_STRIDE + NPs * (17 + Y_ROWS + DZ_ROWS + 5 * mcf + 4 * (size_t)MF);
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
    w.s = uni(w.s); w.lam = uni(w.lam); w.corr = uni(w.corr); w.face = uni(w.face); w.step = uni(w.step);
    return w;
}

__shared__ double sm[L_TOTAL];
// Newton step dz = [du(4); ds(13); 3 pad rows][NP]: written by the forward sweep, read by the step phases and the
// costate sweep -- kept in LDS so that the sweeps carry no global stores for it (a store in the loop makes the
// staging write of the next stage wait for vmcnt(0))
__shared__ double sm_dz32[DZ_ROWS * 32];
__shared__ double sm_dz64[DZ_ROWS * 64];
template <int NP>
__device__ __forceinline__ double *dz_area() { return NP == 32 ? sm_dz32 : sm_dz64; }

#ifdef FRP_PROFILE
__device__ long long g_prof[24];
// per-function segment timers kept in registers, flushed once at the end of the function
#define PROF_BEGIN() long long pacc_[6] = {0, 0, 0, 0, 0, 0}; long long pts_ = clock64()
#define PROF_SEG(i) do { const long long tn_ = clock64(); pacc_[(i) % 6] += tn_ - pts_; pts_ = tn_; } while (0)
#define PROF_END(base) do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 6; q_++) g_prof[(base) + q_] += pacc_[q_]; } while (0)
#else
#define PROF_BEGIN()
#define PROF_SEG(i)
#define PROF_END(base)
#endif

// ------------------------------------------------------------------ 16x16 FP64 tiles in registers
// Tile X: lane l = 16 g + c holds x[r] = X[4r + g][c], r = 0..3 (the C/D layout of
// v_mfma_f64_16x16x4_f64; A operand of slice s = X'[.., 4s+g] i.e. again x[s], B operand = x[s]).
typedef double d4 __attribute__((ext_vector_type(4)));

// D = X' Y + C
__device__ __forceinline__ d4 mm_tn(const d4 x, const d4 y, d4 c)
{
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], c, 0, 0, 0);
    return c;
}
// D = X[0:4,:]' Y[0:4,:] + C  (only the first four rows of X and Y contribute)
__device__ __forceinline__ d4 mm_tn4(double x0, double y0, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, c, 0, 0, 0); }

// LDS offset (within the staged E record) of Mt[row][col], the augmented transition matrix
//   rows: s+ = [w+(0..3); x+(4..12)],  cols: [u(0..3); x(4..12); 13 = d]
//   w+ = u + d_w,  x+ = A x + B u + d_x
__device__ __forceinline__ int m_src(int row, int col)
{
    if (row > 12 || col > 13) return S_ZERO;
    if (col == 13) return S_E + REC_D + row;
    if (row < 4) return (col == row) ? S_ONE : S_ZERO;
    const int i = row - 4, bi = i / 3, ii = i % 3;
    if (col < 4) { // B[i][col]
        if (col == 3) return bi == 0 ? S_E + REC_LIN + 36 + ii : (bi == 1 ? S_E + REC_LIN + 39 + ii : S_ZERO);
        if (bi == 1) return S_E + REC_LIN + 42 + ii * 3 + col;
        if (bi == 2) return ii == col ? S_DTC : S_ZERO;
        return S_ZERO;
    }
    const int j = col - 4, bj = j / 3, jj = j % 3;
    if (bi == 0) return bj == 0 ? (ii == jj ? S_ONE : S_ZERO) : S_E + REC_LIN + (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? S_ZERO : S_E + REC_LIN + (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? S_ONE : S_ZERO;
}
// z index of tile index a over (u, x):  u -> 0..3, x -> 8..16
__device__ __forceinline__ int zi_of(int a) { return a < 4 ? a : a + 4; }
// index into the 10 x 10 dynamics Hessian (rates, T, v, e) of tile index a, or -1
__device__ __forceinline__ int hidx_of(int a) { return a < 4 ? a : (a >= 7 && a <= 12 ? a - 3 : -1); }
// the three LDS sources summed into C~[row][col] = Phi~ (diag + corridor + Hessian) with phi in column 13
__device__ __forceinline__ void c_src(int row, int col, int &o1, int &o2, int &o3)
{
    o1 = o2 = o3 = S_ZERO;
    if (row > 12) return;
    if (col == 13) { o1 = S_E + REC_PHI + zi_of(row); return; }
    if (col > 12) return;
    if (col == row) o1 = S_E + REC_PHID + zi_of(row);
    if (row >= 4 && row <= 6 && col >= 4 && col <= 6) o2 = S_E + REC_PHIPOS + (row - 4) * 3 + (col - 4);
    const int hr = hidx_of(row), hc_ = hidx_of(col);
    if (hr >= 0 && hc_ >= 0) o3 = S_E + REC_HD + hr * 10 + hc_;
}

__device__ __forceinline__ void init_stage_constants(int lane)
{
    if (lane == 0) { sm[S_ZERO] = 0.0; sm[S_ONE] = 1.0; sm[S_DTC] = DT; }
}

struct EvalOut {
    double eq, in, rs, rc, gap, obj;
};

// Staging area of the element-wise phases (aliases the sweep staging, which is dead then):
// gm[17][NP] multiplier part of the stationarity residual, gf[6][NP] corridor sums for pos entries.
__shared__ double sm_big[23 * 64]; // only referenced (hence only allocated) by the NP = 64 instantiation
template <int NP>
__device__ __forceinline__ double *stage_area() { return NP == 32 ? sm : sm_big; }

__device__ __forceinline__ double xhalf_sum(double v) { return v + __shfl_xor(v, 32); }

// part 2 of the evaluation phase (see phase_eval); arrays as __restrict__ parameters so that the loads of
// several row rounds can be batched across the record stores
template <int NP>
__device__ __forceinline__ void eval_rows(cgdouble *__restrict__ ps, cgdouble *__restrict__ pl, cgdouble *__restrict__ pz,
                                          cgdouble *__restrict__ pface, gdouble *__restrict__ prec, cgdouble *__restrict__ pbase,
                                          int np, int N, int MF, int nfk, int model, double *stg,
                                          double &l_in, double &l_rc, double &l_gap, double &l_rs)
{
    constexpr int H = 64 / NP;
    const int lane = threadIdx.x;
    // ---- part 2: all 64 lanes, lane = (half, stage k); rows handled in pairs
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N;
    // corridor rows: sums over the faces of a stage (pos entries 8..10 only)
    {
        double gp0 = 0, gp1 = 0, gp2 = 0, fp0 = 0, fp1 = 0, fp2 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
        if (kact) {
            const double z8 = pz[8 * NP + k], z9 = pz[9 * NP + k], z10 = pz[10 * NP + k];
            for (int j = half; j < nfk; j += H) {
                const double a0 = pface[(3 * j) * NP + k], a1 = pface[(3 * j + 1) * NP + k], a2 = pface[(3 * j + 2) * NP + k];
                const double hj = a0 * z8 + a1 * z9 + a2 * z10 - pface[(3 * MF + j) * NP + k] - HU;
                const double sc = ps[(34 + j) * NP + k], lc = pl[(34 + j) * NP + k];
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
            gdouble *rec = prec + (size_t)k * REC_STRIDE;
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
        gdouble *rec = prec + (size_t)k * REC_STRIDE;
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
            const double zi = pz[i * NP + k];
            double cg = hd * zi + qi; // cost gradient
            if (i0 < 8) cg += cq.hc() * pz[(i < 4 ? i + 4 : i - 4) * NP + k];
            const double sl = ps[i * NP + k], su = ps[(17 + i) * NP + k];
            const double ll = pl[i * NP + k], lu = pl[(17 + i) * NP + k];
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
}

// ------------------------------------------------------------------ E: evaluate
// part 1 (lane == stage): model + linearisation -> record, equality residuals, M'y -> LDS
// part 2 (lane == (row pair, stage), all 64 lanes): corridor rows, then bounds: residual norms,
//        barrier Hessian / affine rhs -> record
template <int NP>
__device__ __noinline__ EvalOut phase_eval(WsView w, cgdouble *pbase, int np, cgdouble *xinit, int N, int MF, int nfk, int model, int hess)
{
    w = uni(w); pbase = uni(pbase); np = uni(np); xinit = uni(xinit); N = uni(N); MF = uni(MF); model = uni(model); hess = uni(hess);
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
            if (hess) {
                // exact Hessian of y_{k+1}' c(z_k): only the pos / vel rows of the RK2 step are non-linear
                rk2_hessian(zk + 8, zk, p10 + 3, yp, yv, [&](int i, int j, double val) {
                    rec[REC_HD + i * 10 + j] = val;
                    if (i != j) rec[REC_HD + j * 10 + i] = val;
                });
            }
        }
#pragma unroll
        for (int i = 0; i < NZ; i++) stg[i * NP + k] = gm[i];
    }
    // ---- part 2: all 64 lanes, lane = (half, stage k); rows handled in pairs
    WSYNC();
    eval_rows<NP>(w.s, w.lam, w.z, w.face, w.rec, pbase, np, N, MF, nfk, model, stg, l_in, l_rc, l_gap, l_rs);
    FULLSYNC();
    EvalOut o;
    o.eq = l_eq; o.in = l_in; o.rs = l_rs; o.rc = l_rc; o.gap = l_gap; o.obj = l_obj;
    return o;
}

// ------------------------------------------------------------------ stage-0 solve (both passes)
// dx_0 = xinit - x_0, dw_0 = -Pww^-1 (Pwx dx_0 + p_w); leaves ds_0 = [dw_0; dx_0] in LDS (S_DS0).
// pw_here: p_w[g] in the lanes (g, 13).
template <int NP>
__device__ __forceinline__ void stage0_solve(const WsView &w, cgdouble *xinit, int lane, double pw_here)
{
    const int g = lane >> 4, c = lane & 15;
    const bool xc = (c >= 4 && c <= 12);
    const double dxc = xc ? xinit[c - 4] - w.z[(8 + c - 4) * NP + 0] : 0.0;
    WSYNC();
    const double prod = xc ? sm[S_PWX + g * 9 + c - 4] * dxc : (c == 13 ? pw_here : 0.0);
    const double rhs = row16_sum(prod);
    const double r0 = lane_bcast(rhs, 0), r1 = lane_bcast(rhs, 16), r2 = lane_bcast(rhs, 32), r3 = lane_bcast(rhs, 48);
    if (lane < 4) {
        sm[S_DS0 + lane] = -(sm[S_RW + lane * 4 + 0] * r0 + sm[S_RW + lane * 4 + 1] * r1 +
                             sm[S_RW + lane * 4 + 2] * r2 + sm[S_RW + lane * 4 + 3] * r3);
    } else if (lane <= 12) {
        sm[S_DS0 + lane] = dxc; // g == 0: index 4 + (c - 4) = c
    } else if (lane < 16) {
        sm[S_DS0 + lane] = 0.0;
    }
    WSYNC();
}

// ------------------------------------------------------------------ factorisation sweep (predictor)
// Backward Riccati recursion with everything in register tiles (see the header).  Per stage:
//   X = P M (col 13: P d + p+),  G = M'X + C~ (col 13: q~),  R = Guu^-1,
//   T = R G_u (Kbar, kbar),  TT = G_u' R (Kbar'),  S = G - G_u' T,
//   P <- [Phi_w - hc^2 R, -hc Kbar_x; -hc Kbar_x', S_xx],  p <- [phi_w - hc kbar; S_x,13].
// Streams T' = [R | Kbar_x | kbar | hc] and P d to the stage record.  Returns 1 when a pivot block
// is not positive definite (exact Hessian: the caller retries with theta = 0, Gauss-Newton).
// Software pipeline: while the MFMA chain of stage k executes, the wave stages the (prefetched) record of
// stage k-1 through LDS and assembles its tiles, and issues the global prefetch of stage k-2.
struct FactorTiles {
    d4 C, M;
    double hc, PhiDw, phiw;
};

template <int NP>
__device__ __noinline__ int sweep_factor(WsView w, cgdouble *xinit, int N, int theta_i)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); theta_i = uni(theta_i);
    FULLSYNC(); // phase boundary: the evaluation phase's record writes are visible
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const double theta = theta_i ? 1.0 : 0.0;
    int mo[4], c1[4], c2[4], c3[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = m_src(4 * r + g, c);
        c_src(4 * r + g, c, c1[r], c2[r], c3[r]);
    }
    init_stage_constants(lane);
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 P = zero, pv = zero;
    bool fail = false;
    double e0, e1, e2, e3;
    auto fetch = [&](int kk) {
        cgdouble *rp = w.rec + (size_t)kk * REC_STRIDE;
        e0 = rp[lane]; e1 = rp[64 + lane]; e2 = rp[128 + lane]; e3 = (lane < 56) ? rp[192 + lane] : 0.0;
    };
    auto stage = [&]() -> FactorTiles { // regs -> LDS -> tiles of that stage
        WSYNC();
        sm[S_E + lane] = e0; sm[S_E + 64 + lane] = e1; sm[S_E + 128 + lane] = e2;
        if (lane < 56) sm[S_E + 192 + lane] = e3;
        WSYNC();
        FactorTiles t;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            t.C[r] = sm[c1[r]] + sm[c2[r]] + theta * sm[c3[r]];
            t.M[r] = sm[mo[r]];
        }
        t.hc = sm[S_E + REC_HC];
        t.PhiDw = sm[S_E + REC_PHID + 4 + g];
        t.phiw = sm[S_E + REC_PHI + 4 + g];
        return t;
    };
    fetch(N - 1);
    FactorTiles cur = stage();
    if (N > 1) fetch(N - 2);
    for (int kk = N - 1; kk >= 0; kk--) {
        gdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
        const bool last = (kk == N - 1);
        d4 G = cur.C;
        if (!last) {
            d4 X = mm_tn(P, cur.M, zero);
            if (c == 13) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    rec[REC_PD + 4 * r + g] = X[r];
                    X[r] += pv[r];
                }
            }
            G = mm_tn(cur.M, X, cur.C);
        }
        // ---- overlapped with the MFMA chain above: tiles of the next stage to be processed
        FactorTiles nxt = cur;
        if (kk > 0) {
            nxt = stage();
            if (kk > 1) fetch(kk - 2);
        }
        // ---- R = Guu^-1 (4 x 4): gather the lower triangle to uniform registers, invert redundantly
        double q[16], R[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(G[0], 16 * i + j);
        if (!spd4_inverse(q, R)) { fail = true; break; }
#pragma unroll
        for (int t = 0; t < 16; t++) sm[S_R + t] = R[t];
        WSYNC();
        const double rt = (c < 4) ? sm[S_R + g * 4 + c] : 0.0;
        const double hc = cur.hc;
        const d4 T = mm_tn4(rt, G[0], zero);
        const d4 TT = mm_tn4(G[0], rt, zero);
        const d4 S = mm_tn4(-G[0], T[0], G);
        rec[REC_T + lane] = (c < 4) ? rt : (c <= 13 ? T[0] : (lane == 14 ? hc : 0.0));
        d4 Pn, pn;
        Pn[0] = (c < 4) ? ((g == c ? cur.PhiDw : 0.0) - hc * hc * rt) : (c <= 12 ? -hc * T[0] : 0.0);
        pn[0] = (c == 13) ? (cur.phiw - hc * T[0]) : 0.0;
#pragma unroll
        for (int r = 1; r < 4; r++) {
            const bool inb = (4 * r + g) <= 12;
            Pn[r] = (inb && c <= 12) ? (c < 4 ? -hc * TT[r] : S[r]) : 0.0;
            pn[r] = (inb && c == 13) ? S[r] : 0.0;
        }
        P = Pn;
        pv = pn;
        cur = nxt;
    }
    if (!fail) {
        // stage 0: keep Pww^-1 and Pwx for the corrector pass, then solve for ds_0
        double q[16], Rw[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(P[0], 16 * i + j);
        if (!spd4_inverse(q, Rw)) fail = true;
        else {
            WSYNC();
#pragma unroll
            for (int t = 0; t < 16; t++) sm[S_RW + t] = Rw[t];
            if (c >= 4 && c <= 12) sm[S_PWX + g * 9 + c - 4] = P[0];
            stage0_solve<NP>(w, xinit, lane, pv[0]);
        }
    }
    FULLSYNC();
    return fail ? 1 : 0;
}

// ------------------------------------------------------------------ vector-only backward sweep (corrector)
// Same factorisation, new rhs phi_cc = PHIB + smu PHIC:  q~ = phi~ + M'(P d + p+),  [kbar; Kbar'q_u] = T'' q_u,
// p_x = q~_x - Kbar' q_u,  p_w = phi_w - hc kbar.  Updates the kbar column of T'.
struct BackvecTiles {
    d4 M, Gp, pd;
    double hc, phiw, tp;
};

template <int NP>
__device__ __noinline__ void sweep_backvec(WsView w, cgdouble *xinit, int N, double smu)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); smu = uni(smu);
    FULLSYNC(); // phase boundary: the corrector rhs written by the step phase is visible
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    int mo[4], po[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = m_src(4 * r + g, c);
        po[r] = (c == 13 && 4 * r + g <= 12) ? zi_of(4 * r + g) : -1;
    }
    init_stage_constants(lane);
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 pv = zero;
    double e0, e1, e2, tpre;
    d4 pdpre = zero;
    auto fetch = [&](int kk) {
        cgdouble *rp = w.rec + (size_t)kk * REC_STRIDE;
        e0 = rp[lane]; e1 = rp[64 + lane]; e2 = (lane < 17) ? rp[REC_PHIC + lane] : 0.0; tpre = rp[REC_T + lane];
        if (c == 13) {
#pragma unroll
            for (int r = 0; r < 4; r++) pdpre[r] = rp[REC_PD + 4 * r + g];
        }
    };
    auto stage = [&]() -> BackvecTiles {
        WSYNC();
        sm[S_E + lane] = e0; sm[S_E + 64 + lane] = e1;
        if (lane < 17) sm[S_E + REC_PHIC + lane] = e2;
        WSYNC();
        BackvecTiles t;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            t.M[r] = sm[mo[r]];
            const int o = po[r] >= 0 ? po[r] : 0;
            const double ph = sm[S_E + REC_PHIB + o] + smu * sm[S_E + REC_PHIC + o];
            t.Gp[r] = (po[r] >= 0) ? ph : 0.0;
        }
        t.hc = sm[S_E + REC_HC];
        t.phiw = sm[S_E + REC_PHIB + 4 + g] + smu * sm[S_E + REC_PHIC + 4 + g];
        t.tp = tpre;
        t.pd = pdpre;
        return t;
    };
    fetch(N - 1);
    BackvecTiles cur = stage();
    if (N > 1) fetch(N - 2);
    for (int kk = N - 1; kk >= 0; kk--) {
        gdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
        const bool last = (kk == N - 1);
        d4 Gp = cur.Gp;
        if (!last) {
            d4 X;
#pragma unroll
            for (int r = 0; r < 4; r++) X[r] = (c == 13) ? cur.pd[r] + pv[r] : 0.0;
            Gp = mm_tn(cur.M, X, Gp);
        }
        const d4 E = mm_tn4(cur.tp, Gp[0], zero);
        BackvecTiles nxt = cur;
        if (kk > 0) {
            nxt = stage();
            if (kk > 1) fetch(kk - 2);
        }
        if (c == 13) rec[REC_T + 16 * g + 13] = E[0]; // kbar
        d4 pn;
        pn[0] = (c == 13) ? (cur.phiw - cur.hc * E[0]) : 0.0;
#pragma unroll
        for (int r = 1; r < 4; r++) pn[r] = (c == 13 && 4 * r + g <= 12) ? Gp[r] - E[r] : 0.0;
        pv = pn;
        cur = nxt;
    }
    stage0_solve<NP>(w, xinit, lane, pv[0]);
    FULLSYNC();
}

// ------------------------------------------------------------------ forward sweep: dz for all stages
// du = -T' [hc dw; dx; 1],  ds+ = Mt [du; dx; 1]; the vectors stay in the column-0 lanes (row layout).
// Software pipeline, branch-free body: the record of stage k+1 is staged through LDS and its operand tiles
// are read BETWEEN the chained MFMAs of stage k (each chained MFMA blocks the wave for 64 cycles anyway);
// the global prefetch runs two stages ahead.  Two register sets (A/B) alternate, so no tile is ever copied.
template <int NP>
__device__ __forceinline__ void forward_step(const WsView &w, int N, int kk, int lane, int g, int c, const int (&tto)[4],
                                             const int (&mto)[4], const d4 &ctt, const d4 &cmt, double chc, d4 &ntt, d4 &nmt,
                                             double &nhc, double &e0, double &tp, d4 &v, long long *pacc_, long long &pts_)
{
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 v1 = v;
    if (c == 0) {
        v1[0] = chc * v[0];
        if (g == 1) v1[3] = 1.0; // row 13 multiplies the kbar column
    }
    PROF_SEG(0);
    d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ctt[0], v1[0], zero, 0, 0, 0);
#ifdef FRP_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PROF_SEG(1);
    // stage the (already fetched) record of the next stage through LDS ...
    WSYNC();
    sm[S_E + lane] = e0; sm[S_T + lane] = tp;
    WSYNC();
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ctt[1], v1[1], D1, 0, 0, 0);
    { // ... prefetch the one after it (clamped: the tail re-reads the last record, unused) ...
        const int kf = (kk + 2 < N) ? kk + 2 : N - 1;
        cgdouble *rp = w.rec + (size_t)kf * REC_STRIDE;
        e0 = rp[lane]; tp = rp[REC_T + lane];
    }
#pragma unroll
    for (int s = 0; s < 4; s++) ntt[s] = sm[tto[s]];
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ctt[2], v1[2], D1, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; s++) nmt[s] = sm[mto[s]];
    nhc = sm[S_T + 14];
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ctt[3], v1[3], D1, 0, 0, 0);
    PROF_SEG(2);
    const double du = -D1[0];
#ifdef FRP_PROFILE
    asm volatile("s_nop 0" :: "v"(du));
#endif
    PROF_SEG(3);
    if (c == 0) { // dz rows 17..19 are padding (tile rows 13..15)
        double *dzl = dz_area<NP>();
        dzl[g * NP + kk] = du;
#pragma unroll
        for (int r = 0; r < 4; r++) dzl[(4 + 4 * r + g) * NP + kk] = v[r];
    }
    d4 v2 = v;
    if (c == 0) {
        v2[0] = du;
        if (g == 1) v2[3] = 1.0; // row 13 multiplies the d column
    }
    const d4 D2 = mm_tn(cmt, v2, zero);
    PROF_SEG(4);
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (c == 0 && 4 * r + g <= 12) ? D2[r] : 0.0;
#ifdef FRP_PROFILE
    asm volatile("s_nop 0" :: "v"(v[0]));
#endif
    PROF_SEG(5);
}

template <int NP>
__device__ __noinline__ void sweep_forward(WsView w, int N)
{
    w = uni(w); N = uni(N);
    FULLSYNC(); // phase boundary: T' / kbar of the backward sweep are visible
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    int mto[4], tto[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        mto[s] = m_src(c, 4 * s + g);                           // Mt' tile: element [c][4s+g]
        tto[s] = (c < 4) ? S_T + 16 * c + 4 * s + g : S_ZERO;   // T'' tile: element T'[c][4s+g]
    }
    init_stage_constants(lane);
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (c == 0 && 4 * r + g <= 12) ? sm[S_DS0 + 4 * r + g] : 0.0;
    double e0, tp;
    {
        cgdouble *rp = w.rec;
        e0 = rp[lane]; tp = rp[REC_T + lane];
    }
    WSYNC();
    sm[S_E + lane] = e0; sm[S_T + lane] = tp;
    WSYNC();
    d4 ttA, mtA, ttB, mtB;
    double hcA, hcB = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) { ttA[s] = sm[tto[s]]; mtA[s] = sm[mto[s]]; }
    hcA = sm[S_T + 14];
    {
        cgdouble *rp = w.rec + (size_t)(N > 1 ? 1 : 0) * REC_STRIDE;
        e0 = rp[lane]; tp = rp[REC_T + lane];
    }
#ifdef FRP_PROFILE
    PROF_BEGIN();
#else
    long long pacc_[1], pts_ = 0;
#endif
    int kk = 0;
    for (; kk + 1 < N; kk += 2) {
        forward_step<NP>(w, N, kk, lane, g, c, tto, mto, ttA, mtA, hcA, ttB, mtB, hcB, e0, tp, v, pacc_, pts_);
        forward_step<NP>(w, N, kk + 1, lane, g, c, tto, mto, ttB, mtB, hcB, ttA, mtA, hcA, e0, tp, v, pacc_, pts_);
    }
    if (kk < N) forward_step<NP>(w, N, kk, lane, g, c, tto, mto, ttA, mtA, hcA, ttB, mtB, hcB, e0, tp, v, pacc_, pts_);
    WSYNC();
    PROF_END(12);
}

struct SlackOut {
    double ap, ad, sigma, smu;
};

// ------------------------------------------------------------------ slack / multiplier steps
// All 64 lanes, lane = (half, stage k), constraints handled in pairs (flattened [row][stage] arrays).
// Per constraint:  ds = -(G z - g + s) - G dz,  dl = (-(s l - smu + corr) - l ds) / s.
// Ratios -ds/s and -dl/l are formed with ONE reciprocal u = 1/(s l) per constraint.
//
// phase_affine (predictor): ONE pass over the constraints gives the step lengths (max ratios), the
// second-order term corr = ds dl, the pieces of the affine complementarity
//     sum (s + ap ds)(l + ad dl) = sum s l + ad sum s dl + ap sum l ds + ap ad sum ds dl
// and the corrector rhs split as  phi_cc = PHIB + (sigma mu) PHIC  (sigma mu is only known after the
// wave-wide reductions, the sweeps apply it):
//     PHIB = grad f + G'((l r_in - corr)/s),   PHIC = G'(1/s).
template <int NP>
__device__ __forceinline__ void affine_body(cgdouble *__restrict__ ps, cgdouble *__restrict__ pl, gdouble *__restrict__ pcorr,
                                            cgdouble *__restrict__ pz, const double *__restrict__ pdz, cgdouble *__restrict__ pface,
                                            gdouble *__restrict__ prec, cgdouble *__restrict__ pbase, int np, int N, int MF, int nfk,
                                            int model, double &m_p, double &m_d, double &s_sdl, double &s_lds, double &s_dsdl)
{
    constexpr int H = 64 / NP;
    constexpr int R = (NZ + H - 1) / H;
    const int lane = threadIdx.x;
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N;
    double *stg = stage_area<NP>();

    // one constraint of the affine step (smu = 0, corr = 0): returns t1 = (l r_in - corr)/s, sinv = 1/s
    auto cstep = [&](int c, double gdz, double viol, double &t1, double &sinv) {
        const double s = ps[c * NP + k], l = pl[c * NP + k];
        const double u = 1.0 / (s * l);
        sinv = u * l;
        const double linv = u * s;
        const double rin = viol + s;
        const double ds = -rin - gdz;
        const double dl = -l * (1.0 + ds * sinv); // (-(s l) - l ds) / s
        m_p = fmax(m_p, -ds * sinv);
        m_d = fmax(m_d, -dl * linv);
        s_sdl += s * dl; s_lds += l * ds;
        const double cr = ds * dl;
        s_dsdl += cr;
        pcorr[c * NP + k] = cr;
        t1 = (l * rin - cr) * sinv;
    };
    // corridor rows first: their sums go to the pos entries of PHIB / PHIC
    {
        double b0 = 0, b1 = 0, b2 = 0, c0 = 0, c1 = 0, c2 = 0;
        if (kact) {
            const double z8 = pz[8 * NP + k], z9 = pz[9 * NP + k], z10 = pz[10 * NP + k];
            const double d8 = pdz[8 * NP + k], d9 = pdz[9 * NP + k], d10 = pdz[10 * NP + k];
            for (int j = half; j < nfk; j += H) {
                const double a0 = pface[(3 * j) * NP + k], a1 = pface[(3 * j + 1) * NP + k], a2 = pface[(3 * j + 2) * NP + k];
                double t1, sinv;
                cstep(34 + j, a0 * d8 + a1 * d9 + a2 * d10, a0 * z8 + a1 * z9 + a2 * z10 - pface[(3 * MF + j) * NP + k] - HU, t1, sinv);
                b0 += a0 * t1; b1 += a1 * t1; b2 += a2 * t1;
                c0 += a0 * sinv; c1 += a1 * sinv; c2 += a2 * sinv;
            }
        }
        if (H == 2) {
            b0 = xhalf_sum(b0); b1 = xhalf_sum(b1); b2 = xhalf_sum(b2);
            c0 = xhalf_sum(c0); c1 = xhalf_sum(c1); c2 = xhalf_sum(c2);
        }
        if (kact && half == 0) {
            stg[0 * NP + k] = b0; stg[1 * NP + k] = b1; stg[2 * NP + k] = b2;
            stg[3 * NP + k] = c0; stg[4 * NP + k] = c1; stg[5 * NP + k] = c2;
        }
    }
    WSYNC();
    if (kact) {
        cgdouble *pk = pbase + (size_t)k * np;
        double pc[NPRE];
        pc[0] = pk[0]; pc[1] = pk[1]; pc[2] = pk[2]; pc[6] = pk[6]; pc[7] = pk[7]; pc[8] = pk[8]; pc[9] = pk[9];
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        gdouble *rec = prec + (size_t)k * REC_STRIDE;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
            if (H == 2 && i1 >= NZ && half) continue;
            const int i = half ? i1 : i0;
            const double hd = half ? cq.hd(i1 < NZ ? i1 : i0) : cq.hd(i0);
            const double qi = half ? cq.q(i1 < NZ ? i1 : i0) : cq.q(i0);
            const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
            const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
            const double zi = pz[i * NP + k], dzi = pdz[i * NP + k];
            double pb = hd * zi + qi; // cost gradient
            if (i0 < 8) pb += cq.hc() * pz[(i < 4 ? i + 4 : i - 4) * NP + k];
            double tl, tu, sil, siu;
            cstep(i, -dzi, lb - zi, tl, sil);
            cstep(17 + i, dzi, zi - ub, tu, siu);
            pb += tu - tl;
            double pcf = siu - sil;
            if (i0 + H > 8 && i0 < 11) {
                if (i >= 8 && i < 11) { pb += stg[(i - 8) * NP + k]; pcf += stg[(3 + i - 8) * NP + k]; }
            }
            rec[REC_PHIB + i] = pb;
            rec[REC_PHIC + i] = pcf;
        }
    }
}

template <int NP>
__device__ __noinline__ SlackOut phase_affine(WsView w, cgdouble *pbase, int np, int N, int MF, int nfk, int model,
                                              double mu, int mtot, double tol_comp)
{
    w = uni(w); pbase = uni(pbase); np = uni(np); N = uni(N); MF = uni(MF); model = uni(model);
    mu = uni(mu); mtot = uni(mtot); tol_comp = uni(tol_comp);
    PROF_BEGIN();
    FULLSYNC(); // phase boundary: dz of the forward sweep is visible
    PROF_SEG(0);
    double m_p = 0.0, m_d = 0.0, s_sdl = 0.0, s_lds = 0.0, s_dsdl = 0.0;
    affine_body<NP>(w.s, w.lam, w.corr, w.z, dz_area<NP>(), w.face, w.rec, pbase, np, N, MF, nfk, model, m_p, m_d, s_sdl, s_lds, s_dsdl);
    PROF_SEG(2);
    m_p = wave_max(m_p); m_d = wave_max(m_d);
    const double ap = (m_p > 1.0) ? 1.0 / m_p : 1.0;
    const double ad = (m_d > 1.0) ? 1.0 / m_d : 1.0;
    const double gap_aff = mu * (double)mtot + ad * wave_sum(s_sdl) + ap * wave_sum(s_lds) + ap * ad * wave_sum(s_dsdl);
    double sigma = gap_aff / ((double)mtot * mu);
    sigma = sigma * sigma * sigma;
    if (sigma > 1.0) sigma = 1.0;
    double smu = sigma * mu;
    if (smu < MU_FLOOR_FRAC * tol_comp) smu = MU_FLOOR_FRAC * tol_comp;
    PROF_SEG(3);
    FULLSYNC();
    PROF_SEG(4);
    PROF_END(0);
    SlackOut o;
    o.ap = ap; o.ad = ad; o.sigma = sigma; o.smu = smu;
    return o;
}

// phase_step (corrector): pass A computes ds, dl and the fraction-to-boundary step lengths, pass B applies
// z += ap dz, s += ap ds, l += ad dl.  The arrays are passed as __restrict__ parameters of an inlined
// helper so that the compiler may batch the loads of several constraint rounds across the stores (it
// cannot prove on its own that the workspace arrays do not alias, which serialises every round on a
// full memory round trip).
template <int NP>
__device__ __forceinline__ void step_body(gdouble *__restrict__ ps, gdouble *__restrict__ pl, cgdouble *__restrict__ pcorr,
                                          gdouble *__restrict__ pz, const double *__restrict__ pdz, cgdouble *__restrict__ pface,
                                          int N, int MF, int nfk, double smu, double ftb, double &ap_out, double &ad_out)
{
    constexpr int H = 64 / NP;
    constexpr int R = (NZ + H - 1) / H;
    constexpr int MAXF = 8; // corridor rounds kept in registers; further rounds are recomputed
    const int lane = threadIdx.x;
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N;
    double m_p = 0.0, m_d = 0.0;
    double dsb[2 * R], dlb[2 * R], dsf[MAXF], dlf[MAXF];
    double z8 = 0, z9 = 0, z10 = 0, d8 = 0, d9 = 0, d10 = 0;
    auto cstep = [&](int c, double gdz, double viol, double &ds, double &dl) {
        const double s = ps[c * NP + k], l = pl[c * NP + k];
        const double u = 1.0 / (s * l);
        const double sinv = u * l, linv = u * s;
        ds = -(viol + s) - gdz;
        const double rc = s * l - smu + pcorr[c * NP + k];
        dl = (-rc - l * ds) * sinv;
        m_p = fmax(m_p, -ds * sinv);
        m_d = fmax(m_d, -dl * linv);
    };
    auto face = [&](int j, double &ds, double &dl) {
        const double a0 = pface[(3 * j) * NP + k], a1 = pface[(3 * j + 1) * NP + k], a2 = pface[(3 * j + 2) * NP + k];
        cstep(34 + j, a0 * d8 + a1 * d9 + a2 * d10, a0 * z8 + a1 * z9 + a2 * z10 - pface[(3 * MF + j) * NP + k] - HU, ds, dl);
    };
    if (kact) {
        z8 = pz[8 * NP + k]; z9 = pz[9 * NP + k]; z10 = pz[10 * NP + k];
        d8 = pdz[8 * NP + k]; d9 = pdz[9 * NP + k]; d10 = pdz[10 * NP + k];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i0 = r * H, i1 = (H == 2) ? i0 + 1 : i0;
            dsb[2 * r] = dlb[2 * r] = dsb[2 * r + 1] = dlb[2 * r + 1] = 0.0;
            if (H == 2 && i1 >= NZ && half) continue;
            const int i = half ? i1 : i0;
            const double lb = half ? lower_bound(i1 < NZ ? i1 : i0) : lower_bound(i0);
            const double ub = half ? upper_bound(i1 < NZ ? i1 : i0) : upper_bound(i0);
            const double zi = pz[i * NP + k], dzi = pdz[i * NP + k];
            cstep(i, -dzi, lb - zi, dsb[2 * r], dlb[2 * r]);
            cstep(17 + i, dzi, zi - ub, dsb[2 * r + 1], dlb[2 * r + 1]);
        }
#pragma unroll
        for (int t = 0; t < MAXF; t++) {
            const int j = half + t * H;
            dsf[t] = dlf[t] = 0.0;
            if (j < nfk) face(j, dsf[t], dlf[t]);
        }
        for (int j = half + MAXF * H; j < nfk; j += H) { double a, b; face(j, a, b); }
    }
    m_p = wave_max(m_p); m_d = wave_max(m_d);
    const double ap = (m_p > ftb) ? ftb / m_p : 1.0;
    const double ad = (m_d > ftb) ? ftb / m_d : 1.0;
    if (kact) {
        for (int j = half + MAXF * H; j < nfk; j += H) { // rare: more corridor rounds than kept in registers
            double a, b;
            face(j, a, b);
            ps[(34 + j) * NP + k] += ap * a;
            pl[(34 + j) * NP + k] += ad * b;
        }
#pragma unroll
        for (int t = 0; t < MAXF; t++) {
            const int j = half + t * H;
            if (j < nfk) {
                ps[(34 + j) * NP + k] += ap * dsf[t];
                pl[(34 + j) * NP + k] += ad * dlf[t];
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = r * H + half;
            if (i >= NZ) continue;
            ps[i * NP + k] += ap * dsb[2 * r];
            pl[i * NP + k] += ad * dlb[2 * r];
            ps[(17 + i) * NP + k] += ap * dsb[2 * r + 1];
            pl[(17 + i) * NP + k] += ad * dlb[2 * r + 1];
            pz[i * NP + k] += ap * pdz[i * NP + k];
        }
    }
    ap_out = ap; ad_out = ad;
}

template <int NP>
__device__ __noinline__ SlackOut phase_step(WsView w, int N, int MF, int nfk, double smu, double ftb)
{
    w = uni(w); N = uni(N); MF = uni(MF); smu = uni(smu); ftb = uni(ftb);
    PROF_BEGIN();
    FULLSYNC(); // phase boundary: dz of the forward sweep is visible
    PROF_SEG(6);
    double ap, ad;
    step_body<NP>(w.s, w.lam, w.corr, w.z, dz_area<NP>(), w.face, N, MF, nfk, smu, ftb, ap, ad);
    PROF_SEG(7);
    FULLSYNC();
    PROF_SEG(8);
    PROF_END(6);
    SlackOut o;
    o.ap = ap; o.ad = ad; o.sigma = 0.0; o.smu = smu;
    return o;
}

// ------------------------------------------------------------------ costate sweep: y <- y + ap (y+ - y)
// y+_k = (Phi_k dz_k + phi_k)_s + [0; A_k' y+_{k+1,x}]:  x rows = (C~' [du; dx] + M' y+)_x + phi_x,
// w rows = Phi_w dw + hc du + phi_w.  Vectors in the column-0 lanes (row layout).
// ------------------------------------------------------------------ costate sweep: y <- y + ap (y+ - y)
// y+_k = (Phi_k dz_k + phi_k)_s + [0; A_k' y+_{k+1,x}]:  x rows = (C~' [du; dx] + M' y+)_x + phi_x,
// w rows = Phi_w dw + hc du + phi_w, with phi = phi_cc = PHIB + smu PHIC.
// Vectors in the column-0 lanes (row layout); dz / y have padding rows so that no row guards are needed.
template <int NP>
__device__ __noinline__ void sweep_costate(WsView w, int N, double ap, double smu, int theta_i)
{
    w = uni(w); N = uni(N); ap = uni(ap); smu = uni(smu); theta_i = uni(theta_i);
    FULLSYNC(); // phase boundary: dz of the forward sweep / updates of the step phase are visible
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const double theta = theta_i ? 1.0 : 0.0;
    int mo[4], c1[4], c2[4], c3[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = m_src(4 * r + g, c);
        c_src(4 * r + g, c, c1[r], c2[r], c3[r]);
    }
    init_stage_constants(lane);
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 y = zero;
    cgdouble *rp = w.rec + (size_t)(N - 1) * REC_STRIDE;
    double e0 = rp[lane], e1 = rp[64 + lane], e2 = rp[128 + lane], e3 = (lane < 56) ? rp[192 + lane] : 0.0;
    d4 nv = zero, nyo = zero; // prefetched dz and old y of the next stage to be processed (column-0 lanes)
    double ndu = 0.0;
    if (c == 0) {
        const double *dzl = dz_area<NP>();
        ndu = dzl[g * NP + N - 1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            nv[r] = dzl[(4 + 4 * r + g) * NP + N - 1];
            nyo[r] = w.y[(4 * r + g) * NP + N - 1];
        }
    }
    for (int kk = N - 1; kk >= 0; kk--) {
        const bool last = (kk == N - 1);
        WSYNC();
        sm[S_E + lane] = e0; sm[S_E + 64 + lane] = e1; sm[S_E + 128 + lane] = e2;
        if (lane < 56) sm[S_E + 192 + lane] = e3;
        const d4 ds = nv, yo = nyo;
        const double du = ndu;
        if (kk > 0) {
            cgdouble *rn = w.rec + (size_t)(kk - 1) * REC_STRIDE;
            e0 = rn[lane]; e1 = rn[64 + lane]; e2 = rn[128 + lane]; e3 = (lane < 56) ? rn[192 + lane] : 0.0;
            if (c == 0) {
                const double *dzl = dz_area<NP>();
                ndu = dzl[g * NP + kk - 1];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    nv[r] = dzl[(4 + 4 * r + g) * NP + kk - 1];
                    nyo[r] = w.y[(4 * r + g) * NP + kk - 1];
                }
            }
        }
        WSYNC();
        const double hc = sm[S_E + REC_HC];
        d4 C, v2 = ds;
        v2[0] = du; // [du; dx] over the (u, x) tile index; ds[0] = dw is used for the w rows below
#pragma unroll
        for (int r = 0; r < 4; r++) C[r] = sm[c1[r]] + sm[c2[r]] + theta * sm[c3[r]];
        d4 D = mm_tn(C, v2, zero);
        if (!last) {
            d4 M;
#pragma unroll
            for (int r = 0; r < 4; r++) M[r] = sm[mo[r]];
            D = mm_tn(M, y, D);
        }
        d4 yn = zero;
        if (c == 0) {
            const int zw = 4 + g;
            yn[0] = sm[S_E + REC_PHID + zw] * ds[0] + hc * du + sm[S_E + REC_PHIB + zw] + smu * sm[S_E + REC_PHIC + zw];
#pragma unroll
            for (int r = 1; r < 4; r++) {
                const int zr = (4 * r + g <= 12) ? 4 * r + g + 4 : 16; // pad rows alias a valid slot, result discarded
                yn[r] = (4 * r + g <= 12) ? D[r] + sm[S_E + REC_PHIB + zr] + smu * sm[S_E + REC_PHIC + zr] : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) w.y[(4 * r + g) * NP + kk] = yo[r] + ap * (yn[r] - yo[r]);
        }
        y = yn;
    }
    WSYNC();
}

// ------------------------------------------------------------------ the solver kernel
template <int NP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRP_WAVES_PER_EU, FRP_WAVES_PER_EU))) void nmpc_ipm_kernel(KernelArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = a.N, M = a.M, MF = a.MF, np = NPRE + 4 * M;
    const int mcf = 34 + MF;
    const bool act = lane < N; // lane == stage in the initialisation
    const int k = lane;

    WsView w;
    {
        gdouble *base = (gdouble *)(a.ws + (size_t)b * ws_doubles_per_problem(N, MF));
        w.rec = base;
        w.z = w.rec + (size_t)N * REC_STRIDE;
        w.y = w.z + 17 * NP;
        w.dz = w.y + Y_ROWS * NP;
        w.s = w.dz + DZ_ROWS * NP;
        w.lam = w.s + (size_t)mcf * NP;
        w.corr = w.lam + (size_t)mcf * NP;
        w.step = w.corr + (size_t)mcf * NP;       // ds | dlam of the corrector step ([2 mcf][NP])
        w.face = w.step + 2 * (size_t)mcf * NP;
    }
    cgdouble *xinit = (cgdouble *)(a.xinit + (size_t)b * 9);
    cgdouble *pbase = (cgdouble *)(a.params + (size_t)b * N * np);
    cgdouble *pk = pbase + (size_t)(act ? k : 0) * np;

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
        gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
        rec[REC_HC] = -2.0 * pk[8]; // (u_i, w_i) cost coupling of this stage (constant)
        for (int i = 0; i < 100; i++) rec[REC_HD + i] = 0.0; // stays zero in Gauss-Newton mode / last stage
        for (int i = 0; i < 64; i++) rec[i] = 0.0;           // linearisation of the last stage is never written
        for (int i = 125; i < 128; i++) rec[i] = 0.0;        // padding slots of the E record
        for (int i = 145; i < 148; i++) rec[i] = 0.0;
        for (int i = 0; i < 3; i++) { // padding rows (tile rows 13..15) of dz and y
            dz_area<NP>()[(17 + i) * NP + k] = 0.0;
            w.y[(13 + i) * NP + k] = 0.0;
        }
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
    const int nfk = __shfl(nf, lane % NP); // face count of stage k = lane % NP for the (half, stage) lane mapping
    const int hess = a.hessian ? 1 : 0;
    FULLSYNC();

    int flag = FRP_EXIT_MAXIT, it = 0, nfallback = 0;
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
        const EvalOut e = phase_eval<NP>(w, pbase, np, xinit, N, MF, nfk, a.model, hess);
        res_eq = wave_max(e.eq); res_in = wave_max(e.in); rs = wave_max(e.rs); rcomp = wave_max(e.rc);
        pobj = wave_sum(e.obj);
        mu = wave_sum(e.gap) / (double)mtot;
        if (!(res_eq == res_eq) || !(rs == rs) || !(pobj == pobj)) { flag = FRP_EXIT_BADFUNCEVAL; break; }
        if (res_eq <= a.tol_eq && res_in <= a.tol_ineq && rs <= a.tol_stat && rcomp <= a.tol_comp) { flag = FRP_EXIT_OPTIMAL; break; }
        if (it >= a.maxit) { flag = FRP_EXIT_MAXIT; break; }
        if (mu > DIVERGE_MU * fmax(1.0, a.mu0) || rs > DIVERGE_RS) { flag = FRP_EXIT_NOPROGRESS; break; }
        TOCK(0);

        // predictor (affine) solve; exact Hessian first, Gauss-Newton if the reduced Hessian is indefinite
        int theta = hess;
        int fr = sweep_factor<NP>(w, xinit, N, theta);
        if (fr && theta) {
            theta = 0;
            nfallback++;
            fr = sweep_factor<NP>(w, xinit, N, 0);
        }
        if (fr) { flag = FRP_EXIT_FACTORIZATION; break; }
        TOCK(1);
        sweep_forward<NP>(w, N);
        TOCK(2);
        const SlackOut s0 = phase_affine<NP>(w, pbase, np, N, MF, nfk, a.model, mu, mtot, a.tol_comp);
        sigma = s0.sigma;
        TOCK(3);
        // corrector solve (same factorisation, new rhs)
        sweep_backvec<NP>(w, xinit, N, s0.smu);
        TOCK(4);
        sweep_forward<NP>(w, N);
        TOCK(2);
        const SlackOut s1 = phase_step<NP>(w, N, MF, nfk, s0.smu, a.ftb);
        step_cc = s1.ap;
        TOCK(3);
        sweep_costate<NP>(w, N, s1.ap, s0.smu, theta);
        TOCK(5);
    }

    // ---------------------------------------------------------------- outputs
    FULLSYNC();
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
            o[0] = res_eq; o[1] = res_in; o[2] = rs; o[3] = rcomp; o[4] = pobj; o[5] = mu; o[6] = step_cc; o[7] = (double)nfallback;
#ifdef FRP_PROFILE
            for (int i = 0; i < 6; i++) o[i] = (double)tph[i]; // cycles: eval, factor, forward(x2), slack(x2), backvec, costate
#endif
        }
    }
    (void)sigma;
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

#ifdef FRP_PROFILE
void debug_read_prof(long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 24);
    long long z[24] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
}
#endif

} // namespace frp

#ifdef FRP_PROFILE
extern "C" void frp_debug_read_prof(long long *out) { frp::debug_read_prof(out); }
#endif
