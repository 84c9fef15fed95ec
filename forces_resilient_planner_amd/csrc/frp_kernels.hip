// frp_kernels.hip -- gfx950 kernels of the batched NMPC solver (hand-written HIP, FP64).
//
// nmpc_ipm_kernel: one 64-lane wavefront == one workgroup == one NMPC problem at a time, resident for the whole
// interior-point solve (no host round trips, per-problem early exit).  Persistent workgroups pull the next problem
// from a device queue that is ordered longest-expected-solve first.  It replaces the reference's closed NLP solver
//   FORCESNLPsolver_{normal,final}_solve  (FORCESNLPsolver_normal.h:323; forces_normal.cpp:139)
// and its model callback FORCESNLPsolver_*_casadi2forces (casadi2forces.c:42-245).
//
// Iteration (same as the CPU oracle so both can be compared iterate by iterate):
//   primal-dual interior point, Mehrotra predictor-corrector, multiplier safeguard s_i lam_i >= mu / 2,
//   stage Hessian = exact cost Hessian + exact Hessian of the RK2 dynamics (Gauss-Newton fallback when the reduced
//   Hessian is indefinite), Newton KKT system solved by a Riccati recursion over the stage chain with
//   state s = [w; x] (13) and control u (4):   s_{k+1} = [u_k; A_k x_k + B_k u_k] + d_k.
//
// Work distribution inside the wavefront
//   * element-wise phases (residuals, barrier terms, step lengths, updates): all 64 lanes,
//     lane = (row group, stage), operands in [row][stage] arrays -> coalesced 64-lane accesses;
//   * model evaluation (RK2 step, Jacobian, exact Hessian): lane == stage;
//   * factorisation sweep: every 13x13(+1) block lives in REGISTERS as a 16x16 FP64 tile in the
//     v_mfma_f64_16x16x4_f64 accumulator layout (lane (g,c), register r <-> element [4r+g][c]).  In that
//     layout D = X'Y is four MFMAs with A := X, B := Y register for register, so
//       X = P M,  G = M'X + Phi,  T = R G_u,  S = G - G_u' T,  P <- S (+ w blocks)
//     runs on the matrix pipe with no cross-lane data movement; the right-hand side rides along as column 13;
//   * vector sweeps (forward, corrector backward): chained mat-vec products on v_mfma_f64_4x4x4_4b_f64 (matvec4).
//   Per stage the sweeps stream 64-lane record rows from/to HBM, prefetched two steps ahead.
#include <hip/hip_runtime.h>
#include <math.h>
#include "frp_model.hpp"
#include "../../include/frp_nmpc.h"
#include "frp_kernels.h"
#include "frp_device.hpp"
#include <cstdlib>

namespace frp {

// occupancy target: waves per SIMD the register allocator must leave room for (propagates to the
// non-inlined phase functions)
#ifndef FRP_WAVES_PER_EU
#define FRP_WAVES_PER_EU 2
#endif
#ifndef FRP_PRIO_IT1 // iteration counts at which a long solve raises its wave priority
#define FRP_PRIO_IT1 7
#define FRP_PRIO_IT2 10
#define FRP_PRIO_IT3 14
#endif
#ifndef FRP_RB_MAX_NP
#define FRP_RB_MAX_NP 32 // largest stage stride whose model phase goes through the LDS transposition buffer
#endif
#ifndef FRP_SLOTS_PER_CU
#define FRP_SLOTS_PER_CU 6
#endif
#define FRP_MAX_SLOTS 4096 // upper bound of resident single-wave workgroups the workspace is sized for


// ------------------------------------------------------------------ LDS layout (doubles)
// [0, 736): staging.  Element-wise phases: gm[17][NP] + corridor sums[6][NP] (NP = 32).
//           Riccati sweeps: the E part of the current stage record + constants + T' + pivot factors + packed P.
// [736, ...): fields that live across phases of one iteration.
constexpr int S_E = 0;                    // E part of the stage record (192 used)
constexpr int S_ZERO = 248, S_ONE = 249, S_DTC = 250;
constexpr int S_T = 256;                  // T' (64)
constexpr int S_MI = 320, S_DI = 328;      // pivot block handed from uniform registers to lanes: m = L^-1 (6), D^-1 (4)
constexpr int S_PRAW = 336;               // forward sweep: staged copy of the next stage's packed P (96) + p (16)
constexpr int S_M = 560;                  // staged M row of the next stage: record slots 0..63 (LIN | d) + the constants 0, 1, dt (67)
constexpr int MS_ZERO = 64, MS_ONE = 65, MS_DT = 66; // row-space slots of the constants
constexpr int S_FOUT = 448;               // factorisation sweep: P d (16) + packed P (96) of the current stage before they are stored
constexpr int S_STAGING = 23 * 32;        // 736
constexpr int S_RW = S_STAGING;           // stage-0 solve: Pww^-1 (16)
constexpr int S_PWX = S_RW + 16;          // stage-0 solve: Pwx (4 x 9)
constexpr int S_DS0 = S_PWX + 36;         // ds_0 = [dw_0; dx_0] (13, padded 16)
constexpr int L_TOTAL = S_DS0 + 16;



struct WsView {
    gdouble *rec, *z, *y, *pre, *s, *lam, *corr, *face, *step; // pre: the 10 leading stage parameters, [row][stage]
};

// stage stride NP of the [row][stage] arrays = lanes per row group of the element-wise phases (H = 64 / NP groups)
__host__ __device__ inline int padded_stages(int N) { return N <= 16 ? 16 : (N <= 20 ? 20 : (N <= 32 ? 32 : 64)); }

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
__device__ __forceinline__ WsView uni(WsView w)
{
    w.rec = uni(w.rec); w.z = uni(w.z); w.y = uni(w.y); w.pre = uni(w.pre);
    w.s = uni(w.s); w.lam = uni(w.lam); w.corr = uni(w.corr); w.face = uni(w.face); w.step = uni(w.step);
    return w;
}

__shared__ double sm[L_TOTAL];
// Newton step dz = [du(4); ds(13); 3 pad rows][NP]: written by the forward sweep, read by the step phases
// -- kept in LDS so that the sweeps carry no global stores for it (a store in the loop makes the
// staging write of the next stage wait for vmcnt(0))
__shared__ double sm_dz16[DZ_ROWS * 16];
__shared__ double sm_dz20[DZ_ROWS * 20];
__shared__ double sm_dz32[DZ_ROWS * 32];
__shared__ double sm_dz64[DZ_ROWS * 64];
// Transposition buffer [stage][RB_LD] for record rows that are PRODUCED lane == stage (or lane == (row, stage)) but live
// in per-stage records: written to LDS first, then flushed with one coalesced 64-lane store per stage instead of one
// 8-byte store per lane into 20+ different cache lines (NP = 64 keeps the direct stores: its LDS budget is spent).
constexpr int RB_LD = 65; // odd leading dimension: conflict-free both ways
__shared__ double sm_rb16[16 * RB_LD];
__shared__ double sm_rb20[20 * RB_LD];
__shared__ double sm_rb32[32 * RB_LD];
template <int NP>
__device__ __forceinline__ double *rb_area() { return NP == 16 ? sm_rb16 : (NP == 20 ? sm_rb20 : sm_rb32); }
// M row (record slots 0..63 + constants) of stage ks as the sweeps gather it: the staged copy in `sm` (stage-independent)
template <int NP>
__device__ __forceinline__ const double *m_row(int ks) { (void)ks; return sm + S_M; }
template <int NP>
__device__ __forceinline__ double *dz_area() { return NP == 16 ? sm_dz16 : (NP == 20 ? sm_dz20 : (NP == 32 ? sm_dz32 : sm_dz64)); }

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


// Row-space slot of Mt[row][col], the augmented transition matrix: a slot of the stage's M row = record slots 0..63
// (compact linearisation | d) followed by the constants 0, 1, dt
//   rows: s+ = [w+(0..3); x+(4..12)],  cols: [u(0..3); x(4..12); 13 = d]
//   w+ = u + d_w,  x+ = A x + B u + d_x
__device__ __forceinline__ int m_src(int row, int col)
{
    if (row > 12 || col > 13) return MS_ZERO;
    if (col == 13) return REC_D + row;
    if (row < 4) return (col == row) ? MS_ONE : MS_ZERO;
    const int i = row - 4, bi = i / 3, ii = i % 3;
    if (col < 4) { // B[i][col]
        if (col == 3) return bi == 0 ? REC_LIN + 36 + ii : (bi == 1 ? REC_LIN + 39 + ii : MS_ZERO);
        if (bi == 1) return REC_LIN + 42 + ii * 3 + col;
        if (bi == 2) return ii == col ? MS_DT : MS_ZERO;
        return MS_ZERO;
    }
    const int j = col - 4, bj = j / 3, jj = j % 3;
    if (bi == 0) return bj == 0 ? (ii == jj ? MS_ONE : MS_ZERO) : REC_LIN + (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? MS_ZERO : REC_LIN + (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? MS_ONE : MS_ZERO;
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
    if (hr >= 0 && hc_ >= 0 && hd_index(hr, hc_) >= 0) o3 = S_E + REC_HD + hd_index(hr, hc_);
}

__device__ __forceinline__ void init_stage_constants(int lane)
{
    if (lane == 0) {
        sm[S_ZERO] = 0.0; sm[S_ONE] = 1.0; sm[S_DTC] = DT;
        sm[S_M + MS_ZERO] = 0.0; sm[S_M + MS_ONE] = 1.0; sm[S_M + MS_DT] = DT;
    }
}

// Per-lane gather offsets of the register tiles (which LDS / record slot feeds tile element (4r+g, c)): they depend
// on the lane only, so the persistent workgroup computes them once into LDS and every sweep reloads its 4..16
// entries with a few ds_reads instead of re-deriving them with ~700 branchy integer instructions per sweep.
constexpr int TAB_M = 0, TAB_C1 = 4, TAB_C2 = 8, TAB_C3 = 12; // 16x16 tiles of the factorisation sweep: element (4r+g, c)
constexpr int TAB4_MT = 16;  // 4x4x4 A-operand layout: Mt[row][col]            (forward sweep, ds+ = Mt [du; dx; 1])
constexpr int TAB4_MTT = 20; //                         Mt[col][row]            (vector backward sweep, q~ = phi~ + Mt' x)
constexpr int TAB4_TT = 24;  //                         T'[row][col], row < 4   (forward sweep, du = -T' [hc dw; dx; 1])
constexpr int TAB4_P = 28;   //                         P_k[row][col] from its packed lower triangle (record offset)
constexpr int TAB_PP = 32;   // LDS slot (S_FOUT) of packed P_k element (4r+g, c), or its dump slot: the writes are unconditional
constexpr int TAB_PD = 36;   // LDS slot (S_FOUT) of (P d)[4r+g] for the lanes of column 13, dump slot elsewhere
constexpr int TAB_ROWS = 40;
__shared__ unsigned short sm_tab[TAB_ROWS * 64];
__device__ __noinline__ void init_lane_tables()
{
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int qk = lane >> 4, qI = (lane >> 2) & 3, qj = lane & 3;
    for (int r = 0; r < 4; r++) {
        int o1, o2, o3;
        c_src(4 * r + g, c, o1, o2, o3);
        sm_tab[(TAB_M + r) * 64 + lane] = (unsigned short)m_src(4 * r + g, c);
        sm_tab[(TAB_C1 + r) * 64 + lane] = (unsigned short)o1;
        sm_tab[(TAB_C2 + r) * 64 + lane] = (unsigned short)o2;
        sm_tab[(TAB_C3 + r) * 64 + lane] = (unsigned short)o3;
        const int row = 4 * qI + qj, col = 4 * ((qI + r) & 3) + qk;
        sm_tab[(TAB4_MT + r) * 64 + lane] = (unsigned short)m_src(row, col);
        sm_tab[(TAB4_MTT + r) * 64 + lane] = (unsigned short)m_src(col, row);
        sm_tab[(TAB4_TT + r) * 64 + lane] = (unsigned short)(row < 4 ? S_T + 16 * row + col : S_ZERO);
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        sm_tab[(TAB4_P + r) * 64 + lane] = (unsigned short)((row <= 12 && col <= 12) ? REC_P + hi * (hi + 1) / 2 + lo : REC_ZERO);
        const int trow = 4 * r + g; // 16x16 tile element (trow, c)
        sm_tab[(TAB_PP + r) * 64 + lane] = (unsigned short)(S_FOUT + 16 + ((trow <= 12 && c <= trow) ? trow * (trow + 1) / 2 + c : 95));
        sm_tab[(TAB_PD + r) * 64 + lane] = (unsigned short)(S_FOUT + ((c == 13 && trow <= 12) ? trow : 16 + 94));
    }
    __syncthreads();
}

struct EvalOut {
    double eq, in, rs, rc, gap, obj;
};

// Staging area of the element-wise phases (aliases the sweep staging, which is dead then):
// gm[17][NP] multiplier part of the stationarity residual, gf[6][NP] corridor sums for pos entries.
__shared__ double sm_big[23 * 64]; // only referenced (hence only allocated) by the NP = 64 instantiation
template <int NP>
__device__ __forceinline__ double *stage_area() { return NP <= 32 ? sm : sm_big; }

// Lane groups of the element-wise phases: lane = sub * NP + k with sub in [0, H), H = 64 / NP (NP = 20: lanes
// 60..63 idle).  xsub_sum adds the values of the H lanes that share a stage k (result valid in the sub == 0 lanes).
template <int NP>
__device__ __forceinline__ double xsub_sum(double v)
{
    constexpr int H = 64 / NP;
    if (H == 2) return v + __shfl_xor(v, 32);
    if (H == 4) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
    if (H == 3) {
        const int lane = threadIdx.x;
        const double a = __shfl(v, lane + NP < 64 ? lane + NP : lane), b = __shfl(v, lane + 2 * NP < 64 ? lane + 2 * NP : lane);
        return v + a + b;
    }
    return v;
}
// value of a per-row constant for row i = r*H + sub, chosen among the H compile-time candidates of round r
#define ROW_PICK(expr_of_i)                                                                         \
    ([&]() {                                                                                          \
        double v_ = [&](int i) { return (double)(expr_of_i); }(ib < NZ ? ib : NZ - 1);                 \
        if (H > 1 && half == 1) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 1 < NZ ? ib + 1 : NZ - 1); \
        if (H > 2 && half == 2) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 2 < NZ ? ib + 2 : NZ - 1); \
        if (H > 3 && half == 3) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 3 < NZ ? ib + 3 : NZ - 1); \
        return v_;                                                                                    \
    }())

// part 2 of the evaluation phase (see phase_eval); arrays as __restrict__ parameters so that the loads of
// several row rounds can be batched across the record stores
template <int NP>
__device__ __forceinline__ void eval_rows(cgdouble *__restrict__ ps, cgdouble *__restrict__ pl, cgdouble *__restrict__ pz,
                                          cgdouble *__restrict__ pface, gdouble *__restrict__ prec, cgdouble *__restrict__ ppre,
                                          int N, int MF, int nfk, int model, double *stg,
                                          double &l_in, double &l_rc, double &l_gap, double &l_rs)
{
    constexpr int H = 64 / NP;
    const int lane = threadIdx.x;
    // ---- part 2: all 64 lanes, lane = (half, stage k); rows handled in pairs
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N && half < H;
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
                const double sg = lc * fast_rcp(sc), t = sg * rc;
                fp0 += a0 * t; fp1 += a1 * t; fp2 += a2 * t;
                p0 += sg * a0 * a0; p1 += sg * a0 * a1; p2 += sg * a0 * a2;
                p3 += sg * a1 * a1; p4 += sg * a1 * a2; p5 += sg * a2 * a2;
            }
        }
        if (H > 1) {
            gp0 = xsub_sum<NP>(gp0); gp1 = xsub_sum<NP>(gp1); gp2 = xsub_sum<NP>(gp2);
            fp0 = xsub_sum<NP>(fp0); fp1 = xsub_sum<NP>(fp1); fp2 = xsub_sum<NP>(fp2);
            p0 = xsub_sum<NP>(p0); p1 = xsub_sum<NP>(p1); p2 = xsub_sum<NP>(p2);
            p3 = xsub_sum<NP>(p3); p4 = xsub_sum<NP>(p4); p5 = xsub_sum<NP>(p5);
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
        double pc[NPRE]; // ref(3), weights(3), yaw_ref from the transposed copy (coalesced; the parameter rows are 1 KB apart)
        pc[0] = ppre[0 * NP + k]; pc[1] = ppre[1 * NP + k]; pc[2] = ppre[2 * NP + k];
        pc[6] = ppre[6 * NP + k]; pc[7] = ppre[7 * NP + k]; pc[8] = ppre[8 * NP + k]; pc[9] = ppre[9 * NP + k];
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        gdouble *rec = prec + (size_t)k * REC_STRIDE;
        constexpr int R = (NZ + H - 1) / H;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int ib = r * H, i = ib + half;
            if (i >= NZ) continue;
            const double hd = ROW_PICK(cq.hd(i));
            const double qi = ROW_PICK(cq.q(i));
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
            const double zi = pz[i * NP + k];
            double cg = hd * zi + qi; // cost gradient
            if (ib < 8) cg += (i < 8 ? cq.hc() : 0.0) * pz[(i < 4 ? i + 4 : (i < 8 ? i - 4 : i)) * NP + k];
            const double sl = ps[i * NP + k], su = ps[(17 + i) * NP + k];
            const double ll = pl[i * NP + k], lu = pl[(17 + i) * NP + k];
            const double vl = lb - zi, vu = zi - ub;
            const double rl = vl + sl, ru = vu + su;
            l_in = fmax(l_in, fmax(fmax(vl, vu), fmax(fabs(rl), fabs(ru))));
            l_rc = fmax(l_rc, fmax(sl * ll, su * lu));
            l_gap += sl * ll + su * lu;
            const double sgl = ll * fast_rcp(sl), sgu = lu * fast_rcp(su);
            double gi = cg + stg[i * NP + k] + lu - ll;
            double ph = cg + sgu * ru - sgl * rl;
            if (ib + H > 8 && ib < 11) {
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
// part 2 (lane == (row group, stage), all 64 lanes): corridor rows, then bounds: residual norms,
//        barrier Hessian / affine rhs -> record
struct ModelOut {
    double eq, obj;
};
// part 1 (lane == stage): model + linearisation -> record, equality residuals, M'y -> LDS staging.  A function of its
// own: it needs most of the register file, and the element-wise part that follows has stage-divergent loops.
template <int NP>
__device__ __noinline__ ModelOut phase_model(WsView w, cgdouble *xinit, int N, int model, int hess)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); model = uni(model); hess = uni(hess);
    FULLSYNC(); // phase boundary: other lanes' global writes of the previous phase are visible
    const int lane = threadIdx.x;
    double *stg = stage_area<NP>();
    double l_eq = 0, l_obj = 0;
    constexpr bool BUF = NP <= FRP_RB_MAX_NP; // record rows go through the LDS transposition buffer (see rb_area)
    double *rb = rb_area<NP>();
    // kept for the Hessian pass that follows the flush of the linearisation
    AccJac J1;
    Trig tg1 = {0, 0, 0, 0, 0, 0}, tg2 = {0, 0, 0, 0, 0, 0};
    double vt[3] = {0, 0, 0}, vk[3] = {0, 0, 0}, ypv[6] = {0, 0, 0, 0, 0, 0}, Tk = 0.0;
    if (lane < N) {
        const int k = lane;
        double p10[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; i++) p10[i] = w.pre[i * NP + k];
        const int sc_k = stage_class(k, N);
        double zk[NZ];
#pragma unroll
        for (int i = 0; i < NZ; i++) zk[i] = w.z[i * NP + k];
        l_obj = stage_cost(zk, p10, sc_k, model, nullptr);
        gdouble *rec = w.rec + (size_t)k * REC_STRIDE;
        double *rbk = rb + k * RB_LD;
        auto put = [&](int slot, double val) { if (BUF) rbk[slot] = val; else rec[slot] = val; };
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
            AccJac J2;
            double a1[3], a2[3], et[3];
            tg1 = make_trig(zk + 14);
            accel_t<true>(zk + 11, tg1, zk[3], p10 + 3, a1, &J1);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                vt[i] = zk[11 + i] + DT * a1[i];
                et[i] = zk[14 + i] + DT * zk[i];
            }
            tg2 = make_trig(et);
            accel_t<true>(vt, tg2, zk[3], p10 + 3, a2, &J2);
#pragma unroll
            for (int i = 0; i < 3; i++) { vk[i] = zk[11 + i]; ypv[i] = yp[i]; ypv[3 + i] = yv[i]; }
            Tk = zk[3];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double d = zk[i] - w.z[(4 + i) * NP + k + 1];
                put(REC_D + i, d);
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
                put(REC_D + 4 + i, dp); put(REC_D + 7 + i, dv); put(REC_D + 10 + i, de);
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
                    put(REC_LIN + i * 3 + j, apv);
                    put(REC_LIN + 9 + i * 3 + j, ape);
                    put(REC_LIN + 18 + i * 3 + j, avv);
                    put(REC_LIN + 27 + i * 3 + j, ave);
                    put(REC_LIN + 42 + i * 3 + j, bvw);
                    gm[j] += bvw * yv[i];
                    gm[11 + j] += apv * yp[i] + avv * yv[i];
                    gm[14 + j] += ape * yp[i] + ave * yv[i];
                    sT += DT * J2.Fvv[i * 3 + j] * J1.gT[j];
                }
                const double bpt = 0.5 * DT * DT * J1.gT[i];
                const double bvt = 0.5 * DT * (J1.gT[i] + sT);
                put(REC_LIN + 36 + i, bpt);
                put(REC_LIN + 39 + i, bvt);
                gT += bpt * yp[i] + bvt * yv[i];
            }
            gm[3] += gT;
        }
#pragma unroll
        for (int i = 0; i < NZ; i++) stg[i * NP + k] = gm[i];
    }
    WSYNC();
    if (BUF) { // flush the linearisation rows (record slots 0..63) of the stages that have dynamics: one 512-byte store each
        for (int k = 0; k < N - 1; k++) w.rec[(size_t)k * REC_STRIDE + lane] = rb[k * RB_LD + lane];
        WSYNC();
    }
    if (hess) {
        if (lane < N - 1) {
            // exact Hessian of y_{k+1}' c(z_k): only the pos / vel rows of the RK2 step are non-linear
            gdouble *rec = w.rec + (size_t)lane * REC_STRIDE;
            double *rbk = rb + lane * RB_LD;
            rk2_hessian_core(vk, Tk, J1, vt, tg1, tg2, ypv, ypv + 3, [&](int i, int j, double val) {
                if (hd_index(i, j) >= 0) { if (BUF) rbk[hd_index(i, j)] = val; else rec[REC_HD + hd_index(i, j)] = val; }
            });
        }
        if (BUF) {
            WSYNC();
            const int slot = lane < REC_HD_SIZE ? REC_HD + lane : REC_E_SIZE - 1; // other lanes: the pad slot (all lanes stay active)
            for (int k = 0; k < N - 1; k++) w.rec[(size_t)k * REC_STRIDE + slot] = rb[k * RB_LD + lane];
        }
    }
    WSYNC();
    ModelOut o;
    o.eq = l_eq; o.obj = l_obj;
    return o;
}

// part 2 (lane == (row group, stage), all 64 lanes): corridor rows, then bounds: residual norms, barrier Hessian /
// affine rhs -> record
template <int NP>
__device__ __noinline__ EvalOut phase_eval(WsView w, int N, int MF, int nfk, int model, double l_eq, double l_obj)
{
    w = uni(w); N = uni(N); MF = uni(MF); model = uni(model);
    WSYNC();
    double *stg = stage_area<NP>();
    double l_in = 0, l_rs = 0, l_rc = 0, l_gap = 0;
    eval_rows<NP>(w.s, w.lam, w.z, w.face, w.rec, w.pre, N, MF, nfk, model, stg, l_in, l_rc, l_gap, l_rs);
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
// Software pipeline as in the other sweeps: while the MFMA chain of stage k executes, the wave stages the
// (prefetched) record of stage k-1 through LDS and assembles its tiles into the alternate register set, and issues
// the global prefetch of stage k-2.
template <int NP>
__device__ __forceinline__ bool factor_step(const WsView &w, int kk, bool last, int lane, int g, int c, double theta, int mgo, int mco,
                                            const int (&ppo)[4], const int (&pdo)[4],
                                            const int (&mo)[4], const int (&c1)[4], const int (&c2)[4], const int (&c3)[4],
                                            const d4 &cC, const d4 &cM, double chc, double cPhiDw, double cphiw,
                                            d4 &nC, d4 &nM, double &nhc, double &nPhiDw, double &nphiw,
                                            double &e0, double &e1, double &e2, d4 &P, d4 &pv)
{
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    gdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
    d4 G = cC;
    if (!last) {
        d4 X = mm_tn(P, cM, zero);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sm[pdo[r]] = X[r]; // P d (column 13; every other lane writes the dump slot), flushed with the packed P below
            X[r] += pv[r];     // pv is zero outside column 13
        }
        G = mm_tn(cM, X, cC);
    }
    // ---- between / behind the MFMAs: tiles of stage kk-1 (clamped at 0: the tail re-stages stage 0, unused)
    WSYNC();
    sm[S_M + lane] = e0; sm[S_E + 64 + lane] = e1; sm[S_E + 128 + lane] = e2;
    WSYNC();
    { // this register set is staged again two steps from now: the loads have two full steps to land
        const int k3 = kk > 2 ? kk - 3 : 0;
        cgdouble *r3 = w.rec + (size_t)k3 * REC_STRIDE;
        e0 = r3[lane]; e1 = r3[64 + lane]; e2 = r3[128 + lane];
    }
    const int ks = kk > 0 ? kk - 1 : 0; // the stage whose tiles are assembled now
#pragma unroll
    for (int r = 0; r < 4; r++) {
        nC[r] = sm[c1[r]] + sm[c2[r]] + theta * sm[c3[r]];
        nM[r] = m_row<NP>(ks)[mo[r]];
    }
    nhc = sm[S_E + REC_HC];
    nPhiDw = sm[S_E + REC_PHID + 4 + g];
    nphiw = sm[S_E + REC_PHI + 4 + g];
    // ---- pivot block Guu = L D L' (4 x 4): gather the lower triangle to uniform registers, factor redundantly
    double q[16], Mi[6], Di[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(G[0], 16 * i + j);
#ifdef FRP_DEBUG_FACTOR
    if (lane == 0) printf("gpu stage %2d theta %.3g Guu %.9e %.9e %.9e %.9e | %.6e %.6e %.6e %.6e %.6e %.6e\n", kk, theta, q[0], q[5], q[10], q[15], q[4], q[8], q[9], q[12], q[13], q[14]);
#endif
    if (!ldl4(q, Mi, Di)) return false;
    // uniform values handed to the lanes through LDS (every lane writes the same value to the same slot: no branch)
#pragma unroll
    for (int t = 0; t < 6; t++) sm[S_MI + t] = Mi[t];
#pragma unroll
    for (int t = 0; t < 4; t++) sm[S_DI + t] = Di[t];
    WSYNC();
    // m = L^-1.  The elimination is carried out in factored form,
    //     K = m G_u,   R = m' D^-1 m,   T = m' D^-1 K (= R G_u),   TT = K' D^-1 m (= G_u' R),   S = G - K' D^-1 K,
    // not as G - G_u'(R G_u): when a state bound far down the horizon is active, B'PB puts a rank-one term of 1e10 on
    // Guu and G_u; the explicit-inverse form then cancels O(1e10) quantities against an R that is only accurate to
    // cond(Guu) eps and S comes out with O(100) errors (seen as a spurious indefinite pivot), whereas the factored form
    // subtracts a symmetric product and is as accurate as a Cholesky-based elimination.
    const double m_gc = sm[mgo], m_cg = sm[mco]; // m[g][c], m[c][g] (unit diagonal / zeros from the constant slots)
    const double dg = sm[S_DI + g];
    const double hc = chc;
    const double md = dg * m_gc;
    const d4 K = mm_tn4(m_cg, G[0], zero);
    const d4 Rt = mm_tn4(m_gc, md, zero);
    const double rt = Rt[0]; // R[g][c] in the lanes c < 4 (zero elsewhere)
    const double Kd = dg * K[0];
    const d4 T = mm_tn4(m_gc, Kd, zero);
    const d4 TT = mm_tn4(K[0], md, zero);
    const d4 S = mm_tn4(-Kd, K[0], G);
    rec[REC_T + lane] = (c < 4) ? rt : (c <= 13 ? T[0] : (lane == 14 ? hc : 0.0));
    d4 Pn, pn;
    Pn[0] = (c < 4) ? ((g == c ? cPhiDw : 0.0) - hc * hc * rt) : (c <= 12 ? -hc * T[0] : 0.0);
    pn[0] = (c == 13) ? (cphiw - hc * T[0]) : 0.0;
#pragma unroll
    for (int r = 1; r < 4; r++) {
        const bool inb = (4 * r + g) <= 12;
        Pn[r] = (inb && c <= 12) ? (c < 4 ? -hc * TT[r] : S[r]) : 0.0;
        pn[r] = (inb && c == 13) ? S[r] : 0.0;
    }
    P = Pn;
    pv = pn;
    // P_k (packed lower triangle) for the multiplier recovery y_k = P_k ds_k + p_k in the forward sweep
#pragma unroll
    for (int r = 0; r < 4; r++) sm[ppo[r]] = Pn[r]; // lanes outside the lower triangle write the dump slot
    // [P d | packed P] leave as two coalesced stores (the record keeps them adjacent)
    WSYNC();
    rec[REC_PD + lane] = sm[S_FOUT + lane];
    if (lane < 48) rec[REC_PD + 64 + lane] = sm[S_FOUT + 64 + lane];
    return true;
}

template <int NP>
__device__ __noinline__ int sweep_factor(WsView w, cgdouble *xinit, int N, double theta)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); theta = uni(theta);
    FULLSYNC(); // phase boundary: the evaluation phase's record writes are visible
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    int mo[4], c1[4], c2[4], c3[4], ppo[4], pdo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = sm_tab[(TAB_M + r) * 64 + lane];
        c1[r] = sm_tab[(TAB_C1 + r) * 64 + lane];
        c2[r] = sm_tab[(TAB_C2 + r) * 64 + lane];
        c3[r] = sm_tab[(TAB_C3 + r) * 64 + lane];
        ppo[r] = sm_tab[(TAB_PP + r) * 64 + lane];
        pdo[r] = sm_tab[(TAB_PD + r) * 64 + lane];
    }
    init_stage_constants(lane); // the element-wise phases reuse this part of LDS as staging
    // LDS slot of m[g][c] / m[c][g] for this lane (m = L^-1 of the pivot block, strictly lower part in S_MI)
    const int mgo = c < 4 ? (c < g ? S_MI + g * (g - 1) / 2 + c : (c == g ? S_ONE : S_ZERO)) : S_ZERO;
    const int mco = c < 4 ? (g < c ? S_MI + c * (c - 1) / 2 + g : (c == g ? S_ONE : S_ZERO)) : S_ZERO;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 P = zero, pv = zero;
    bool ok = true;
    double e0, e1, e2, f0, f1, f2; // two prefetch sets: (e*) staged by the odd steps, (f*) by the even ones
    d4 CA, MA, CB = zero, MB = zero;
    double hcA, PhiDwA, phiwA, hcB = 0.0, PhiDwB = 0.0, phiwB = 0.0;
    { // prologue: tiles of stage N-1 into set A, prefetch of stages N-2 and N-3
        cgdouble *rp = w.rec + (size_t)(N - 1) * REC_STRIDE;
        e0 = rp[lane]; e1 = rp[64 + lane]; e2 = rp[128 + lane];
        cgdouble *r3 = w.rec + (size_t)(N > 2 ? N - 3 : 0) * REC_STRIDE;
        f0 = r3[lane]; f1 = r3[64 + lane]; f2 = r3[128 + lane];
        WSYNC();
        sm[S_M + lane] = e0; sm[S_E + 64 + lane] = e1; sm[S_E + 128 + lane] = e2;
        WSYNC();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            CA[r] = sm[c1[r]] + sm[c2[r]] + theta * sm[c3[r]];
            MA[r] = m_row<NP>(N - 1)[mo[r]];
        }
        hcA = sm[S_E + REC_HC];
        PhiDwA = sm[S_E + REC_PHID + 4 + g];
        phiwA = sm[S_E + REC_PHI + 4 + g];
        cgdouble *r2 = w.rec + (size_t)(N > 1 ? N - 2 : 0) * REC_STRIDE;
        e0 = r2[lane]; e1 = r2[64 + lane]; e2 = r2[128 + lane];
    }
    int kk = N - 1;
    for (; kk >= 1 && ok; kk -= 2) {
        ok = factor_step<NP>(w, kk, kk == N - 1, lane, g, c, theta, mgo, mco, ppo, pdo, mo, c1, c2, c3, CA, MA, hcA, PhiDwA, phiwA, CB, MB, hcB, PhiDwB, phiwB, e0, e1, e2, P, pv);
        if (!ok) break;
        ok = factor_step<NP>(w, kk - 1, false, lane, g, c, theta, mgo, mco, ppo, pdo, mo, c1, c2, c3, CB, MB, hcB, PhiDwB, phiwB, CA, MA, hcA, PhiDwA, phiwA, f0, f1, f2, P, pv);
    }
    if (ok && kk == 0) ok = factor_step<NP>(w, 0, N == 1, lane, g, c, theta, mgo, mco, ppo, pdo, mo, c1, c2, c3, CA, MA, hcA, PhiDwA, phiwA, CB, MB, hcB, PhiDwB, phiwB, e0, e1, e2, P, pv);
    bool fail = !ok;
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
// p_x = q~_x - Kbar' q_u,  p_w = phi_w - hc kbar.  Updates the kbar column of T' and stores p_k.
// All mat-vec products on the 4x4x4 MFMA, vectors in V layout: q~ = matvec4(Mt', x, phi~); the second product
// contracts over the 4 inputs only, i.e. ONE MFMA whose A operand is the T' record exactly as it is stored
// (lane 16k + c <-> T'[k][c]) and whose B operand is q_u broadcast from quad 0 to all quads of each row.
// Same software pipeline as the forward sweep: LDS-staged operands (Mt', phi) are gathered one stage ahead while the
// MFMAs execute, register operands (T', P d) are loaded one stage ahead straight into the alternate register set.
template <int NP>
__device__ __forceinline__ void backvec_step(const WsView &w, int kk, bool last, int lane, int idx, double smu,
                                             const int (&mo)[4], int pho, int pwo, int pdo,
                                             const d4 &cM, double cphi, double chc, double cphiw, double &ctp, double &cpd,
                                             d4 &nM, double &nphi, double &nhc, double &nphiw,
                                             double &e0, double &e1, double &e2, double &pv)
{
    gdouble *rec = w.rec + (size_t)kk * REC_STRIDE;
    double q = cphi;
    if (!last) q = matvec4(cM, cpd + pv, cphi);
    // ---- while the MFMAs execute: operands of stage kk-1 (clamped at 0: the tail re-stages stage 0, unused)
    WSYNC();
    sm[S_M + lane] = e0; sm[S_E + 64 + lane] = e1;
    if (lane < 14) sm[S_E + 128 + lane] = e2;
    WSYNC();
    { // staged again two steps from now
        const int k3 = kk > 2 ? kk - 3 : 0;
        cgdouble *r3 = w.rec + (size_t)k3 * REC_STRIDE;
        e0 = r3[lane]; e1 = r3[64 + lane]; e2 = r3[128 + (lane < 14 ? lane : 0)];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) nM[r] = m_row<NP>(kk > 0 ? kk - 1 : 0)[mo[r]];
    nphi = sm[S_E + REC_PHIB + pho] + smu * sm[S_E + REC_PHIC + pho];
    nphi = (idx <= 12) ? nphi : 0.0;
    nphiw = sm[S_E + REC_PHIB + pwo] + smu * sm[S_E + REC_PHIC + pwo];
    nhc = sm[S_E + REC_HC];
    // q_u (rows 0..3, quad 0) to every quad of its row, then E[c] = sum_k T'[k][c] q_u[k]
    const int qI = (lane >> 2) & 3;
    const double r1 = quad_rot<1>(q), r2 = quad_rot<2>(q), r3 = quad_rot<3>(q);
    const double qu = qI == 0 ? q : (qI == 1 ? r3 : (qI == 2 ? r2 : r1));
    const double E = mfma4(ctp, qu, 0.0);
    { // register operands (T', P d) of this set's next stage, two steps from now
        const int k2 = kk > 1 ? kk - 2 : 0;
        cgdouble *r2 = w.rec + (size_t)k2 * REC_STRIDE;
        ctp = r2[REC_T + lane];
        cpd = r2[pdo];
    }
    const bool q0 = idx < 4;
    const double pn = q0 ? cphiw - chc * E : (idx <= 12 ? q - E : 0.0);
    if ((lane & 3) == 0) {
        if (q0) rec[REC_T + 16 * idx + 13] = E; // kbar
        rec[REC_PV + idx] = pn;                 // p_k for y_k = P_k ds_k + p_k (rows 13..15: zero padding)
    }
    pv = pn;
}

template <int NP>
__device__ __noinline__ void sweep_backvec(WsView w, cgdouble *xinit, int N, double smu)
{
    w = uni(w); xinit = uni(xinit); N = uni(N); smu = uni(smu);
    FULLSYNC(); // phase boundary: the corrector rhs written by the step phase is visible
    const int lane = threadIdx.x;
    const int idx = 4 * ((lane >> 2) & 3) + (lane >> 4); // V layout: the vector row this lane holds
    int mo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) mo[r] = sm_tab[(TAB4_MTT + r) * 64 + lane];
    const int pho = idx <= 12 ? zi_of(idx) : 0;  // q~ rows [u; x] -> z index
    const int pwo = 4 + (idx & 3);               // p_w rows -> z index of w
    const int pdo = idx <= 12 ? REC_PD + idx : REC_ZERO;
    init_stage_constants(lane);
    double pv = 0.0;
    double e0, e1, e2, f0, f1, f2; // two prefetch sets: (e*) staged by the odd steps, (f*) by the even ones
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 MA, MB = zero;
    double phiA, hcA, phiwA, tpA, pdA, phiB = 0.0, hcB = 0.0, phiwB = 0.0, tpB, pdB;
    { // prologue: operands of stage N-1 into set A, prefetch of stage N-2
        cgdouble *rp = w.rec + (size_t)(N - 1) * REC_STRIDE;
        e0 = rp[lane]; e1 = rp[64 + lane]; e2 = rp[128 + (lane < 14 ? lane : 0)];
        tpA = rp[REC_T + lane];
        pdA = rp[pdo];
        WSYNC();
        sm[S_M + lane] = e0; sm[S_E + 64 + lane] = e1;
        if (lane < 14) sm[S_E + 128 + lane] = e2;
        WSYNC();
#pragma unroll
        for (int r = 0; r < 4; r++) MA[r] = m_row<NP>(N - 1)[mo[r]];
        phiA = sm[S_E + REC_PHIB + pho] + smu * sm[S_E + REC_PHIC + pho];
        phiA = (idx <= 12) ? phiA : 0.0;
        phiwA = sm[S_E + REC_PHIB + pwo] + smu * sm[S_E + REC_PHIC + pwo];
        hcA = sm[S_E + REC_HC];
        cgdouble *r2 = w.rec + (size_t)(N > 1 ? N - 2 : 0) * REC_STRIDE, *r3 = w.rec + (size_t)(N > 2 ? N - 3 : 0) * REC_STRIDE;
        e0 = r2[lane]; e1 = r2[64 + lane]; e2 = r2[128 + (lane < 14 ? lane : 0)];
        f0 = r3[lane]; f1 = r3[64 + lane]; f2 = r3[128 + (lane < 14 ? lane : 0)];
        tpB = r2[REC_T + lane]; pdB = r2[pdo];
    }
    int kk = N - 1;
    for (; kk >= 1; kk -= 2) {
        backvec_step<NP>(w, kk, kk == N - 1, lane, idx, smu, mo, pho, pwo, pdo, MA, phiA, hcA, phiwA, tpA, pdA, MB, phiB, hcB, phiwB, e0, e1, e2, pv);
        backvec_step<NP>(w, kk - 1, false, lane, idx, smu, mo, pho, pwo, pdo, MB, phiB, hcB, phiwB, tpB, pdB, MA, phiA, hcA, phiwA, f0, f1, f2, pv);
    }
    if (kk == 0) backvec_step<NP>(w, 0, N == 1, lane, idx, smu, mo, pho, pwo, pdo, MA, phiA, hcA, phiwA, tpA, pdA, MB, phiB, hcB, phiwB, e0, e1, e2, pv);
    // p_w[g] sits in the quad-0 lanes of row g; the stage-0 solve wants it in the lanes (g, 13)
    stage0_solve<NP>(w, xinit, lane, __shfl(pv, lane & 48));
    FULLSYNC();
}

// ------------------------------------------------------------------ forward sweep: dz for all stages
// du = -T' [hc dw; dx; 1],  ds+ = Mt [du; dx; 1]: two chained mat-vec products per stage on the 4x4x4 MFMA (matvec4),
// the vectors stay in V layout.
// WITH_Y (corrector pass): also the multipliers of the Newton system  y+_k = P_k ds_k + p_k  (P_k gathered from its
// packed lower triangle straight into the operand registers, prefetched one stage ahead), written to ynew.
// Software pipeline, branch-free body: the record of stage k+1 is staged through LDS and its operands are gathered
// while the chained MFMAs of stage k execute; the global prefetch runs two stages ahead.  Two register sets (A/B)
// alternate, so no operand is ever copied.
template <int NP, bool WITH_Y>
__device__ __forceinline__ void forward_step(const WsView &w, int N, int kk, int lane, int idx, const int (&tto)[4],
                                             const int (&mto)[4], const int (&pmo)[4], int pvo, const d4 &ctt, const d4 &cmt, double chc,
                                             const d4 &cP, double cpv, d4 &ntt, d4 &nmt, double &nhc, d4 &nP, double &npv,
                                             double &e0, double &tp, double &pr0, double &pr1, double &v)
{
    const bool q0 = idx < 4 && (lane & 12) == 0; // rows 0..3 (quad 0 of every 16-lane row)
    const double v1 = q0 ? chc * v : (idx == 13 ? 1.0 : v); // row 13 multiplies the kbar column
    const double D1 = matvec4(ctt, v1, 0.0);
    // stage the (already fetched) record of the next stage through LDS ...
    WSYNC();
    sm[S_M + lane] = e0; sm[S_T + lane] = tp;
    if (WITH_Y) { sm[S_PRAW + lane] = pr0; if (lane < 48) sm[S_PRAW + 64 + lane] = pr1; }
    WSYNC();
    { // ... prefetch the one after it (clamped: the tail re-reads the last record, unused) ...
        const int kf = (kk + 3 < N) ? kk + 3 : N - 1; // staged again two steps from now
        cgdouble *rp = w.rec + (size_t)kf * REC_STRIDE;
        e0 = rp[lane]; tp = rp[REC_T + lane];
        if (WITH_Y) { pr0 = rp[REC_P + lane]; pr1 = rp[REC_P + 64 + (lane < 48 ? lane : 0)]; } // packed P (96) and p (16)
    }
#pragma unroll
    for (int s = 0; s < 4; s++) { ntt[s] = sm[tto[s]]; nmt[s] = m_row<NP>(kk + 1 < N ? kk + 1 : N - 1)[mto[s]]; }
    nhc = sm[S_T + 14];
    if (WITH_Y) { // P and p of the next stage, gathered from the staged (coalesced) copy of its packed block
#pragma unroll
        for (int r = 0; r < 4; r++) nP[r] = sm[pmo[r]];
        npv = sm[pvo];
    }
    double Y = 0.0;
    if (WITH_Y) Y = matvec4(cP, v, cpv); // y+_k = P_k ds_k + p_k
    const double du = -D1;
    const double v2 = q0 ? du : (idx == 13 ? 1.0 : v); // row 13 multiplies the d column
    const double D2 = matvec4(cmt, v2, 0.0);
    { // branch-free LDS writes: the four replicas of a row (lane & 3) write the same value to the same slot, du goes to
      // the padding row 19 from the lanes that do not hold it; dz rows 17..19 / ynew rows 13..15 are padding
        double *dzl = dz_area<NP>();
        dzl[(q0 ? idx : DZ_ROWS - 1) * NP + kk] = du;
        dzl[(4 + idx) * NP + kk] = v; // (row 19 = padding: whichever write lands there is never read)
        if (WITH_Y) { // y+ of the Newton system: kept in LDS (the model phase's transposition buffer is idle here)
            if (NP <= FRP_RB_MAX_NP) rb_area<NP>()[idx * NP + kk] = Y;
            else if ((lane & 3) == 0) w.step[idx * NP + kk] = Y;
        }
    }
    v = D2; // rows 13..15 of Mt are zero
}

template <int NP, bool WITH_Y>
__device__ __noinline__ void sweep_forward(WsView w, int N)
{
    w = uni(w); N = uni(N);
    FULLSYNC(); // phase boundary: T' / kbar of the backward sweep are visible
    const int lane = threadIdx.x;
    const int idx = 4 * ((lane >> 2) & 3) + (lane >> 4); // V layout: the vector row this lane holds
    int mto[4], tto[4], pmo[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        mto[s] = sm_tab[(TAB4_MT + s) * 64 + lane];
        tto[s] = sm_tab[(TAB4_TT + s) * 64 + lane];
        const int po = sm_tab[(TAB4_P + s) * 64 + lane]; // record slot of P[row][col] (or REC_ZERO) -> slot of its staged copy
        pmo[s] = po == REC_ZERO ? S_ZERO : S_PRAW + (po - REC_P);
    }
    const int pvo = idx <= 12 ? S_PRAW + (REC_PV - REC_P) + idx : S_ZERO;
    init_stage_constants(lane);
    double v = sm[S_DS0 + idx]; // ds_0 (entries 13..15 are zero)
    double e0, tp, pr0 = 0.0, pr1 = 0.0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 PA = zero, PB = zero;
    double pvA = 0.0, pvB = 0.0;
    {
        cgdouble *rp = w.rec;
        e0 = rp[lane]; tp = rp[REC_T + lane];
        if (WITH_Y) { pr0 = rp[REC_P + lane]; pr1 = rp[REC_P + 64 + (lane < 48 ? lane : 0)]; }
    }
    WSYNC();
    sm[S_M + lane] = e0; sm[S_T + lane] = tp;
    if (WITH_Y) { sm[S_PRAW + lane] = pr0; if (lane < 48) sm[S_PRAW + 64 + lane] = pr1; }
    WSYNC();
    d4 ttA, mtA, ttB, mtB;
    double hcA, hcB = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) { ttA[s] = sm[tto[s]]; mtA[s] = m_row<NP>(0)[mto[s]]; }
    hcA = sm[S_T + 14];
    if (WITH_Y) {
#pragma unroll
        for (int r = 0; r < 4; r++) PA[r] = sm[pmo[r]];
        pvA = sm[pvo];
    }
    double f0, fp, fr0 = 0.0, fr1 = 0.0; // second prefetch set (staged by the odd steps)
    {
        cgdouble *rp = w.rec + (size_t)(N > 1 ? 1 : 0) * REC_STRIDE, *rq = w.rec + (size_t)(N > 2 ? 2 : N - 1) * REC_STRIDE;
        e0 = rp[lane]; tp = rp[REC_T + lane];
        f0 = rq[lane]; fp = rq[REC_T + lane];
        if (WITH_Y) {
            pr0 = rp[REC_P + lane]; pr1 = rp[REC_P + 64 + (lane < 48 ? lane : 0)];
            fr0 = rq[REC_P + lane]; fr1 = rq[REC_P + 64 + (lane < 48 ? lane : 0)];
        }
    }
    int kk = 0;
    for (; kk + 1 < N; kk += 2) {
        forward_step<NP, WITH_Y>(w, N, kk, lane, idx, tto, mto, pmo, pvo, ttA, mtA, hcA, PA, pvA, ttB, mtB, hcB, PB, pvB, e0, tp, pr0, pr1, v);
        forward_step<NP, WITH_Y>(w, N, kk + 1, lane, idx, tto, mto, pmo, pvo, ttB, mtB, hcB, PB, pvB, ttA, mtA, hcA, PA, pvA, f0, fp, fr0, fr1, v);
    }
    if (kk < N) forward_step<NP, WITH_Y>(w, N, kk, lane, idx, tto, mto, pmo, pvo, ttA, mtA, hcA, PA, pvA, ttB, mtB, hcB, PB, pvB, e0, tp, pr0, pr1, v);
    WSYNC();
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
                                            gdouble *__restrict__ prec, cgdouble *__restrict__ ppre, int N, int MF, int nfk,
                                            int model, double &m_p, double &m_d, double &s_sdl, double &s_lds, double &s_dsdl)
{
    constexpr int H = 64 / NP;
    constexpr int R = (NZ + H - 1) / H;
    const int lane = threadIdx.x;
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N && half < H;
    double *stg = stage_area<NP>();

    // one constraint of the affine step (smu = 0, corr = 0): returns t1 = (l r_in - corr)/s, sinv = 1/s
    auto cstep = [&](int c, double gdz, double viol, double &t1, double &sinv) {
        const double s = ps[c * NP + k], l = pl[c * NP + k];
        const double u = fast_rcp(s * l);
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
        if (H > 1) {
            b0 = xsub_sum<NP>(b0); b1 = xsub_sum<NP>(b1); b2 = xsub_sum<NP>(b2);
            c0 = xsub_sum<NP>(c0); c1 = xsub_sum<NP>(c1); c2 = xsub_sum<NP>(c2);
        }
        if (kact && half == 0) {
            stg[0 * NP + k] = b0; stg[1 * NP + k] = b1; stg[2 * NP + k] = b2;
            stg[3 * NP + k] = c0; stg[4 * NP + k] = c1; stg[5 * NP + k] = c2;
        }
    }
    WSYNC();
    if (kact) {
        double pc[NPRE]; // ref(3), weights(3), yaw_ref from the transposed copy (coalesced; the parameter rows are 1 KB apart)
        pc[0] = ppre[0 * NP + k]; pc[1] = ppre[1 * NP + k]; pc[2] = ppre[2 * NP + k];
        pc[6] = ppre[6 * NP + k]; pc[7] = ppre[7 * NP + k]; pc[8] = ppre[8 * NP + k]; pc[9] = ppre[9 * NP + k];
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        gdouble *rec = prec + (size_t)k * REC_STRIDE;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int ib = r * H, i = ib + half;
            if (i >= NZ) continue;
            const double hd = ROW_PICK(cq.hd(i));
            const double qi = ROW_PICK(cq.q(i));
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
            const double zi = pz[i * NP + k], dzi = pdz[i * NP + k];
            double pb = hd * zi + qi; // cost gradient
            if (ib < 8) pb += (i < 8 ? cq.hc() : 0.0) * pz[(i < 4 ? i + 4 : (i < 8 ? i - 4 : i)) * NP + k];
            double tl, tu, sil, siu;
            cstep(i, -dzi, lb - zi, tl, sil);
            cstep(17 + i, dzi, zi - ub, tu, siu);
            pb += tu - tl;
            double pcf = siu - sil;
            if (ib + H > 8 && ib < 11) {
                if (i >= 8 && i < 11) { pb += stg[(i - 8) * NP + k]; pcf += stg[(3 + i - 8) * NP + k]; }
            }
            rec[REC_PHIB + i] = pb;
            rec[REC_PHIC + i] = pcf;
        }
    }
}

template <int NP>
__device__ __noinline__ SlackOut phase_affine(WsView w, int N, int MF, int nfk, int model,
                                              double mu, int mtot, double tol_comp)
{
    w = uni(w); N = uni(N); MF = uni(MF); model = uni(model);
    mu = uni(mu); mtot = uni(mtot); tol_comp = uni(tol_comp);
    PROF_BEGIN();
    FULLSYNC(); // phase boundary: dz of the forward sweep is visible
    PROF_SEG(0);
    double m_p = 0.0, m_d = 0.0, s_sdl = 0.0, s_lds = 0.0, s_dsdl = 0.0;
    affine_body<NP>(w.s, w.lam, w.corr, w.z, dz_area<NP>(), w.face, w.rec, w.pre, N, MF, nfk, model, m_p, m_d, s_sdl, s_lds, s_dsdl);
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
                                          gdouble *__restrict__ py, const double *__restrict__ pynew,
                                          int N, int MF, int nfk, double smu, double ftb, double gap, double inv_mtot_kappa, double &ap_out, double &ad_out)
{
    constexpr int H = 64 / NP;
    constexpr int R = (NZ + H - 1) / H;
    constexpr int MAXF = 8; // corridor rounds kept in registers; further rounds are recomputed
    const int lane = threadIdx.x;
    const int k = lane % NP, half = lane / NP;
    const bool kact = k < N && half < H;
    double m_p = 0.0, m_d = 0.0;
    double q1 = 0.0, q2 = 0.0, q3 = 0.0; // sums of ds l, s dl, ds dl: the average complementarity after the step
    double dsb[2 * R], dlb[2 * R], dsf[MAXF], dlf[MAXF];
    double z8 = 0, z9 = 0, z10 = 0, d8 = 0, d9 = 0, d10 = 0;
    auto cstep = [&](int c, double gdz, double viol, double &ds, double &dl) {
        const double s = ps[c * NP + k], l = pl[c * NP + k];
        const double u = fast_rcp(s * l);
        const double sinv = u * l, linv = u * s;
        ds = -(viol + s) - gdz;
        const double rc = s * l - smu + pcorr[c * NP + k];
        dl = (-rc - l * ds) * sinv;
        m_p = fmax(m_p, -ds * sinv);
        m_d = fmax(m_d, -dl * linv);
        q1 = fma(ds, l, q1); q2 = fma(s, dl, q2); q3 = fma(ds, dl, q3);
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
            const int ib = r * H, i = ib + half;
            dsb[2 * r] = dlb[2 * r] = dsb[2 * r + 1] = dlb[2 * r + 1] = 0.0;
            if (i >= NZ) continue;
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
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
    // multiplier safeguard: s_i lam_i >= mu_new / KAPPA_LAM for every pair after the step
    q1 = wave_sum(q1); q2 = wave_sum(q2); q3 = wave_sum(q3);
    const double fprod = (gap + ap * q1 + ad * (q2 + ap * q3)) * inv_mtot_kappa;
    auto commit = [&](int c, double ds, double dl) {
        const double sn = ps[c * NP + k] + ap * ds;
        double ln = pl[c * NP + k] + ad * dl;
        if (ln * sn < fprod) ln = fprod * fast_rcp(sn);
        ps[c * NP + k] = sn;
        pl[c * NP + k] = ln;
    };
    if (kact) {
        for (int j = half + MAXF * H; j < nfk; j += H) { // rare: more corridor rounds than kept in registers
            double a, b;
            face(j, a, b);
            commit(34 + j, a, b);
        }
#pragma unroll
        for (int t = 0; t < MAXF; t++) {
            const int j = half + t * H;
            if (j < nfk) commit(34 + j, dsf[t], dlf[t]);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = r * H + half;
            if (i >= NZ) continue;
            commit(i, dsb[2 * r], dlb[2 * r]);
            commit(17 + i, dsb[2 * r + 1], dlb[2 * r + 1]);
            pz[i * NP + k] += ap * pdz[i * NP + k];
            if (i < NS) py[i * NP + k] += ap * (pynew[i * NP + k] - py[i * NP + k]); // y <- y + ap (y+ - y)
        }
    }
    ap_out = ap; ad_out = ad;
}

template <int NP>
__device__ __noinline__ SlackOut phase_step(WsView w, int N, int MF, int nfk, double smu, double ftb, double gap, double inv_mtot_kappa)
{
    w = uni(w); N = uni(N); MF = uni(MF); smu = uni(smu); ftb = uni(ftb); gap = uni(gap); inv_mtot_kappa = uni(inv_mtot_kappa);
    PROF_BEGIN();
    FULLSYNC(); // phase boundary: dz of the forward sweep is visible
    PROF_SEG(6);
    double ap, ad;
    const double *ynew = NP <= FRP_RB_MAX_NP ? (const double *)rb_area<NP>() : (const double *)w.step;
    step_body<NP>(w.s, w.lam, w.corr, w.z, dz_area<NP>(), w.face, w.y, ynew, N, MF, nfk, smu, ftb, gap, inv_mtot_kappa, ap, ad);
    PROF_SEG(7);
    FULLSYNC();
    PROF_SEG(8);
    PROF_END(6);
    SlackOut o;
    o.ap = ap; o.ad = ad; o.sigma = 0.0; o.smu = smu;
    return o;
}

// ------------------------------------------------------------------ initialisation of one solve (lane == stage)
// Its own (non-inlined) function like every other phase: the stage-divergent loops below (face counts differ per
// stage) must not share a register allocation with the long-lived state of the solver loop -- a VGPR spill placed
// at the exit of such a loop executes with an empty EXEC mask and saves nothing.
struct InitOut {
    int nfk, mtot, bad;
};
template <int NP>
__device__ __noinline__ InitOut phase_init(WsView w, cgdouble *pk, const int *nfaces, const double *x0, int N, int M, int MF, double mu0)
{
    w = uni(w); N = uni(N); M = uni(M); MF = uni(MF); mu0 = uni(mu0);
    const int lane = threadIdx.x, k = lane;
    const bool act = lane < N;
    int nf = 0;
    int bad_param = 0;
    double smin = 1e300;
    if (act) {
        if (nfaces) nf = nfaces[k];
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
        const double *z0 = x0 + (size_t)k * NZ;
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
#pragma unroll
        for (int i = 0; i < NPRE; i++) w.pre[i * NP + k] = pk[i]; // transposed copy of the leading parameters for the per-iteration phases
        for (int i = 0; i < 3; i++) { // padding rows (tile rows 13..15) of dz and y
            dz_area<NP>()[(17 + i) * NP + k] = 0.0;
            w.y[(13 + i) * NP + k] = 0.0;
        }
    }
    // record slots that are read before (or without ever) being written: the linearisation of the last stage (0..63), the
    // zero slot, the dynamics Hessian (stays zero in Gauss-Newton mode / last stage) and the pad (128..191 covers them;
    // PHIC in 128..141 is rewritten every iteration) -- coalesced, all lanes
    for (int kk = 0; kk < N; kk++) {
        w.rec[(size_t)kk * REC_STRIDE + lane] = 0.0;
        w.rec[(size_t)kk * REC_STRIDE + 128 + lane] = 0.0;
    }
    smin = wave_min(smin);
    const int mtot = (int)wave_sum(act ? (double)(34 + nf) : 0.0);
    InitOut o;
    o.bad = wave_max((double)bad_param) > 0.0 ? 1 : 0;
    o.mtot = mtot;
    o.nfk = 0;
    if (o.bad) return o;
    {
        // infeasible-start initialisation: uniform slack shift (see oracle/nmpc_ipm.c)
        const double shift = (smin >= S_MIN) ? 0.0 : (S_MIN - smin) + fmax(0.0, -smin);
        if (act) {
            for (int i = 0; i < 34 + nf; i++) {
                const double s = w.s[i * NP + k] + shift;
                w.s[i * NP + k] = s;
                w.lam[i * NP + k] = mu0 / s;
            }
        }
    }
    o.nfk = __shfl(nf, lane % NP); // face count of stage k = lane % NP for the (half, stage) lane mapping
    return o;
}

// ------------------------------------------------------------------ the solver kernel
// One problem `b`, using workspace slot `slot` (slots are reused by successive problems of the same workgroup, so
// the HBM footprint of the solver state is (#resident waves) x (state size), independent of the batch size).
template <int NP>
__device__ __forceinline__ void solve_one(const KernelArgs &a, const int b, const int slot)
{
    const int lane = threadIdx.x;
    const int N = a.N, M = a.M, MF = a.MF, np = NPRE + 4 * M;
    const int mcf = 34 + MF;
    const bool act = lane < N; // lane == stage in the initialisation
    const int k = lane;

    WsView w;
    {
        gdouble *base = (gdouble *)(a.ws + (size_t)slot * ws_doubles_per_problem(N, MF));
        w.rec = base;
        w.z = w.rec + (size_t)N * REC_STRIDE;
        w.y = w.z + 17 * NP;
        w.pre = w.y + Y_ROWS * NP;   // [NPRE][NP] (the block keeps its historical size of DZ_ROWS rows)
        w.s = w.pre + DZ_ROWS * NP;
        w.lam = w.s + (size_t)mcf * NP;
        w.corr = w.lam + (size_t)mcf * NP;
        w.step = w.corr + (size_t)mcf * NP;       // ds | dlam of the corrector step ([2 mcf][NP])
        w.face = w.step + 2 * (size_t)mcf * NP;
    }
    cgdouble *xinit = (cgdouble *)(a.xinit + (size_t)b * 9);
    cgdouble *pbase = (cgdouble *)(a.params + (size_t)b * N * np);
    cgdouble *pk = pbase + (size_t)(act ? k : 0) * np;

    // ---------------------------------------------------------------- init (lane == stage)
    const InitOut ini = phase_init<NP>(w, pk, a.nfaces ? a.nfaces + (size_t)b * N : nullptr, a.x0 + (size_t)b * N * NZ, N, M, MF, a.mu0);
    if (ini.bad) { // a stage has more live corridor rows than the workspace was sized for (MF)
        if (lane == 0) { a.exitflag[b] = FRP_EXIT_PARAM_VALUE; a.iters[b] = 0; }
        if (act) {
            const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
            for (int i = 0; i < NZ; i++) a.z[((size_t)b * N + k) * NZ + i] = z0[i];
        }
        return;
    }
    const int nfk = ini.nfk, mtot = ini.mtot;
    const int hess = a.hessian ? 1 : 0;
    const int model = a.models ? a.models[b] : a.model; // normal / final objective of THIS problem (switch_to_final, nmpc_solver.cpp:381)
    FULLSYNC();

    int flag = FRP_EXIT_MAXIT, it = 0, nfallback = 0;
    double theta_h = hess ? 1.0 : 0.0; // weight of the dynamics Hessian
    double res_eq = 0, res_in = 0, rs = 0, rcomp = 0, pobj = 0, mu = 0, sigma = 0, step_cc = 0;

#ifdef FRP_PROFILE
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tc0, tc1;
    const long long t_start = wall_clock64(); // 100 MHz constant clock: start / end of this solve on the launch time line
#define TICK() tc0 = clock64()
#define TOCK(i) do { tc1 = clock64(); tph[i] += tc1 - tc0; tc0 = tc1; } while (0)
#else
#define TICK()
#define TOCK(i)
#endif
    __builtin_amdgcn_s_setprio(0);
    for (it = 0;; it++) {
        // Long solves set the duration of a launch (the batch is done when its slowest problem is): a wave that is
        // past the typical iteration count gets issue priority over the wave it shares the SIMD with.
        if (it == FRP_PRIO_IT1) __builtin_amdgcn_s_setprio(1);
        else if (it == FRP_PRIO_IT2) __builtin_amdgcn_s_setprio(2);
        else if (it == FRP_PRIO_IT3) __builtin_amdgcn_s_setprio(3);
        TICK();
        const ModelOut mo_ = phase_model<NP>(w, xinit, N, model, hess);
        const EvalOut e = phase_eval<NP>(w, N, MF, nfk, model, mo_.eq, mo_.obj);
        res_eq = wave_max(e.eq); res_in = wave_max(e.in); rs = wave_max(e.rs); rcomp = wave_max(e.rc);
        pobj = wave_sum(e.obj);
        mu = wave_sum(e.gap) / (double)mtot;
        if (!(res_eq == res_eq) || !(rs == rs) || !(pobj == pobj)) { flag = FRP_EXIT_BADFUNCEVAL; break; }
        if (res_eq <= a.tol_eq && res_in <= a.tol_ineq && rs <= a.tol_stat && rcomp <= a.tol_comp) { flag = FRP_EXIT_OPTIMAL; break; }
        if (it >= a.maxit) { flag = FRP_EXIT_MAXIT; break; }
        if (mu > DIVERGE_MU * fmax(1.0, a.mu0) || rs > DIVERGE_RS) { flag = FRP_EXIT_NOPROGRESS; break; }
        TOCK(0);

        // predictor (affine) solve with the Hessian  H_GN + theta_h H_dyn.  theta_h = 1 is the exact Hessian; when a
        // Riccati pivot block is indefinite the iteration is redone with the Gauss-Newton Hessian (theta 0), theta_h is
        // quartered and then recovers by 0.1 per successful iteration: alternating between the two Hessians at full
        // weight can cycle for 100+ iterations on a locally non-convex problem.
        int fr = sweep_factor<NP>(w, xinit, N, theta_h);
        if (fr && theta_h > 0.0) {
            nfallback++;
            theta_h *= THETA_DOWN;
            fr = sweep_factor<NP>(w, xinit, N, 0.0);
        } else if (hess) {
            theta_h = fmin(1.0, theta_h + THETA_UP);
        }
        if (fr) { flag = FRP_EXIT_FACTORIZATION; break; }
        TOCK(1);
        sweep_forward<NP, false>(w, N);
        TOCK(2);
        const SlackOut s0 = phase_affine<NP>(w, N, MF, nfk, model, mu, mtot, a.tol_comp);
        sigma = s0.sigma;
        TOCK(3);
        // corrector solve (same factorisation, new rhs)
        sweep_backvec<NP>(w, xinit, N, s0.smu);
        TOCK(4);
        sweep_forward<NP, true>(w, N);
        TOCK(5);
        const SlackOut s1 = phase_step<NP>(w, N, MF, nfk, s0.smu, a.ftb, mu * (double)mtot, 1.0 / (KAPPA_LAM * (double)mtot));
        step_cc = s1.ap;
        TOCK(3);
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
            for (int i = 0; i < 6; i++) o[i] = (double)tph[i]; // cycles: eval, factor, forward(affine), affine+step, backvec, forward(corrector, with y)
            o[6] = (double)t_start; o[7] = (double)wall_clock64();
#endif
        }
    }
    (void)sigma;
}

// Persistent workgroups: grid = min(B, resident slots); each wave pulls the next problem index from a device
// counter (zeroed by the launcher) until the batch is exhausted -- natural load balancing over very different
// iteration counts, and a bounded, cache-friendly workspace.
template <int NP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRP_WAVES_PER_EU, FRP_WAVES_PER_EU))) void nmpc_ipm_kernel(KernelArgs a)
{
    const int slot = blockIdx.x;
    init_lane_tables();
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = atomicAdd(a.counter, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= a.B) break;
        if (a.order) b = a.order[b]; // longest-expected-first launch order (see order_keys_kernel)
        solve_one<NP>(a, b, slot);
        FULLSYNC();
    }
}

// ------------------------------------------------------------------ launch order: longest expected solve first
// The launch ends when its slowest problem does, and a problem that needs 3-5x the typical iteration count should not
// be the last one to start.  Proxy for the work of a solve: the objective of the caller's initial guess (a plan that
// starts far from its reference / corridor costs more and takes more interior-point iterations; rank correlation with
// the iteration count ~0.45 on the BASELINE workloads, and the hardest problems are reliably in the upper half, i.e.
// in the first wave of resident workgroups).  Only the ORDER in which the persistent workgroups pull problems
// changes; every problem is solved exactly as before.
__global__ __launch_bounds__(256) void order_keys_kernel(int B, int N, int np, int model, const int *__restrict__ models, const double *__restrict__ x0,
                                                         const double *__restrict__ params, double *__restrict__ keys)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), k = threadIdx.x & 63; // one wavefront per problem, lane = stage
    if (b >= B) return;
    double c = 0.0;
    if (k < N) {
        const size_t t = (size_t)b * N + k;
        double zl[NZ], p10[NPRE];
#pragma unroll
        for (int i = 0; i < NZ; i++) zl[i] = x0[t * NZ + i];
#pragma unroll
        for (int i = 0; i < NPRE; i++) p10[i] = params[t * np + i];
        c = stage_cost(zl, p10, stage_class(k, N), models ? models[b] : model, nullptr);
    }
    c = wave_sum(c);
    if (k == 0) keys[b] = (c == c && c < 1e300) ? c : 0.0;
}
// order = the problems sorted by decreasing key, to bucket resolution (also zeroes the work-queue head): one workgroup, a 1024-bin counting sort on the
// (monotone) bit pattern of the non-negative keys -- i.e. on a log scale -- with the bins spread over the key range of
// this batch.  The order inside a bin is arbitrary (atomics); order[] is a permutation for any input.
__global__ __launch_bounds__(1024) void order_bucket_kernel(int B, const double *__restrict__ keys, int *__restrict__ order, int *__restrict__ counter, int *__restrict__ cu_slots)
{
    if (threadIdx.x == 0) *counter = 0; // queue head of the solve that follows on this stream
    for (int i = threadIdx.x; i < CU_SLOT_ENTRIES; i += 1024) cu_slots[i] = 0;
    __shared__ unsigned long long s_min, s_max;
    __shared__ int hist[1024];
    const int t = threadIdx.x;
    if (t == 0) { s_min = ~0ull; s_max = 0ull; }
    hist[t] = 0;
    __syncthreads();
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = t; i < B; i += 1024) {
        const double k = keys[i];
        const unsigned long long u = (unsigned long long)__double_as_longlong(k > 0.0 ? k : 0.0);
        lo = u < lo ? u : lo; hi = u > hi ? u : hi;
    }
    atomicMin(&s_min, lo); atomicMax(&s_max, hi);
    __syncthreads();
    const unsigned long long base = s_min, span = s_max - s_min;
    int shift = 0;
    while ((span >> shift) >= 1024ull) shift++;
    for (int i = t; i < B; i += 1024) {
        const double k = keys[i];
        const unsigned long long u = (unsigned long long)__double_as_longlong(k > 0.0 ? k : 0.0);
        atomicAdd(&hist[1023 - (int)((u - base) >> shift)], 1); // bin 0 = largest keys
    }
    __syncthreads();
    { // exclusive prefix sum over the 1024 bins: a shuffle scan inside every wavefront, then the sixteen wave totals
      // (two barriers instead of the twenty of a Hillis-Steele scan over the workgroup: the kernel is all latency)
        __shared__ int wtot[16];
        const int own = hist[t], lane = t & 63, wv = t >> 6;
        int v = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(v, off);
            if (lane >= off) v += up;
        }
        if (lane == 63) wtot[wv] = v;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) before += w < wv ? wtot[w] : 0;
        hist[t] = before + v - own;
    }
    __syncthreads();
    for (int i = t; i < B; i += 1024) {
        const double k = keys[i];
        const unsigned long long u = (unsigned long long)__double_as_longlong(k > 0.0 ? k : 0.0);
        order[atomicAdd(&hist[1023 - (int)((u - base) >> shift)], 1)] = i;
    }
}

// ------------------------------------------------------------------ batched model callback
// One thread per (problem, stage): the reference's extfunc for B*N stage points (casadi2forces.c:42-245).  HBM-bound:
// 147 doubles in, 282 out per point, 221 of them the dense 13 x 17 Jacobian (column-major, ld 13, as the reference's
// sparse2fullcopy writes it).  A thread that stored its own Jacobian would touch 64 cache lines per store instruction, so
// the Jacobian leaves through LDS: every thread parks the 51 entries of its compact linearisation, then the wavefront
// writes the points' Jacobians one after the other as runs of 64 consecutive doubles (source of each dense entry: a
// compile-time table into the parked record, whose slots 51 / 52 / 53 hold 0, 1 and dt).
constexpr int SE_SLOTS = 55; // 51 + constants (0, 1, dt), odd stride
constexpr int SE_MAXM = 64;  // corridor rows staged per point
__host__ __device__ constexpr int jc_source(int e) // dense entry e = col * 13 + row  ->  slot of the parked record
{
    const int col = e / 13, row = e % 13;
    if (row >= 9) return col == row - 9 ? 52 : 51;
    const int bi = row / 3, ii = row % 3;
    if (col < 4) { // lin_B(row, col)
        if (col == 3) return bi == 0 ? 36 + ii : (bi == 1 ? 39 + ii : 51);
        if (bi == 1) return 42 + ii * 3 + col;
        if (bi == 2) return ii == col ? 53 : 51;
        return 51;
    }
    if (col < 8) return 51;
    const int j = col - 8, bj = j / 3, jj = j % 3; // lin_A(row, j)
    if (bi == 0) return bj == 0 ? (ii == jj ? 52 : 51) : (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? 51 : (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? 52 : 51;
}
struct JcTable {
    unsigned char v[256];
};
constexpr JcTable make_jc_table()
{
    JcTable t{};
    for (int e = 0; e < 256; e++) t.v[e] = (unsigned char)(e < 221 ? jc_source(e) : 51);
    return t;
}
__device__ const JcTable g_jc_table = make_jc_table();

__global__ __launch_bounds__(64) void stage_eval_kernel(int B, int N, int M, int model, const double *__restrict__ z,
                                                          const double *__restrict__ params, double *__restrict__ f,
                                                          double *__restrict__ gf, double *__restrict__ c,
                                                          double *__restrict__ Jc, double *__restrict__ h)
{
    __shared__ double s_lin[1][64 * SE_SLOTS]; // one wavefront per workgroup: 31 KB of LDS each, five per CU
    __shared__ double s_pos[64 * 3]; // one point's corridor rows [A (3 M) | b (M)]
    const size_t total = (size_t)B * N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool act = t < total;
    const int k = act ? (int)(t % N) : 0;
    const int np = NPRE + 4 * M;
    const double *zk = z + (act ? t : 0) * NZ, *pk = params + (act ? t : 0) * np;
    double zl[NZ], p10[NPRE];
#pragma unroll
    for (int i = 0; i < NZ; i++) zl[i] = zk[i];
#pragma unroll
    for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
    const int sc = stage_class(k, N);
    if (act && (f || gf)) {
        double g[NZ];
        const double cost = stage_cost(zl, p10, sc, model, g);
        if (f) f[t] = cost;
        if (gf) {
#pragma unroll
            for (int i = 0; i < NZ; i++) gf[t * NZ + i] = g[i];
        }
    }
    double *mine = s_lin[wv] + lane * SE_SLOTS;
    s_pos[lane * 3] = zl[8]; s_pos[lane * 3 + 1] = zl[9]; s_pos[lane * 3 + 2] = zl[10];
    if (c || Jc) {
        if (act && sc != STAGE_LAST) {
            Lin L;
            double xn[9];
            rk2<true>(zl + 8, zl, p10 + 3, xn, &L);
            if (c) {
#pragma unroll
                for (int i = 0; i < 9; i++) c[t * 13 + i] = xn[i];
#pragma unroll
                for (int i = 0; i < 4; i++) c[t * 13 + 9 + i] = zl[i];
            }
            const double *Lc = reinterpret_cast<const double *>(&L);
#pragma unroll
            for (int i = 0; i < 51; i++) mine[i] = Lc[i];
            mine[51] = 0.0; mine[52] = 1.0; mine[53] = DT;
        } else {
            if (act && c) for (int i = 0; i < 13; i++) c[t * 13 + i] = 0.0;
            for (int i = 0; i < 54; i++) mine[i] = 0.0; // the last stage has no dynamics: a zero Jacobian
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const size_t t0 = t - lane; // first point of this wavefront
    const int npts = t0 < total ? (int)((total - t0) < 64 ? (total - t0) : 64) : 0;
    if (Jc) {
        const int src0 = g_jc_table.v[lane], src1 = g_jc_table.v[64 + lane], src2 = g_jc_table.v[128 + lane], src3 = g_jc_table.v[192 + lane];
        for (int p = 0; p < npts; p++) {
            const double *rec = s_lin[wv] + p * SE_SLOTS;
            double *J = Jc + (t0 + p) * 221;
            J[lane] = rec[src0];
            J[64 + lane] = rec[src1];
            J[128 + lane] = rec[src2];
            if (lane < 221 - 192) J[192 + lane] = rec[src3];
        }
    }
    if (h && M > 0) {
        // corridor rows h = A pos - b, eight points at a time: the wavefront reads their [A | b] (4 M consecutive doubles each)
        // with coalesced loads -- sixteen in flight per lane -- into the LDS area the Jacobian pass has just vacated, then
        // writes the 8 M results, consecutive in memory, with coalesced stores
        if (M <= SE_MAXM) {
            constexpr int PB = 8;
            double *ab = s_lin[0]; // [PB][4 SE_MAXM]
            static_assert(PB * 4 * SE_MAXM <= 64 * SE_SLOTS, "staging area");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int p0 = 0; p0 < npts; p0 += PB) {
                const int nb = npts - p0 < PB ? npts - p0 : PB;
                double r0[PB], r1[PB], r2[PB], r3[PB];
#pragma unroll
                for (int q = 0; q < PB; q++) {
                    const double *src = params + (t0 + p0 + (q < nb ? q : 0)) * np + NPRE;
                    r0[q] = lane < 4 * M ? src[lane] : 0.0;
                    r1[q] = lane + 64 < 4 * M ? src[lane + 64] : 0.0;
                    r2[q] = lane + 128 < 4 * M ? src[lane + 128] : 0.0;
                    r3[q] = lane + 192 < 4 * M ? src[lane + 192] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < PB; q++) {
                    double *dst = ab + q * 4 * SE_MAXM;
                    dst[lane] = r0[q]; dst[lane + 64] = r1[q]; dst[lane + 128] = r2[q]; dst[lane + 192] = r3[q];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double *out = h + (t0 + p0) * M;
                for (int o = lane; o < nb * M; o += 64) {
                    const int q = o / M, j = o - q * M;
                    const double *a = ab + q * 4 * SE_MAXM, *ps = s_pos + (p0 + q) * 3;
                    out[o] = a[3 * j] * ps[0] + a[3 * j + 1] * ps[1] + a[3 * j + 2] * ps[2] - a[3 * M + j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (act) {
            const double *A = pk + NPRE, *bb = pk + NPRE + 3 * M;
            for (int j = 0; j < M; j++) h[t * M + j] = A[3 * j] * zl[8] + A[3 * j + 1] * zl[9] + A[3 * j + 2] * zl[10] - bb[j];
        }
    }
}

// ------------------------------------------------------------------ launchers
// Resident single-wave workgroups.  Measured on MI355X (profiles/r01_slots_sweep.txt): 6 per CU (1.5 waves per SIMD)
// beats the 8 the register budget allows -- the working set of 2048 resident solves (~176 KB each) no longer fits the
// 256 MB Infinity Cache, and the Riccati sweeps are latency-bound.  For a batch of a few rounds the slots are evened
// out over the rounds (4096 problems -> 3 rounds of 1366 instead of 1536 + 1536 + 1024).
static int resident_slots(int B)
{
    static int cap = 0;
    if (cap == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cap = cus * FRP_SLOTS_PER_CU;
        if (const char *e = getenv("FRP_RESIDENT_SLOTS")) { // tuning knob
            const int v = atoi(e);
            if (v > 0) cap = v;
        }
        if (cap > FRP_MAX_SLOTS) cap = FRP_MAX_SLOTS;
        if (cap < 1) cap = 1;
    }
    if (B <= cap) return B;
    const int rounds = (B + cap - 1) / cap;
    return (B + rounds - 1) / rounds;
}

// workspace = [solver state of min(B, FRP_MAX_SLOTS) slots][work-queue counter, 256 B][per-CU arrival counters, 8 KB][keys: B doubles][order: B ints]
constexpr int QUEUE_RESERVED = 32 + CU_SLOT_ENTRIES / 2; // doubles
static size_t queue_offset_doubles(int B, int N, int MF)
{
    const size_t slots = B < FRP_MAX_SLOTS ? B : FRP_MAX_SLOTS;
    return slots * ws_doubles_per_problem(N, MF);
}
size_t ws_bytes(int B, int N, int MF)
{
    return (queue_offset_doubles(B, N, MF) + QUEUE_RESERVED + (size_t)B + ((size_t)B + 1) / 2) * sizeof(double);
}

__global__ void reset_counter_kernel(int *counter, int *cu_slots)
{
    if (threadIdx.x == 0) *counter = 0;
    for (int i = threadIdx.x; i < CU_SLOT_ENTRIES; i += blockDim.x) cu_slots[i] = 0;
}

// Kernel selection: the LDS-resident four-wave kernel (frp_ipm_lds.hip) is the product path; FRP_KERNEL=r01 selects the
// round-1 single-wave kernel with its HBM workspace (kept for same-box A/B measurements only).
static bool use_r01_kernel()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("FRP_KERNEL");
        v = (e && (e[0] == 'r' || e[0] == 'h')) ? 1 : 0;
    }
    return v == 1;
}
static int lds_resident_slots(int B, int N)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int cap = cus * lds_workgroups_per_cu(N);
    if (const char *e = getenv("FRP_RESIDENT_SLOTS")) { // tuning knob
        const int v = atoi(e);
        if (v > 0) cap = v;
    }
    // every resident workgroup is used: the queue is pulled dynamically, so evening the slots over the "rounds" of the batch
    // (what the single-wave kernel does) only lowers the residency -- measured 1.58 vs 1.71 ms at B = 4096 (768 vs 683)
    return B <= cap ? B : cap;
}

hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream)
{
    KernelArgs k = a;
    const bool lds = !use_r01_kernel() && lds_kernel_supports(a.N, a.MF);
    const int slots = lds ? lds_resident_slots(a.B, a.N) : resident_slots(a.B);
    double *q = a.ws + queue_offset_doubles(a.B, a.N, a.MF);
    k.counter = reinterpret_cast<int *>(q);
    k.cu_slots = reinterpret_cast<int *>(q + 32);
    k.order = nullptr;
    if (a.B > slots) { // more problems than resident workgroups: order the queue, longest expected solve first
        double *keys = q + QUEUE_RESERVED;
        int *order = reinterpret_cast<int *>(keys + a.B);
        k.order = order;
        hipLaunchKernelGGL(order_keys_kernel, dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, stream, a.B, a.N, NPRE + 4 * a.M, a.model,
                           a.models, a.x0, a.params, keys);
        hipLaunchKernelGGL(order_bucket_kernel, dim3(1), dim3(1024), 0, stream, a.B, keys, order, k.counter, k.cu_slots);
    } else {
        // a one-thread kernel rather than hipMemsetAsync: as a node of a captured hipGraph the 4-byte memset was not
        // ordered before the solve on the graph's first launch (ROCm 7.2; tools/graph_tick.py), which left the queue
        // exhausted and the previous outputs in place
        hipLaunchKernelGGL(reset_counter_kernel, dim3(1), dim3(256), 0, stream, k.counter, k.cu_slots);
    }
    if (lds) return launch_ipm_lds(k, slots, stream);
    switch (padded_stages(a.N)) {
    case 16: hipLaunchKernelGGL(nmpc_ipm_kernel<16>, dim3(slots), dim3(64), 0, stream, k); break;
    case 20: hipLaunchKernelGGL(nmpc_ipm_kernel<20>, dim3(slots), dim3(64), 0, stream, k); break;
    case 32: hipLaunchKernelGGL(nmpc_ipm_kernel<32>, dim3(slots), dim3(64), 0, stream, k); break;
    default: hipLaunchKernelGGL(nmpc_ipm_kernel<64>, dim3(slots), dim3(64), 0, stream, k); break;
    }
    return hipGetLastError();
}

hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream)
{
    const size_t total = (size_t)B * N;
    const int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL(stage_eval_kernel, dim3(blocks), dim3(64), 0, stream, B, N, M, model, z, params, f, gf, c, Jc, h);
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
