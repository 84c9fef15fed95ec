// frp_ipm_lds.hip -- gfx950 interior-point NMPC solver, LDS-resident, four cooperating wavefronts per problem.
//
// Replaces the reference's closed NLP solver FORCESNLPsolver_{normal,final}_solve (FORCESNLPsolver_normal.h:323,
// called at forces_normal.cpp:139) and its model callback (FORCESNLPsolver_normal_casadi2forces.c:42-245).
// Same iteration as the CPU oracle (oracle/nmpc_ipm.c): Mehrotra predictor-corrector primal-dual interior point,
// multiplier safeguard, exact dynamics Hessian with Gauss-Newton fallback, Riccati recursion over the stage chain.
//
// One workgroup = 4 wavefronts = one NMPC problem at a time; persistent workgroups pull problems from a device queue.
// Nothing of the per-iteration state lives in HBM:
//   * the per-stage records the Riccati sweeps work on (linearisation, barrier Hessian, T', packed P, rhs vectors,
//     Newton step) stay in LDS for the whole solve: RS = 309 doubles per stage, 49 KB at N = 20 -> 3 problems per CU;
//   * slacks, multipliers, second-order terms, corridor faces, the iterate z and the equality multipliers y live in the
//     REGISTERS of the wave that owns them.
// Wave roles (wave-uniform control flow, 5 workgroup barriers per interior-point iteration):
//   wave 0  "Riccati": factorisation sweep (16x16x4 FP64 MFMA register tiles), forward / backward vector sweeps
//           (4x4x4 FP64 MFMA mat-vec chains); gathers its operands straight from the LDS records;
//   wave 1  "model":   lane = stage; owns z and y; Heun step, compact Jacobian, exact Hessian, equality residuals;
//   wave 2  "bounds":  lane = (row group, stage); owns the 34 bound slacks / multipliers of every stage;
//   wave 3  "faces":   lane = (face group, stage); owns the corridor rows (A pos - b <= 1e-5) of every stage.
// Norms, step lengths and the centring parameter are combined from per-wave partial results in LDS by every wave
// redundantly (bitwise identical), so all four waves take the same branches without a second barrier.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <cstdlib>
#include <atomic>
#include "frp_model.hpp"
#include "../../include/frp_nmpc.h"
#include "frp_kernels.h"
#include "frp_device.hpp"

// (the Q4 translation unit instantiates the same templates on another record layout: a namespace of its own)
#if defined(FRP_LDS_Q4_TU)
#define FRP_LR lrq
#elif defined(FRP_LDS_Q30_TU)
#define FRP_LR lrs
#elif defined(FRP_LDS_S2_TU)
#define FRP_LR lr2
#else
#define FRP_LR lr
#endif

namespace frp {
namespace FRP_LR {

// ------------------------------------------------------------------ per-stage LDS record (doubles)
// Two layouts.  The "plain" one (RS = 309: three problems per CU) keeps the packed P_k of every stage in LDS.  The "Q4" one
// (-DFRP_LDS_Q4_TU, frp_ipm_lds_q4.hip: RS = 241 -> 38.6 KB per problem, FOUR problems per CU on three-wave workgroups) moves P_k
// -- written once by the factorisation sweep, read once by the multiplier recovery y = P ds + p -- to an L2-resident workspace in
// global memory (KernelArgs::pws, 92 doubles per stage and resident workgroup) and lets T' and p share the slots of the
// barrier-augmented Hessian, which is dead once the factorisation step of its stage has gathered it.
// (three independent parts, switchable one by one for bisection builds: FRP_QP = P in global memory, FRP_QL = the 241-double
// record, FRP_QW = three-wave workgroups; the Q4 translation unit defines all three)
#ifdef FRP_LDS_Q4_TU
#define FRP_QP 1
#define FRP_QL 1
#define FRP_QW 1
#endif
// The "Q30" translation unit (frp_ipm_lds_q30.hip, round 6): N <= 30 at THREE problems per CU.  P in global memory and the overlay record as in Q4, on four-wave
// workgroups with the lane == stage model phase, and a record of 215 doubles (FRP_QS): no trig hand-over slots, P d aliased onto the p slots, no hole between the
// B and C twins, the external force read from the parameters instead of the workgroup scratch: 30 x 215 x 8 + 1.4 KB = 53 032 B per workgroup.
#ifdef FRP_LDS_Q30_TU
#define FRP_QP 1
#define FRP_QL 1
#define FRP_QS 1
#define FRP_NO_YPARK 1
#endif
#ifdef FRP_QS
constexpr bool QS = true;
#else
constexpr bool QS = false;
#endif
#ifdef FRP_QP
constexpr bool QP = true;
#else
constexpr bool QP = false;
#endif
#ifdef FRP_QL
constexpr bool QL = true;
#else
constexpr bool QL = false;
#endif
#ifdef FRP_QW
constexpr bool QW = true;
#else
constexpr bool QW = false;
#endif
static_assert(!QL || QP, "the short record has no room for P");
constexpr int R_LIN = 0;     // compact linearisation (51): Apv Ape Avv Ave BpT BvT Bvw
constexpr int R_D = 51;      // d = prev(z_k) - s_{k+1}, s-order [w; x] (13)            } the "M row": 64 slots
#ifndef FRP_QL
constexpr int R_T = 64;      // T' = [R | Kbar_x | kbar | hc]: lane (g, c) <-> T'[g][c] (64)
// overlay region (104): the barrier-augmented Hessian is dead once the factorisation step of the stage has gathered
// it, and that step's output P_k takes its place
constexpr int R_PHID = 128;  // diag of Phi = cost Hessian + bound barriers (17)
constexpr int R_PHIPOS = 145; // corridor barrier block on pos (3 x 3)
constexpr int R_PHI = 154;   // predictor rhs gradient phi_aff without its corridor part (17)
constexpr int R_HD = 171;    // exact Hessian of y'c(z): 45 structurally non-zero entries (hd_pack)
constexpr int R_P = 128;     // P_k, packed lower triangle (91)
constexpr int R_PV = 219;    // p_k of the corrector solve (13)
constexpr int R_PD = 232;    // P_{k+1} d_k (13): written by the factorisation sweep, read by the corrector's backward sweep
constexpr int R_PHIB = 245;  // corrector rhs phi_cc = PHIB + (sigma mu) PHIC (17 + 17), bound rows and cost;
constexpr int R_PHIW = R_HD + 40; // (QP bisection builds on this layout) Phi_w of the stage (4), left by the factorisation step in consumed Hessian slots
#else
// overlay region (88): [HD 45 | PHID 17 | PHIPOS 9 | PHI 17] until the factorisation step of the stage has gathered it, then
// [T' 64 | p 13] (T' from that step, p from the corrector's backward sweep).  Scratch in it (all inside HD, whose only writer in the
// evaluation phase is the wave that uses the scratch): the Riccati wave's trig hand-over (12) and the Hessian's inputs the model
// wave leaves in the step phase (RT_HZ 10, RT_HY 6 -- T' slots, dead after the last forward sweep).
constexpr int R_HD = 64;     // exact Hessian of y'c(z): 45 structurally non-zero entries (hd_pack)
constexpr int R_PHID = 109;  // diag of Phi = cost Hessian + bound barriers (17)
constexpr int R_PHIPOS = 126; // corridor barrier block on pos (3 x 3)
constexpr int R_PHI = 135;   // predictor rhs gradient phi_aff without its corridor part (17)
constexpr int R_T = 64;      // T' (64), over HD / PHID / the head of PHIPOS
constexpr int R_PV = 128;    // p_k of the corrector solve (13), over PHIPOS / PHI
constexpr int R_P = 0;       // (P_k lives in global memory: packed index 0..90 of the stage's block, 91 = dump)
#ifndef FRP_QS
constexpr int R_PD = 152;    // P_{k+1} d_k (13): written by the factorisation sweep, read by the corrector's backward sweep;
                             //   from there to the next factorisation: the model wave's parked y (RT_Y)
constexpr int R_PHIB = 165;  // corrector rhs phi_cc = PHIB + (sigma mu) PHIC (17 + 17), bound rows and cost;
#else
// QS: P_{k+1} d_k lives in the p slots of stage k.  The factorisation step of stage k+1 stores it into stage k's record AFTER it has gathered stage k's
// tiles (the slots lie in the overlay over PHIPOS / PHI: the timing argument of T'), the vector backward sweep reads it one step before its own step at
// stage k overwrites the slot with p_k, and nothing reads p before that sweep.
constexpr int R_PD = R_PV;
constexpr int R_PHIB = 152;
#endif
constexpr int R_PHIW = R_PV + 13; // Phi_w of the stage (4) behind p, left by the factorisation step (the y+ rows of w are formed from T' and this)
static_assert(QS || R_PHIW + 4 <= R_PD, "Phi_w stash");
#endif
constexpr int R_CB = R_PHIB + 17; //   corridor part (pos entries) of PHIB; evaluation phase: A' lam (stationarity residual)
#ifndef FRP_QS
constexpr int R_BC = 21;     // distance from every "B" slot (PHIB, CB) to its "C" twin (PHIC, CC): one ds_read2 fetches both
#else
constexpr int R_BC = 20;     // (no hole between CB and PHIC: the spare is the twisted solve's)
#endif
constexpr int R_PHIC = R_PHIB + R_BC; // evaluation phase: PHIB = cost gradient + bound multipliers, PHIC = M'y part (stationarity residual)
constexpr int R_CC = R_CB + R_BC;     // corridor part of PHIC; evaluation phase: corridor part of phi_aff
#ifndef FRP_QS
constexpr int R_HC = R_PHIB + 41;    // (u_i, w_i) cost coupling -2 w_rate of this stage
constexpr int R_ZERO = R_HC + 1, R_ONE = R_HC + 2, R_DT = R_HC + 3; // constants the gathers pick up
constexpr int R_DUMP = R_HC + 4;  // target of masked-out writes (never read)
constexpr int R_DZ = R_HC + 5;    // Newton step [du(4); ds(13)]
#else
// [0 | Newton step 17 | hc | 1 | 0' | dt | dump]: the two zeros R_BC apart around the step
constexpr int R_ZERO = R_CC + 3, R_DZ = R_ZERO + 1, R_HC = R_DZ + 17, R_ONE = R_HC + 1, R_DT = R_ONE + 2, R_DUMP = R_DT + 1;
#endif
constexpr int R_ZERO2 = R_ZERO + R_BC; // second zero, R_BC behind the first: masked (B, C) pair reads
// Twisted solve: a stage of the FIRST half keeps the inverted transition [u; x]_k = T~ [w+; x+] + t~ where a stage of
// the second half keeps the linearisation: A~ = A^-1 has A's block pattern (A~pv, A~pe, A~vv, A~ve in the slots of Apv, Ape, Avv, Ave),
// B~ = -A^-1 B is dense in its p and v rows (12 + 12 entries), t~ takes d's place, and the record's dt slot holds -dt (the e rows of
// B~).  The nine entries of B~ that do not fit the 51 linearisation slots live where a first-half record has room: the columns 14, 15
// of T' (eight slots that only ever meet the zero rows 14, 15 of a sweep vector; the first-half sweeps do not store there), the second
// zero (only the second half's vector sweep reads it) -- and one spare: the hole between CB and PHIC.
#ifndef FRP_QL
constexpr int RS = 309;      // odd stride: lane == stage accesses are conflict-free
static_assert(R_HD + REC_HD_SIZE <= R_PV && R_PV + 13 <= R_PD, "overlay region");
static_assert(R_PHIB == 245 && R_HC == 286 && R_DZ == 291 && R_ZERO2 == 308, "the plain record");
#elif !defined(FRP_QS)
constexpr int RQ_MTRIG = R_ZERO2 + 1; // 12 slots of the model + corridor wave's own: four parking slots per lane of the stage (FRP_Q4_PARK; without it: that wave's trig hand-over)
constexpr int RS = RQ_MTRIG + 12;     // 241, odd
static_assert(RS == 241 && (RS & 1), "the Q4 record");
static_assert(R_HD + REC_HD_SIZE == R_PHID && R_PHID + 17 == R_PHIPOS && R_PHIPOS + 9 == R_PHI && R_PHI + 17 == R_PD, "overlay region");
static_assert(R_T + 64 <= R_PV && R_PV + 13 <= R_PD && R_PD + 13 == R_PHIB, "overlay region after the factorisation");
#else
constexpr int RQ_MTRIG = R_DUMP;      // (the lane == stage model phase has no trig hand-over)
constexpr int RS = R_DUMP + 1;        // 215, odd
static_assert(RS == 215 && (RS & 1) && R_ZERO2 == R_ONE + 1 && R_ZERO2 == R_DT - 1, "the Q30 record");
static_assert(R_HD + REC_HD_SIZE == R_PHID && R_PHID + 17 == R_PHIPOS && R_PHIPOS + 9 == R_PHI && R_PHI + 17 == R_PHIB, "overlay region");
static_assert(R_T + 64 <= R_PV && R_PHIW + 4 <= R_PHIB, "overlay region after the factorisation");
#endif
#ifndef FRP_QS
static_assert(R_PHIC + 17 <= R_CC && R_CC + 3 <= R_HC && R_DZ + 17 <= R_ZERO2 && R_ZERO2 < RS, "record tail");
#else
static_assert(R_PHIC + 17 == R_CC && R_CC + 3 == R_ZERO && R_DZ + 17 == R_HC && R_DUMP < RS, "record tail");
#endif
#ifndef FRP_TW_RHO
#define FRP_TW_RHO 1e12
#endif
constexpr double TW_RHO = FRP_TW_RHO; // penalty that pins x_0 in the arrival-cost recursion (tools/study/twisted_riccati.py)
// QP: what the multiplier recovery y+ = P ds + p needs of P_k that is NOT in the record goes to global memory.  The w rows of P are
// [Phi_w - hc^2 R | -hc Kbar_x] and the (x, w) block is their transpose: both follow from T' (in the record) and the four Phi_w (stashed in
// the record by the factorisation step).  Only S_xx (9 x 9, symmetric) leaves: per resident workgroup NP blocks of PG = 48 doubles, three
// groups of 16 -- group a (the lane of a stage that owns x rows 3a .. 3a+2) holds the diagonal block S_aa (lower triangle, 6) and the
// block S_{a,a+1} (row-major, 9; a+1 cyclic): every unique entry exactly once, the same SHAPE for every lane (uniform code), one
// 128-byte line per lane.  Slot 15 of group 0 is the dump of the masked lanes.
constexpr int PG = 48, PG_DUMP = 15;
__host__ __device__ constexpr int pg_slot(int trow, int c) // tile element (trow, c) -> slot of the stage's block, or the dump
{
    const int i = trow - 4, j = c - 4;
    if (i < 0 || i > 8 || j < 0 || j > 8) return PG_DUMP;
    const int bi = i / 3, ii = i % 3, bj = j / 3, jj = j % 3;
    if (bi == bj) return ii >= jj ? bi * 16 + ii * (ii + 1) / 2 + jj : PG_DUMP;
    if (bj == (bi + 1) % 3) return bi * 16 + 6 + 3 * ii + jj;
    return PG_DUMP;
}

// workgroup-shared scratch (doubles)
constexpr int X_RW = 16;              // stage-0 solve: Pww^-1 (16)
constexpr int X_PWX = 32;             // stage-0 solve: Pwx (4 x 9)
constexpr int X_DS0 = 68;             // ds_0 = [dw_0; dx_0] (16)
constexpr int X_XINIT = 84;           // xinit (9)
constexpr int X_DX0 = 96;             // xinit - x_0 (9)
constexpr int X_C0 = 93, X_C1 = 94;   // constants 0, 1
constexpr int X_IKM = 95;             // 1 / (KAPPA_LAM mtot) of the problem, for the wave that has no register for it (PARK)
constexpr int X_RED = 106;            // per-wave partial results: [4][16] (evaluation 0..2, affine 3..7, step 8..12: no slot is reused inside an iteration)
constexpr int X_FEXT = 170;           // external-force acceleration of every stage: [3][NP]
constexpr int X_TOTAL = 170;          // (+ 3 NP)

// ------------------------------------------------------------------ per-lane gather tables (record offsets)
// 16x16 register tiles (factorisation sweep): lane (g, c), register r <-> element (4r+g, c).
// 4x4x4 mat-vec operands (vector sweeps): register m, lane l <-> A[4 qI + qj][4 ((qI + m) & 3) + qk].
__host__ __device__ constexpr int zi_of(int a) { return a < 4 ? a : a + 4; }          // tile index (u, x) -> z index
__host__ __device__ constexpr int hidx_of(int a) { return a < 4 ? a : (a >= 7 && a <= 12 ? a - 3 : -1); }
// Mt[row][col], the augmented transition matrix: rows s+ = [w+(0..3); x+(4..12)], cols [u(0..3); x(4..12); 13 = d]
__host__ __device__ constexpr int m_src(int row, int col)
{
    if (row > 12 || col > 13) return R_ZERO;
    if (col == 13) return R_D + row;
    if (row < 4) return (col == row) ? R_ONE : R_ZERO;
    const int i = row - 4, bi = i / 3, ii = i % 3;
    if (col < 4) { // B[i][col]
        if (col == 3) return bi == 0 ? R_LIN + 36 + ii : (bi == 1 ? R_LIN + 39 + ii : R_ZERO);
        if (bi == 1) return R_LIN + 42 + ii * 3 + col;
        if (bi == 2) return ii == col ? R_DT : R_ZERO;
        return R_ZERO;
    }
    const int j = col - 4, bj = j / 3, jj = j % 3;
    if (bi == 0) return bj == 0 ? (ii == jj ? R_ONE : R_ZERO) : R_LIN + (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? R_ZERO : R_LIN + (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? R_ONE : R_ZERO;
}
// the three sources summed into C~[row][col] = Phi~ (diag + corridor + theta Hessian), rhs in column 13
__host__ __device__ constexpr int c_src(int row, int col, int which)
{
    int o1 = R_ZERO, o2 = R_ZERO, o3 = R_ZERO;
    if (row <= 12) {
        if (col == 13) {
            o1 = R_PHI + zi_of(row);
            if (row >= 4 && row <= 6) o2 = R_CC + (row - 4);
        } else if (col <= 12) {
            if (col == row) o1 = R_PHID + zi_of(row);
            if (row >= 4 && row <= 6 && col >= 4 && col <= 6) o2 = R_PHIPOS + (row - 4) * 3 + (col - 4);
            const int hr = hidx_of(row), hc_ = hidx_of(col);
            if (hr >= 0 && hc_ >= 0 && hd_pack(hr, hc_) >= 0) o3 = R_HD + hd_pack(hr, hc_);
        }
    }
    return which == 0 ? o1 : (which == 1 ? o2 : o3);
}
// T~[row][col] of a first-half stage: rows [u(0..3); x(4..12)] of stage k, cols [w+(0..3); x+(4..12); 13 = t~]
__host__ __device__ constexpr int lbv_slot(int e) // entry e = 4 i + col of the v rows of B~
{
    if (e < 3) return R_LIN + 48 + e;
    if (e == 11) return R_ZERO2;
    const int q = e - 3; // 0..7 -> T'[q / 2][14 + q % 2]
    return R_T + 16 * (q / 2) + 14 + (q % 2);
}
__host__ __device__ constexpr int ma_src(int row, int col)
{
    if (row > 12 || col > 13) return R_ZERO;
    if (col == 13) return R_D + row;
    if (row < 4) return (col == row) ? R_ONE : R_ZERO;
    const int i = row - 4, bi = i / 3, ii = i % 3;
    if (col < 4) { // B~[i][col]
        if (bi == 0) return R_LIN + 36 + ii * 4 + col;
        if (bi == 1) return lbv_slot(ii * 4 + col);
        return (col < 3 && ii == col) ? R_DT : R_ZERO; // (a first-half record's dt slot holds -dt)
    }
    const int j = col - 4, bj = j / 3, jj = j % 3;
    if (bi == 0) return bj == 0 ? (ii == jj ? R_ONE : R_ZERO) : R_LIN + (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? R_ZERO : R_LIN + (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? R_ONE : R_ZERO;
}
// twisted solve: workgroup scratch of its own for the arrival cost at the meeting stage -- (Q_m packed lower triangle (91), q_m (13)),
// a dump slot, the factor of the meeting system (L strictly lower, packed (78), 1 / D (13)), a zero, the model wave's copy of ds_m (16)
constexpr int TW_Q = 0, TW_QV = 91, TW_DUMP = 104, TW_L = 105, TW_DINV = 183, TW_ZERO = 196, TW_DS = 198, TW_TOTAL = 214;
enum { T_M = 0, T_C1 = 4, T_C2 = 8, T_C3 = 12, T_PP = 16, T_PD = 20, T4_MT = 24, T4_MTT = 28, T4_P = 32, T4_MU = 36, T4_MTTU = 37, T4_TS = 38,
       T_MA = 39, T4_MA = 43, T4_MAT = 47, T_SQ = 51, T_ROWS = 55 };
struct LaneTables {
    unsigned short v[T_ROWS][64];
};
constexpr LaneTables make_tables()
{
    LaneTables t{};
    for (int lane = 0; lane < 64; lane++) {
        const int g = lane >> 4, c = lane & 15;
        const int qk = lane >> 4, qI = (lane >> 2) & 3, qj = lane & 3;
        for (int r = 0; r < 4; r++) {
            const int trow = 4 * r + g;
            t.v[T_M + r][lane] = (unsigned short)m_src(trow, c);
            t.v[T_C1 + r][lane] = (unsigned short)c_src(trow, c, 0);
            t.v[T_C2 + r][lane] = (unsigned short)c_src(trow, c, 1);
            t.v[T_C3 + r][lane] = (unsigned short)c_src(trow, c, 2);
            // (QP: the slot inside the stage's block in global memory)
            t.v[T_PP + r][lane] = (unsigned short)(QP ? pg_slot(trow, c) : ((trow <= 12 && c <= trow) ? R_P + trow * (trow + 1) / 2 + c : R_DUMP));
            t.v[T_PD + r][lane] = (unsigned short)((c == 13 && trow <= 12) ? R_PD + trow : R_DUMP);
            const int row = 4 * qI + qj, col = 4 * ((qI + r) & 3) + qk;
            // forward sweep: the u columns go through T4_MU, row 13 carries the constant 1 from stage to stage
            t.v[T4_MT + r][lane] = (unsigned short)(col < 4 ? R_ZERO : (row == 13 && col == 13 ? R_ONE : m_src(row, col)));
            // backward vector sweep: q rows 0..3 come from T4_MTTU (the full product keeps phi_w there), rows 13..15 are unused
            t.v[T4_MTT + r][lane] = (unsigned short)((row < 4 || row > 12) ? R_ZERO : m_src(col, row));
            const int hi = row > col ? row : col, lo = row > col ? col : row;
            t.v[T4_P + r][lane] = (unsigned short)((row <= 12 && col <= 12) ? R_P + hi * (hi + 1) / 2 + lo : R_ZERO);
            // first half: the tile of T~, [u; x] = T~ v (row 13 carries the constant 1), and its transpose for the arrival vector sweep
            t.v[T_MA + r][lane] = (unsigned short)ma_src(trow, c);
            t.v[T4_MA + r][lane] = (unsigned short)((row == 13 && col == 13) ? R_ONE : ma_src(row, col));
            t.v[T4_MAT + r][lane] = (unsigned short)((row > 12 || col > 12) ? R_ZERO : ma_src(col, row));
            // store of the arrival-cost tile at the meeting stage into the workgroup scratch (lower triangle + column 13)
            t.v[T_SQ + r][lane] = (unsigned short)((trow <= 12 && c <= trow) ? TW_Q + trow * (trow + 1) / 2 + c
                                                   : ((c == 13 && trow <= 12) ? TW_QV + trow : TW_DUMP));
        }
        // "sliced" operands (one register): block qI contracts ITS quarter 4 qI .. 4 qI + 3 of the long dimension, the four
        // blocks are summed afterwards -- for products with only four output rows (du, q_u) and for Mt[:, u] du
        t.v[T4_MU][lane] = (unsigned short)m_src(4 * qI + qj, qk);                  // A_b[i][k] = Mt[4b+i][k], k = u column
        t.v[T4_MTTU][lane] = (unsigned short)m_src(4 * qI + qk, qj);                // A_b[i][k] = Mt[4b+k][i] = Mt'[i][4b+k]
        t.v[T4_TS][lane] = (unsigned short)(R_T + 16 * qj + 4 * qI + qk);           // A_b[i][k] = T'[i][4b+k]
    }
    return t;
}
static __device__ const LaneTables g_tab = make_tables();

__device__ __forceinline__ int tab(int row, int lane) { return (int)g_tab.v[row][lane]; }
#define BAR() __syncthreads()
#define FRP_SB() __builtin_amdgcn_sched_barrier(0)
// between the rounds / rows of the element-wise loops (unrolled): keeps the scheduler from interleaving them, i.e. from multiplying
// their temporaries in waves whose persistent state leaves ~30 registers (-DFRP_ROUND_SB)
#ifdef FRP_ROUND_SB
#define FRP_RSB() __builtin_amdgcn_sched_barrier(0)
#else
#define FRP_RSB()
#endif

// LDS pointers carry their address space: through a generic double* every access of a non-inlined function would pay
// 64-bit address arithmetic and the null check of the address-space cast
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(3))) const double cldouble;

// -DFRP_PROFILE: per-wave cycle counts of the work before each of the five barriers of an iteration and of the wait at it
#ifdef FRP_PROFILE
static __device__ long long g_prof_lds[4][16];
#define PROF_T0() const long long pinit0_ = clock64(); const long long pwall0_ = wall_clock64()
#define PROF_DECL() long long pw_[5] = {0, 0, 0, 0, 0}, pb_[5] = {0, 0, 0, 0, 0}, pt_ = clock64(); const long long pinit_ = pt_ - pinit0_
#define BAR_P(i)                                                    \
    do {                                                            \
        const long long t0_ = clock64();                            \
        pw_[i] += t0_ - pt_;                                        \
        __syncthreads();                                            \
        pt_ = clock64();                                            \
        pb_[i] += pt_ - t0_;                                        \
    } while (0)
#define PROF_FLUSH(wave, its)                                                                         \
    do {                                                                                              \
        if ((threadIdx.x & 63) == 0) {                                                                \
            for (int q_ = 0; q_ < 5; q_++) {                                                          \
                atomicAdd((unsigned long long *)&g_prof_lds[wave][q_], (unsigned long long)pw_[q_]);       \
                atomicAdd((unsigned long long *)&g_prof_lds[wave][5 + q_], (unsigned long long)pb_[q_]);   \
            }                                                                                         \
            atomicAdd((unsigned long long *)&g_prof_lds[wave][10], (unsigned long long)(its));          \
            atomicAdd((unsigned long long *)&g_prof_lds[wave][11], (unsigned long long)pinit_);         \
            atomicAdd((unsigned long long *)&g_prof_lds[wave][12], 1ull);                               \
        }                                                                                             \
    } while (0)
#else
#define PROF_T0()
#define PROF_DECL()
#define BAR_P(i) __syncthreads()
#define PROF_FLUSH(wave, its)
#endif
#ifdef FRP_PROFILE
static __device__ long long g_prof_seg[32];
#endif
#ifdef FRP_PROFILE_SEG // (each timer read drains lgkmcnt: the segments are serialised, use them for proportions only) // factor sweep: mfma X/G, gather, pivot, tail mfma, P update + stores; then whole sweeps: factor, forward, backvec, forward+y
#define SEG_DECL() long long sg_[6] = {0, 0, 0, 0, 0, 0}, st_ = clock64()
#define SEG(i) do { const long long tn_ = clock64(); sg_[i] += tn_ - st_; st_ = tn_; } while (0)
#define SEG_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 6; q_++) atomicAdd((unsigned long long *)&g_prof_seg[q_], (unsigned long long)sg_[q_]); } while (0)
#define SEGM_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int q_ = 0; q_ < 6; q_++) atomicAdd((unsigned long long *)&g_prof_seg[16 + q_], (unsigned long long)sg_[q_]); } while (0)
#else
#define SEG_DECL()
#define SEG(i)
#define SEG_FLUSH()
#define SEGM_FLUSH()
#endif
#ifdef FRP_PROFILE
#define SWEEP_T0() const long long sw0_ = clock64()
#define SWEEP_T1(i) do { if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&g_prof_seg[8 + (i)], (unsigned long long)(clock64() - sw0_)); } while (0)
// twisted solve: slot 22 + i gets the cycles since the last TW_T0() / TW_T1() of this wave
#define TW_T0() long long tw0_ = clock64()
#define TW_T1(i) do { const long long tn_ = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&g_prof_seg[(i)], (unsigned long long)(tn_ - tw0_)); tw0_ = tn_; } while (0)
#else
#define SWEEP_T0()
#define SWEEP_T1(i)
#define TW_T0()
#define TW_T1(i)
#endif

// ------------------------------------------------------------------ values every wave derives from the LDS partials
struct Norms {
    double eq, in, rs, rc, gap, obj;
};

// rows of the element-wise lane mapping: lane = sub * NP + k, sub < H = 64 / NP, row i = r * H + sub
template <int NP>
struct RowMap {
    static constexpr int H = 64 / NP;
    static constexpr int R = (NZ + H - 1) / H;
};

// value of a per-row constant for row i = ib + half, chosen among the H compile-time candidates of round r.
// `halfp` = a copy of `half` made opaque once per phase (opq below): the selects are loop invariant, and hoisted out of the
// interior-point loop they cost two VGPRs per constant for the whole solve in waves that are at the register cap -- where the
// allocator then spills them, an L2 round trip to save two v_cndmask
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }
#define ROW_PICK(expr_of_i)                                                                          \
    ([&]() {                                                                                           \
        double v_ = [&](int i) { return (double)(expr_of_i); }(ib < NZ ? ib : NZ - 1);                  \
        if (H > 1 && halfp == 1) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 1 < NZ ? ib + 1 : NZ - 1); \
        if (H > 2 && halfp == 2) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 2 < NZ ? ib + 2 : NZ - 1); \
        if (H > 3 && halfp == 3) v_ = [&](int i) { return (double)(expr_of_i); }(ib + 3 < NZ ? ib + 3 : NZ - 1); \
        return v_;                                                                                     \
    }())

template <int NP>
__device__ __forceinline__ double xsub_sum(double v) // sum over the H lanes that share a stage (valid in every lane of the group for H = 2, 4; in sub == 0 for H = 3)
{
    constexpr int H = 64 / NP;
    if (H == 2 && NP == 32) return v + __shfl_xor(v, 32);
    if (H == 2) { const int lane = threadIdx.x & 63; return v + __shfl(v, lane < NP ? (lane + NP < 64 ? lane + NP : lane) : lane - NP); } // (lanes k and NP + k; the lanes beyond 2 NP are inactive)
    if (H == 4) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
    if (H == 3) {
        const int lane = threadIdx.x & 63;
        const double a = __shfl(v, lane + NP < 64 ? lane + NP : lane), b = __shfl(v, lane + 2 * NP < 64 ? lane + 2 * NP : lane);
        return v + a + b;
    }
    return v;
}

// max |stationarity residual| from the three parts the evaluation phase left in the records (all waves, redundantly)
template <int NP>
__device__ __forceinline__ double stationarity_norm(cldouble *recs, int N)
{
    constexpr int H = RowMap<NP>::H, R = RowMap<NP>::R;
    const int lane = threadIdx.x & 63, k = lane % NP, half = lane / NP;
    double rs = 0.0;
    if (k < N && half < H) {
        cldouble *rec = recs + k * RS;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = r * H + half;
            if (i >= NZ) continue;
            double g = rec[R_PHIB + i] + rec[R_PHIC + i];
            if (i >= 8 && i < 11) g += rec[R_CB + i - 8];
            rs = fmax(rs, fabs(g));
        }
    }
    return wave_max(rs);
}

struct Ctl { // workgroup-shared control words
    int next, fail, bad, mtot, fail1, fail2; // twisted solve: fail1 = the first half's factorisation, fail2 = the system where the halves meet
    // FRP_EARLY_FACTOR: the termination decision of an iteration, taken by the helper waves while the Riccati wave is already factoring
    // (dec_tag = (problem << 8 | iteration) once dec_stop / dec_flag are valid; reset at kernel start: LDS keeps a previous launch's words)
    unsigned dec_tag;
    int dec_stop, dec_flag;
    int head_first; // (Q4) >= 0: this workgroup was first on its CU and took a head-start problem (KernelArgs::head_start): its first solve holds the CU's mark
};

typedef __attribute__((address_space(3))) Ctl lctl_t;
// ================================================================== wave 0: Riccati sweeps
// ---- stage-0 solve (both passes): dx_0 = xinit - x_0, dw_0 = -Pww^-1 (Pwx dx_0 + p_w); leaves ds_0 in X_DS0.
// pw_here: p_w[g] in the lanes (g, 13).
#ifndef FRP_S0_OPQ // 1: the lane opaque at the call in the factorisation sweep -- inlined into the kernel, the stage-0 solve's lane-dependent LDS addresses are loop
#define FRP_S0_OPQ 1 // invariants of the interior-point loop: hoisted, spilled, and fetched from scratch one round trip at a time at the tail of the factorisation
#endif
__device__ __forceinline__ void stage0_solve(ldouble *xs, int lane, double pw_here)
{
    const int g = lane >> 4, c = lane & 15;
    const bool xc = (c >= 4 && c <= 12);
    const double dxc = xc ? xs[X_DX0 + c - 4] : 0.0;
    WSYNC();
    const double prod = xc ? xs[X_PWX + g * 9 + c - 4] * dxc : (c == 13 ? pw_here : 0.0);
    const double rhs = row16_sum(prod);
    const double r0 = lane_bcast(rhs, 0), r1 = lane_bcast(rhs, 16), r2 = lane_bcast(rhs, 32), r3 = lane_bcast(rhs, 48);
    if (lane < 4) {
        xs[X_DS0 + lane] = -(xs[X_RW + lane * 4 + 0] * r0 + xs[X_RW + lane * 4 + 1] * r1 +
                             xs[X_RW + lane * 4 + 2] * r2 + xs[X_RW + lane * 4 + 3] * r3);
    } else if (lane <= 12) {
        xs[X_DS0 + lane] = dxc; // g == 0: index 4 + (c - 4) = c
    } else if (lane < 16) {
        xs[X_DS0 + lane] = 0.0;
    }
    WSYNC();
}

template <int N4> // lane c of a 16-lane row <- lane c + N4 (row_shl)
__device__ __forceinline__ double row_shl(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x100 + N4, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x100 + N4, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// y = A x + c on the 4x4x4 MFMA (see matvec4 in frp_device.hpp) with the four K-steps on two independent accumulators:
// a dependent MFMA issues every ~25 cycles and its result reaches the VALU ~23 cycles later, so two chains of two plus
// one add (~95 cycles) beat one chain of four (~125) on the serial path of the vector sweeps.
__device__ __forceinline__ double matvec4s(const d4 &A, double x, double c)
{
    const double x1 = quad_rot<1>(x), x2 = quad_rot<2>(x), x3 = quad_rot<3>(x);
    double d0 = mfma4(A[0], x, c);
    double d1 = mfma4(A[1], x1, 0.0);
    d0 = mfma4(A[2], x2, d0);
    d1 = mfma4(A[3], x3, d1);
    return d0 + d1;
}

// ---- factorisation sweep (predictor).  Backward Riccati recursion on 16x16 FP64 register tiles:
//   X = P M (col 13: P d + p+),  G = M'X + C~ (col 13: q~),  Guu = L D L',  K = L^-1 G_u,
//   T = L^-T D^-1 K (Kbar, kbar),  S = G - K' D^-1 K,  P <- [Phi_w - hc^2 R, -hc Kbar_x; -hc Kbar_x', S_xx].
// The tiles of stage k-1 are gathered from its LDS record as soon as the MFMAs that read the tiles of stage k have
// been issued: the gathers land while those MFMAs and the pivot-block factorisation execute.  The rank-4 products
// around the pivot block (K, R, T) run on the 4x4x4 MFMA (24 cycles instead of 64), whose block layout coincides
// with register 0 of the 16x16 tiles: A[i][k] in lane 16k+4b+i, B[k][j] in lane 16k+4b+j, D[i][j] in lane 16i+4b+j.
// Written for instruction count (round 3: 288 -> ~240 per stage):
//   * rows 4.. of P come out of ONE MFMA: S' = [0 | G_x.] - (D^-1 K)' [hc m | K_x.] has S_xx (and p_x) in the x columns and
//     -hc K_x' D^-1 m = -hc Kbar_x', the (x, w) block of P, in the u columns -- no transposed product, no row shifts, no selects;
//   * column 13 of the P tile simply carries p (its partner row of M is zero, so it does not enter X = P M); the affine part of
//     X is X += (column-13 mask) * P;  rows 13..15 of the tiles hold finite junk that meets zero rows of M;
//   * the w rows of P are one multiply-add on a gathered value: P_w = pq - hc (hc4 * [R | T]), pq = Phi_w on the diagonal lanes
//     and phi_w in column 13 (masks in the gather address), hc4 = hc in the u columns and 1 elsewhere;
//   * one select chain hands m = L^-1 to the lanes (entry (max, min) of the pair (g, c & 3)); its two uses mask it.
// Returns 1 when a pivot block is not positive definite.
__device__ __forceinline__ void gather_tiles(cldouble *rn, const int (&c1)[4], const int (&c2)[4], const int (&c3)[4],
                                             const int (&mo)[4], int pqo, double theta, d4 &C, d4 &Mt, double &hc, double &pq)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        // (the corridor block and the corridor part of the gradient sit in tile rows 4..6: register 1 only)
        C[r] = r == 1 ? rn[c1[r]] + rn[c2[r]] + theta * rn[c3[r]] : rn[c1[r]] + theta * rn[c3[r]];
        Mt[r] = rn[mo[r]];
    }
    hc = rn[R_HC];
    pq = rn[pqo];
}

// Q4: P_k leaves for the workgroup's block in global memory -- saddr form: uniform 64-bit base in scalar registers, a 32-bit byte
// offset per lane, the stage as the instruction's immediate.
template <int OFF>
__device__ __forceinline__ void gst(const double *base, unsigned boff, double v)
{
    static_assert(OFF >= 0 && OFF < 4096, "immediate offset of a global store");
#ifndef FRP_GST_CXX
    // Inline asm: written in C++ the compiler keeps a 64-bit address per store in vector registers (8 VGPRs in a wave at the cap).
    // CAUTION: the hazard recognizer does not look inside inline asm -- a memory instruction that reads the result of an MFMA needs
    // 18 wait states behind it, which the compiler inserts for its own stores (s_nop 15; s_nop 0) and NOT here: the caller makes sure
    // the MFMA has retired (first version: the rows 4..12 of P, which ARE an MFMA's result, left with stale registers -- the solves
    // still converged, two iterations late, on wrong multipliers).  Not on the compiler's vmcnt scoreboard either: nothing in the
    // sweeps reads global memory, and the predictor phase ends with an explicit s_waitcnt vmcnt(0).
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" : : "v"(boff), "v"(v), "s"(base), "n"(OFF) : "memory");
#else
    *(__attribute__((address_space(1))) double *)((__attribute__((address_space(1))) char *)base + boff + OFF) = v;
#endif
}
#ifdef FRP_QP
#define FRP_GP_PARAM , double *gp
#define FRP_GP_ARG(x) , x
#else
#define FRP_GP_PARAM
#define FRP_GP_ARG(x)
#endif
#ifdef FRP_INLINE_FACTOR
#define FRP_FACTOR_LINKAGE __forceinline__
#else
#define FRP_FACTOR_LINKAGE __noinline__
#endif
#ifdef FRP_INLINE_SWEEPS // (the vector sweeps too)
#define FRP_SWEEP_LINKAGE __forceinline__
#else
#define FRP_SWEEP_LINKAGE __noinline__
#endif
// twist (second half of a twisted solve: `recs` is the record of the meeting stage, N the stages from there to the end): the stage-0
// tail is replaced by publishing p of the meeting stage (column 13 of the tile) in its R_PV slots.
// POLL (FRP_EARLY_FACTOR): the sweep starts before the iteration's termination test is known; between its passes it looks for the helper
// waves' decision (Ctl::dec_tag == want) and leaves at once when that says "stop".  Return value: bit 0 = a pivot block failed, bit 1 =
// stopped by the decision, bit 2 = the decision has been seen (and said "go on").
template <bool twist, bool POLL = false>
__device__ FRP_FACTOR_LINKAGE int sweep_factor(ldouble *recs, ldouble *xs, int N, double theta FRP_GP_PARAM, lctl_t *ctl = nullptr, unsigned want = 0)
{
    N = uni(N); theta = uni(theta);
    want = (unsigned)uni((int)want); // (the problem index comes out of an LDS load: uniform, but not to the compiler -- a loop exit on it would make every value of the loop "divergent")
    bool decided = false;
    auto poll = [&]() -> bool { // true: stop
        if constexpr (POLL) {
            // (an LDS-address-space pointer: through a generic one the load is a FLAT instruction, whose wait drains the S_xx stores too -- measured +6 %)
            if (!decided && uni((int)*(volatile __attribute__((address_space(3))) unsigned *)&ctl->dec_tag) == (int)want) {
                decided = true;
                return uni(*(volatile __attribute__((address_space(3))) int *)&ctl->dec_stop) != 0;
            }
        }
        return false;
    };
#ifdef FRP_QP
    const double *gpk = uni(gp) + (size_t)(N - 1) * PG; // block of the stage the pointers sit on
#endif
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, c3_ = c & 3;
    int mo[4], c1[4], c2[4], c3[4], ppo[4], pdo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = tab(T_M + r, lane); c1[r] = tab(T_C1 + r, lane); c2[r] = tab(T_C2 + r, lane);
        c3[r] = tab(T_C3 + r, lane); ppo[r] = tab(T_PP + r, lane); pdo[r] = tab(T_PD + r, lane);
    }
    // Phi_w[g] on the diagonal lanes of the w block, phi_w[g] in column 13, zero elsewhere
    const int pqo = (c < 4 && g == c) ? R_PHID + 4 + g : (c == 13 ? R_PHI + 4 + g : R_ZERO);
    // the entry of m = L^-1 of the pivot block this lane needs, as m[g][c & 3] (c & 3 <= g) and / or m[c & 3][g] (g <= c & 3):
    // index into the strictly lower triangle (1,0) (2,0) (2,1) (3,0) (3,1) (3,2), 6 = unit diagonal
    const int mhi = g > c3_ ? g : c3_, mlo = g > c3_ ? c3_ : g;
    const int msel = mhi == mlo ? 6 : mhi * (mhi - 1) / 2 + mlo;
    const bool m_lower = c3_ <= g, m_upper = g <= c3_;
    const double m4 = c < 4 ? 1.0 : 0.0, m4c = 1.0 - m4, m13 = c == 13 ? 1.0 : 0.0; // lane masks as factors
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 P = zero, C, Mt;
    double hcn, pqn;
    gather_tiles(recs + (N - 1) * RS, c1, c2, c3, mo, pqo, theta, C, Mt, hcn, pqn);
    d4 G = C; // the last stage has no successor: G = C~
    bool ok = true;
    SEG_DECL();
    // Every LDS operand of a stage is (per-lane pointer) + (compile-time offset): the pointers sit on the record of the lowest
    // stage a step touches and move once per step -- per PASS OF FOUR STAGES while at least five are left (offsets 4 RS .. 0),
    // then per stage (offsets RS, 0) -- instead of ~25 address computations per stage from the table indices.
    ldouble *base = recs + (N - 1) * RS;
    ldouble *p_c1[4], *p_c3[4], *p_mo[4], *p_pp[4], *p_pd[4];
    ldouble *p_c2 = base + c2[1], *p_pq = base + pqo, *p_t = base + lane;
    ldouble *p_pw = base + ((c < 4 && g == c) ? R_PHIW + g : R_DUMP); // QP: Phi_w[g] stays in the record (the diagonal lanes of the w block hold it)
    unsigned gb[4]; // Q4: byte offsets of this lane's four tile entries inside a stage's block of packed P
#pragma unroll
    for (int r = 0; r < 4; r++) {
        p_c1[r] = base + c1[r]; p_c3[r] = base + c3[r]; p_mo[r] = base + mo[r]; p_pd[r] = base + pdo[r];
        if constexpr (QP) { gb[r] = 8u * (unsigned)ppo[r]; p_pp[r] = nullptr; }
        else { p_pp[r] = base + ppo[r]; gb[r] = 0; }
    }
    if constexpr (QP) asm volatile("" : "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3]));
    // (laundered through an empty asm: left to itself the optimiser turns every pointer back into base + 8 * index and
    // recomputes it at each use -- 28 address instructions per stage, which is what this is here to remove)
    auto opaque = [](ldouble *&q) { unsigned v = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)q; asm volatile("" : "+v"(v)); q = (ldouble *)(unsigned long long)v; };
    auto move = [&](int stages) {
        const int by = stages * RS;
        base -= by; p_c2 -= by; p_pq -= by; p_t -= by;
        opaque(base); opaque(p_c2); opaque(p_pq); opaque(p_t);
        if constexpr (QP) { p_pw -= by; opaque(p_pw); }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            p_c1[r] -= by; p_c3[r] -= by; p_mo[r] -= by; p_pd[r] -= by;
            opaque(p_c1[r]); opaque(p_c3[r]); opaque(p_mo[r]); opaque(p_pd[r]);
            if constexpr (!QP) { p_pp[r] -= by; opaque(p_pp[r]); }
        }
#ifdef FRP_QP
        gpk -= stages * PG;
#endif
    };
    // P_k (packed lower triangle) for the multiplier recovery y_k = P_k ds_k + p_k: to the Hessian part of this stage's record, which was
    // gathered one step ago (plain) / to the workgroup's block in global memory (Q4)
    auto store_p = [&](auto orc, const d4 &Pt) {
        constexpr int OR = decltype(orc)::value;
#ifdef FRP_QP
#pragma unroll
        for (int r = 0; r < 4; r++) gst<(OR / RS) * PG * 8>(gpk, gb[r], Pt[r]);
#else
#pragma unroll
        for (int r = 0; r < 4; r++) p_pp[r][OR] = Pt[r];
#endif
    };
    // one stage: OR / OG = offsets of its record and of the record below it from the pointers; LAST = stage 0 (nothing below)
    auto stage = [&](auto orc, auto lastc) {
        constexpr int OR = decltype(orc)::value, OG = OR - RS;
        constexpr bool LAST = decltype(lastc)::value;
        const double hc = hcn, pq = pqn; // of this stage
        if constexpr (!LAST) {
            // the tiles of this stage are consumed: gather those of the stage below while the pivot block is factored
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // (the corridor block and the corridor part of the gradient sit in tile rows 4..6: register 1 only)
                C[r] = r == 1 ? p_c1[r][OG] + p_c2[OG] + theta * p_c3[r][OG] : p_c1[r][OG] + theta * p_c3[r][OG];
                Mt[r] = p_mo[r][OG];
            }
            hcn = base[OG + R_HC];
            pqn = p_pq[OG];
        }
        // ---- pivot block Guu = L D L' (4 x 4): lower triangle to uniform registers, factored redundantly
        double q[16], Mi[6], Di[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(G[0], 16 * i + j);
        ok &= ldl4(q, Mi, Di); // (a failed pivot poisons the rest of the sweep, which is discarded: no branch in the loop)
        // the uniform factors reach the lanes through selects on lane-constant predicates (an LDS round trip costs ~250 cycles
        // of the dependency chain)
        double me = msel == 6 ? 1.0 : Mi[0];
#pragma unroll
        for (int i = 1; i < 6; i++) me = msel == i ? Mi[i] : me;
        // elimination in factored form: an explicit inverse of Guu cancels O(1e10) barrier terms against cond(Guu) eps errors
        const double m_gc = m_lower ? me : 0.0, m_cg = m_upper ? me : 0.0; // m[g][c & 3], m[c & 3][g]
        double dg = Di[3]; // (a chain, not a nested conditional: the nested form compiles to EXEC-masked regions)
        dg = g == 2 ? Di[2] : dg; dg = g == 1 ? Di[1] : dg; dg = g == 0 ? Di[0] : dg;
        const double md = dg * m_gc;
        const double K0 = mfma4(m_cg, G[0], 0.0);  // K = m G_u          (4 x 16, register-0 layout)
        const double Kd = dg * K0;
        // [R | Kbar_x | kbar] = m' D^-1 [m | K_x | k] in one 4x4x4 product: column block 0 of the B operand carries D^-1 m
        // instead of D^-1 K_u (which is not needed: K_u = D L')
        const double tsel = mfma4(m_gc, c < 4 ? md : Kd, 0.0); // T' of the sweeps; its columns 14, 15 are zero
        // (the u columns of rows 4.. would come out as G_xu - K_x' D^-1 K_u, zero only up to rounding RELATIVE TO G_xu, which carries
        //  barrier terms: they are taken out of both operands instead -- three multiplies -- and the block is exactly -hc K_x' D^-1 m)
        const double hcm = hc * m4;
        const double Kb = __builtin_fma(hcm, m_gc, K0 * m4c); // [hc m | K_x | k]
        d4 S = G;
        S[1] *= m4c; S[2] *= m4c; S[3] *= m4c;
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(-Kd, Kb, S, 0, 0, 0); // [-hc Kbar_x' | S_xx | p_x] in rows 4..12
        p_t[OR + R_T] = tsel;
        if constexpr (QP) p_pw[OR] = pq;
        const double hc4 = __builtin_fma(hc, m4, m4c);
        P[0] = __builtin_fma(-hc, hc4 * tsel, pq); // [Phi_w - hc^2 R | -hc Kbar_x | phi_w - hc kbar]
        P[1] = S[1]; P[2] = S[2]; P[3] = S[3];
        if constexpr (LAST) {
            if constexpr (QP) { FRP_SB(); asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); } // (S, an MFMA result, goes out through inline asm: see gst)
            store_p(orc, P);
        } else {
            // ---- X = P_k M_{k-1} (col 13: P d)
            d4 X = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], Mt[0], zero, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], Mt[1], X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(P[2], Mt[2], X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(P[3], Mt[3], X, 0, 0, 0);
            // (QP: behind the four dependent MFMAs of X -- pinned there -- the MFMA that produced S has long retired: see gst)
            if constexpr (QP) FRP_SB();
            store_p(orc, P);
            if constexpr (QP) FRP_SB();
            // ---- G of the stage below = M'X + C~ (col 13: q~)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                p_pd[r][OG] = X[r];                       // P d (column 13; every other lane writes the dump slot)
                X[r] = __builtin_fma(m13, P[r], X[r]);    // + p in column 13
            }
            // the rows w+ of M (tile rows 0..3) are [I 0 | d_w], so their slice adds X[w+_j][.] to G[u_j][.] (and something to
            // the unused row 13): one vector add instead of an MFMA
            C[0] += X[0];
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[1], X[1], C, 0, 0, 0);
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[2], X[2], G, 0, 0, 0);
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[3], X[3], G, 0, 0, 0);
        }
    };
    using std::integral_constant;
    int kk = N - 1;
    for (; kk >= 4; kk -= 4) {
        if (poll()) return 2;
        move(4); // pointers on stage kk - 4
        stage(integral_constant<int, 4 * RS>{}, std::false_type{});
        stage(integral_constant<int, 3 * RS>{}, std::false_type{});
        stage(integral_constant<int, 2 * RS>{}, std::false_type{});
        stage(integral_constant<int, 1 * RS>{}, std::false_type{});
    }
    if (kk == 3) { // (N = 4 m: the reference's 20) the last four stages as one pass on stage 0's record
        move(3);
        stage(integral_constant<int, 3 * RS>{}, std::false_type{});
        stage(integral_constant<int, 2 * RS>{}, std::false_type{});
        stage(integral_constant<int, 1 * RS>{}, std::false_type{});
    } else {
        for (; kk >= 1; kk--) {
            move(1);
            stage(integral_constant<int, RS>{}, std::false_type{});
        }
    }
    if (poll()) return 2;
    stage(integral_constant<int, 0>{}, std::true_type{}); // stage 0
    SEG_FLUSH();
    int fail = ok ? 0 : 1;
    if constexpr (twist) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (c == 13 && 4 * r + g <= 12) recs[R_PV + 4 * r + g] = P[r];
    } else if (!fail) {
        // stage 0: keep Pww^-1 and Pwx for the corrector pass, then solve for ds_0
        double q[16], Rw[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(P[0], 16 * i + j);
        if (!spd4_inverse(q, Rw)) fail = 1;
        else {
            WSYNC();
#pragma unroll
            for (int i = 0; i < 16; i++) xs[X_RW + i] = Rw[i];
            if (c >= 4 && c <= 12) xs[X_PWX + g * 9 + c - 4] = P[0];
            stage0_solve(xs, FRP_S0_OPQ ? opq(lane) : lane, P[0]); // (p_w sits in column 13 of the w rows)
        }
    }
    WSYNC();
    return fail | (decided ? 4 : 0);
}

// ---- the two vector sweeps.  One wavefront issues in order and its MFMAs do not overlap its own VALU / LDS instructions
// (tools/ubench/issue_rate.hip, fwd_model.hip: a 4x4x4 FP64 MFMA costs 16 issue cycles, every other instruction ~4.5, and
// independent work only ADDS its issue cycles), so a sweep costs the SUM of its instructions plus the dependency stalls
// that are left: both sweeps are written for instruction count.
//   * masks are encoded in the gather addresses (a lane that must see 0 / 1 / hc reads the record's constant slots), the
//     constant 1 of the affine column rides along as row 13 of the sweep vector: no selects on the serial path;
//   * products with four output rows (du, q_u) are ONE "sliced" MFMA (block b contracts quarter b of the long dimension)
//     followed by a sum over the quads of every 16-lane row, which also leaves the result where the next MFMA wants it;
//   * the LDS traffic is issued through inline asm: every operand has its own per-lane byte address that advances once
//     per pass of four stages, the stage offset is the instruction's immediate (the compiler re-derives every address
//     from the tables each stage and, with branches around, waits for ALL outstanding LDS operations), and the waits
//     count exactly the operations younger than the operand set a step is about to consume.
// Vectors are in V layout: lane 16 a + 4 b + j holds row 4 b + a.  Two operand sets alternate, each refilled for the
// stage two steps on, so a gather has a whole step to land.
template <int OFF>
__device__ __forceinline__ double lds_ld(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_st(unsigned addr, double v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(ldouble *p) { return (unsigned)(unsigned long long)p; }
constexpr int RSB = RS * 8; // record stride in bytes

// vector-only backward sweep (corrector): new rhs phi_cc = PHIB + smu PHIC (+ the corridor parts on the pos rows):
//   q~ = phi~ + M'(P d + p+),  E = T'' q_u,  p_x = q~_x - E_x,  p_w = phi_w - hc E_w;  kbar = E_w replaces column 13 of T'.
// The addresses sit on the LOWEST stage a pass touches (DS offsets are unsigned): a pass of two steps at stages kk, kk-1
// refills for kk-2, kk-3 with the addresses at kk-3.
struct BackAddr {
    unsigned m0, m1, m2, m3, mu, ph, cb, pu, hf, tp, pd, wk, wp;
    __device__ __forceinline__ void step(int n)
    {
        m0 += n; m1 += n; m2 += n; m3 += n; mu += n; ph += n; cb += n; pu += n; hf += n; tp += n; pd += n; wk += n; wp += n;
        // opaque to the optimiser: otherwise every address is re-derived from the tables and the induction variable at
        // every use (two VALU additions per LDS access instead of thirteen per pass)
        asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(mu), "+v"(ph), "+v"(cb), "+v"(pu), "+v"(hf), "+v"(tp), "+v"(pd), "+v"(wk), "+v"(wp));
    }
};
struct BackOps {
    double m0, m1, m2, m3, mu, phb, phc, cbb, cbc, pub, puc, hf, tp, pd;
};
template <int K>
__device__ __forceinline__ void back_gather(const BackAddr &p, BackOps &x)
{
    x.pd = lds_ld<K * RSB>(p.pd);
    x.mu = lds_ld<K * RSB>(p.mu);
    x.pub = lds_ld<K * RSB>(p.pu);               // quad 0: phi_u parts, other quads: 0
    x.puc = lds_ld<K * RSB + R_BC * 8>(p.pu);
    x.m0 = lds_ld<K * RSB>(p.m0);
    x.m1 = lds_ld<K * RSB>(p.m1);
    x.m2 = lds_ld<K * RSB>(p.m2);
    x.m3 = lds_ld<K * RSB>(p.m3);
    x.phb = lds_ld<K * RSB>(p.ph);               // rows 0..3 of the full q carry phi_w (their M' rows are masked out), rows 4..12 phi_x
    x.phc = lds_ld<K * RSB + R_BC * 8>(p.ph);
    x.cbb = lds_ld<K * RSB>(p.cb);               // corridor parts: pos rows, 0 elsewhere
    x.cbc = lds_ld<K * RSB + R_BC * 8>(p.cb);
    x.tp = lds_ld<K * RSB>(p.tp);                // read before the stage's own step rewrites its kbar column
    x.hf = lds_ld<K * RSB>(p.hf);                // quad 0: hc, other quads: 1
}
template <int YOUNGER>
__device__ __forceinline__ void back_wait(BackOps &x)
{
    asm volatile("s_waitcnt lgkmcnt(%14)"
                 : "+v"(x.m0), "+v"(x.m1), "+v"(x.m2), "+v"(x.m3), "+v"(x.mu), "+v"(x.phb), "+v"(x.phc), "+v"(x.cbb), "+v"(x.cbc), "+v"(x.pub),
                   "+v"(x.puc), "+v"(x.hf), "+v"(x.tp), "+v"(x.pd)
                 : "n"(YOUNGER));
}
template <int K, bool FIRST>
__device__ __forceinline__ void back_step(const BackAddr &p, const BackOps &x, double smu, double &pv)
{
    // serial path: p+ -> q_u -> E -> p;  the full q (needed only for p_x = q_x - E) runs beside it
    double d = __builtin_fma(smu, x.puc, x.pub);
    double q = __builtin_fma(smu, x.phc + x.cbc, x.phb + x.cbb);
    if (!FIRST) { // (the last stage of the horizon has no successor)
        const double tv = x.pd + pv;
        FRP_SB();
        d = mfma4(x.mu, tv, d);
        FRP_SB();
        const double t1 = quad_rot<1>(tv), t2 = quad_rot<2>(tv), t3 = quad_rot<3>(tv);
        q = mfma4(x.m0, tv, q);
        FRP_SB();
        const double r = d + quad_rot<1>(d);
        FRP_SB();
        q = mfma4(x.m1, t1, q);
        FRP_SB();
        d = r + quad_rot<2>(r);
        FRP_SB();
        q = mfma4(x.m2, t2, q);
        q = mfma4(x.m3, t3, q);
        FRP_SB();
    } else {
        const double r = d + quad_rot<1>(d);
        d = r + quad_rot<2>(r);
    }
    const double E = mfma4(x.tp, d, 0.0); // d = q_u[a] in every lane of row a;  E[c] = sum_k T'[k][c] q_u[k], V layout
    const double pn = __builtin_fma(-x.hf, E, q);
    FRP_SB();
    // (pn first: the hazard recognizer does not see inside inline asm, and an LDS store that reads an MFMA result needs wait
    // states after the MFMA; behind the store of pn -- a VALU result computed FROM E -- the MFMA has long retired)
    lds_st<K * RSB>(p.wp, pn); // p_k for y_k = P_k ds_k + p_k
    lds_st<K * RSB>(p.wk, E);  // kbar (rows 0..3; the other lanes write the dump slot)
    pv = pn;
}

template <int K> // stages K .. 0 above the addresses; X holds the operands of stage K
__device__ __forceinline__ void back_tail(const BackAddr &p, BackOps &X, BackOps &Y, double smu, double &pv)
{
    if constexpr (K > 0) {
        back_gather<K - 1>(p, Y); FRP_SB(); back_step<K, false>(p, X, smu, pv); back_wait<2>(Y);
        back_tail<K - 1>(p, Y, X, smu, pv);
    } else {
        back_step<0, false>(p, X, smu, pv); // stage 0
    }
}

template <bool twist>
__device__ FRP_SWEEP_LINKAGE void sweep_backvec(ldouble *recs, ldouble *xs, int N, double smu)
{
    N = uni(N); smu = uni(smu);
    const int lane = threadIdx.x & 63, a = lane >> 4, b = (lane >> 2) & 3;
    const int idx = 4 * b + a; // V layout: the vector row this lane holds
    ldouble *base = recs + (N - 2) * RS; // (N >= 2)
    BackAddr p;
    p.m0 = lds_addr(base + tab(T4_MTT + 0, lane)); p.m1 = lds_addr(base + tab(T4_MTT + 1, lane));
    p.m2 = lds_addr(base + tab(T4_MTT + 2, lane)); p.m3 = lds_addr(base + tab(T4_MTT + 3, lane));
    p.mu = lds_addr(base + tab(T4_MTTU, lane));
    p.ph = lds_addr(base + R_PHIB + 4 + idx);                                   // z index of q row idx: w (4..7), x (8..16); rows 13..15 read finite junk that meets zero columns
    p.cb = lds_addr(base + ((idx >= 4 && idx <= 6) ? R_CB + idx - 4 : R_ZERO)); // corridor parts: pos rows only
    p.pu = lds_addr(base + (b == 0 ? R_PHIB + a : R_ZERO));
    p.hf = lds_addr(base + (b == 0 ? R_HC : R_ONE));
    p.tp = lds_addr(base + R_T + lane);
    p.pd = lds_addr(base + (idx <= 12 ? R_PD + idx : R_ZERO));
    p.wk = lds_addr(base + (b == 0 ? R_T + 16 * a + 13 : R_DUMP));
    p.wp = lds_addr(base + (idx <= 12 ? R_PV + idx : R_DUMP));
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): nothing of the compiler's own LDS traffic is left in flight
    // Every step starts by issuing the gather of the NEXT stage into the other operand set and ends with a wait that ties
    // that set: a load result never crosses a control-flow edge while in flight (the register allocator may copy a value
    // on an edge, and a copy of a register whose load has not landed copies stale data).
    double pv = 0.0;
    BackOps A, B;
    back_gather<1>(p, A); // stage N-1
    back_wait<0>(A);
    back_gather<0>(p, B); // stage N-2
    FRP_SB();
    back_step<1, true>(p, A, smu, pv);
    back_wait<2>(B);
    int s_ = N - 2; // B holds stage s_, the addresses sit on it
    for (; s_ >= 4; s_ -= 4) { // pass: steps at stages s_ .. s_-3, gathers for s_-1 .. s_-4, addresses at s_-4
        p.step(-4 * RSB);
        back_gather<3>(p, A); FRP_SB(); back_step<4, false>(p, B, smu, pv); back_wait<2>(A);
        back_gather<2>(p, B); FRP_SB(); back_step<3, false>(p, A, smu, pv); back_wait<2>(B);
        back_gather<1>(p, A); FRP_SB(); back_step<2, false>(p, B, smu, pv); back_wait<2>(A);
        back_gather<0>(p, B); FRP_SB(); back_step<1, false>(p, A, smu, pv); back_wait<2>(B);
    }
    // stages s_ .. 0 left (s_ <= 3), B holds stage s_: unrolled by count, addresses moved to stage 0 once
    p.step(-s_ * RSB);
    switch (s_) {
    case 3: back_tail<3>(p, B, A, smu, pv); break;
    case 2: back_tail<2>(p, B, A, smu, pv); break;
    case 1: back_tail<1>(p, B, A, smu, pv); break;
    default: back_tail<0>(p, B, A, smu, pv); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // p_w[g] sits in the quad-0 lanes of row g; the stage-0 solve wants it in the lanes (g, 13)
    // (twisted solve: p of the meeting stage is already in its R_PV slots, stored like every stage's)
    if constexpr (!twist) stage0_solve(xs, lane, __shfl(pv, lane & 48));
    WSYNC();
}

// forward sweep: dz for all stages.  du = -T' [hc dw; dx; 1],  ds+ = Mt [du; dx; 1].
struct FwdAddr {
    unsigned m0, m1, m2, m3, ts, mu, hf, wu, ws;
    __device__ __forceinline__ void step(int n)
    {
        m0 += n; m1 += n; m2 += n; m3 += n; ts += n; mu += n; hf += n; wu += n; ws += n;
        asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(ts), "+v"(mu), "+v"(hf), "+v"(wu), "+v"(ws)); // (see BackAddr::step)
    }
};
struct FwdOps {
    double m0, m1, m2, m3, ts, mu, hf;
};
template <int K> // operands of the stage K records above the pass base
__device__ __forceinline__ void fwd_gather(const FwdAddr &p, FwdOps &x)
{
    x.ts = lds_ld<K * RSB>(p.ts);
    x.hf = lds_ld<K * RSB>(p.hf); // quad 0: hc, other quads: 1
    x.m0 = lds_ld<K * RSB>(p.m0);
    x.m1 = lds_ld<K * RSB>(p.m1);
    x.m2 = lds_ld<K * RSB>(p.m2);
    x.m3 = lds_ld<K * RSB>(p.m3);
    x.mu = lds_ld<K * RSB>(p.mu);
}
template <int YOUNGER> // wait until at most YOUNGER LDS operations are outstanding; the operand set is usable afterwards
__device__ __forceinline__ void fwd_wait(FwdOps &x)
{
    asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(x.m0), "+v"(x.m1), "+v"(x.m2), "+v"(x.m3), "+v"(x.ts), "+v"(x.mu), "+v"(x.hf) : "n"(YOUNGER));
}
template <int K>
__device__ __forceinline__ void fwd_step(const FwdAddr &p, const FwdOps &x, double &v)
{
    // v = [dw; dx; 1; 0; 0].  Serial path: v -> du -> v+; the x / affine part of Mt [du; dx; 1] is dealt out into the
    // latency gaps of that path by hand (FRP_SB pins the order: the scheduler does not put the path first).
    const double vin = v;
    const double m = x.hf * v;
    FRP_SB();
    const double v1 = quad_rot<1>(v);
    FRP_SB();
    const double d = mfma4(x.ts, m, 0.0);
    FRP_SB();
    const double v2 = quad_rot<2>(v), v3 = quad_rot<3>(v);
    double acc = mfma4(x.m0, v, 0.0);
    FRP_SB();
    const double r = -d - quad_rot<1>(d);
    FRP_SB();
    acc = mfma4(x.m1, v1, acc);
    FRP_SB();
    const double du = r + quad_rot<2>(r); // du[a] in every lane of row a
    FRP_SB();
    acc = mfma4(x.m2, v2, acc);
    acc = mfma4(x.m3, v3, acc);
    FRP_SB();
    v = mfma4(x.mu, du, acc);
    FRP_SB();
    lds_st<K * RSB>(p.wu, du);
    lds_st<K * RSB>(p.ws, vin); // (the stores and the refill that follows them fill the latency of the closing MFMA)
}

template <int K, int L> // stages K .. L-1 above the addresses; X holds the operands of stage K
__device__ __forceinline__ void fwd_tail(const FwdAddr &p, FwdOps &X, FwdOps &Y, double &v)
{
    if constexpr (K + 1 < L) {
        fwd_gather<K + 1>(p, Y); FRP_SB(); fwd_step<K>(p, X, v); fwd_wait<2>(Y);
        fwd_tail<K + 1, L>(p, Y, X, v);
    } else {
        fwd_step<K>(p, X, v); // the last stage
    }
}

__device__ FRP_SWEEP_LINKAGE void sweep_forward(ldouble *recs, ldouble *xs, int N)
{
    N = uni(N);
    const int lane = threadIdx.x & 63, a = lane >> 4, b = (lane >> 2) & 3;
    const int idx = 4 * b + a; // V layout: the vector row this lane holds
    FwdAddr p;
    p.m0 = lds_addr(recs + tab(T4_MT + 0, lane)); p.m1 = lds_addr(recs + tab(T4_MT + 1, lane));
    p.m2 = lds_addr(recs + tab(T4_MT + 2, lane)); p.m3 = lds_addr(recs + tab(T4_MT + 3, lane));
    p.ts = lds_addr(recs + tab(T4_TS, lane));
    p.mu = lds_addr(recs + tab(T4_MU, lane));
    p.hf = lds_addr(recs + (b == 0 ? R_HC : R_ONE));
    p.wu = lds_addr(recs + (b == 0 ? R_DZ + a : R_DUMP));
    p.ws = lds_addr(recs + (idx <= 12 ? R_DZ + 4 + idx : R_DUMP));
    double v = idx == 13 ? 1.0 : xs[X_DS0 + idx]; // ds_0 (entries 14, 15 are zero), the constant 1 in row 13
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): nothing of the compiler's own LDS traffic is left in flight
    FwdOps A, B;
    fwd_gather<0>(p, A);
    fwd_wait<0>(A);
    int left = N; // stages to go; A holds the operands of the next one, the addresses sit on it
    for (; left >= 5; left -= 4) { // (see sweep_backvec: the next stage's gather opens a step, the wait that ties it closes it)
        fwd_gather<1>(p, B); FRP_SB(); fwd_step<0>(p, A, v); fwd_wait<2>(B);
        fwd_gather<2>(p, A); FRP_SB(); fwd_step<1>(p, B, v); fwd_wait<2>(A);
        fwd_gather<3>(p, B); FRP_SB(); fwd_step<2>(p, A, v); fwd_wait<2>(B);
        fwd_gather<4>(p, A); FRP_SB(); fwd_step<3>(p, B, v); fwd_wait<2>(A);
        p.step(4 * RSB);
    }
    // one to four stages left: unrolled by count (no operand-set copies, no address stepping)
    switch (left) {
    case 4: fwd_tail<0, 4>(p, A, B, v); break;
    case 3: fwd_tail<0, 3>(p, A, B, v); break;
    case 2: fwd_tail<0, 2>(p, A, B, v); break;
    default: fwd_tail<0, 1>(p, A, B, v); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WSYNC();
}

// The plain recursion as the END GAME of the twisted variants (TW_EXACT_BELOW): both sweeps of a phase behind one call
__device__ __noinline__ int endgame_predictor(ldouble *recs, ldouble *xs, int N, double theta FRP_GP_PARAM)
{
    const int fr = sweep_factor<false>(recs, xs, N, theta FRP_GP_ARG(gp));
    if (!fr) sweep_forward(recs, xs, N);
    return fr;
}
__device__ __noinline__ void endgame_corrector(ldouble *recs, ldouble *xs, int N, double smu)
{
    sweep_backvec<false>(recs, xs, N, smu);
    sweep_forward(recs, xs, N);
}

// ================================================================== twisted solve: the first half
// Stages 0 .. m-1 are eliminated FORWARD by the wave that is idle during the sweeps (the model wave) while the Riccati wave runs its
// backward recursion over stages m .. N-1: an arrival-cost recursion in information form,
//     F_{k+1}(s+) = min_{w_k} [ l_k(u_k, w_k, x_k) + F_k(w_k, x_k) ],   [u_k; x_k] = T~_k s+ + t~_k   (T~ from the model wave),
// whose stage is the mirror image of a backward stage -- u and w swap roles -- on the same gathered operands:
//     Z = Q_k + [Phi_w, phi_w; C_xx, c_x]      (stage cost of (w, x) added BEFORE the pivot)
//     pivot on w (4 x 4):  K, T' = [R | Kbar_x | kbar] (gains of the back-substitution),  S = Z - K' D^-1 K  with -hc Kbar_x' in the u columns
//     G^ = [C_u. - hc hc4 [R | Kbar_x | kbar] ; S + C_xu]      (the (u, .) rows analytically, like the w rows of P in the backward stage)
//     X = G^ T~ (+ column 13),  Q_{k+1} = T~' X.
// The pinned x_0 enters as the penalty TW_RHO / 2 |dx_0 - r_0|^2: Q_0 = diag(0, rho I), q_0 = [0; -rho r_0] (the exact recursion is
// rank-deficient for three stages; validated in tools/study/twisted_riccati.py and, as a whole solver, in oracle/nmpc_ipm.c).
// Packed Q_k (lower triangle) is stored in stage k's overlay slots for the multipliers y_k = -(Q_k ds_k + q_k); Q_m goes to the workgroup
// scratch where both halves read it.  Returns 1 when a pivot block is not positive definite.
__device__ __noinline__ int sweep_arrive(ldouble *recs, ldouble *xs, ldouble *tw, int m, double theta)
{
    m = uni(m); theta = uni(theta);
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, c3_ = c & 3;
    int mo[4], c1[4], c2[4], c3[4], ppo[4], pdo[4], sqo[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        mo[r] = tab(T_MA + r, lane); c1[r] = tab(T_C1 + r, lane); c2[r] = tab(T_C2 + r, lane);
        c3[r] = tab(T_C3 + r, lane); ppo[r] = tab(T_PP + r, lane); pdo[r] = tab(T_PD + r, lane); sqo[r] = tab(T_SQ + r, lane);
    }
    const int pqo = (c < 4 && g == c) ? R_PHID + 4 + g : (c == 13 ? R_PHI + 4 + g : R_ZERO);
    const int mhi = g > c3_ ? g : c3_, mlo = g > c3_ ? c3_ : g;
    const int msel = mhi == mlo ? 6 : mhi * (mhi - 1) / 2 + mlo;
    const bool m_lower = c3_ <= g, m_upper = g <= c3_;
    const double m4 = c < 4 ? 1.0 : 0.0, m4c = 1.0 - m4, m13 = c == 13 ? 1.0 : 0.0;
    const int to = c < 14 ? R_T + lane : R_DUMP; // (columns 14, 15 of T' hold part of B~ in a first-half record)
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 Qn = zero;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int trow = 4 * r + g;
        if (trow >= 4 && trow <= 12) {
            if (c == trow) Qn[r] = TW_RHO;
            if (c == 13) Qn[r] = -TW_RHO * xs[X_DX0 + trow - 4];
        }
    }
    bool ok = true;
    for (int k = 0; k < m; k++) {
        ldouble *rec = recs + k * RS;
        d4 C, Mt;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            C[r] = r == 1 ? rec[c1[r]] + rec[c2[r]] + theta * rec[c3[r]] : rec[c1[r]] + theta * rec[c3[r]];
            Mt[r] = rec[mo[r]];
        }
        const double hc = rec[R_HC], pq = rec[pqo];
        // Q_k (packed lower triangle) for y_k; it overwrites the Hessian pieces of this stage, gathered above
#pragma unroll
        for (int r = 0; r < 4; r++) rec[ppo[r]] = Qn[r];
        d4 Z;
        Z[0] = Qn[0] + pq;
        Z[1] = __builtin_fma(m4c, C[1], Qn[1]); Z[2] = __builtin_fma(m4c, C[2], Qn[2]); Z[3] = __builtin_fma(m4c, C[3], Qn[3]);
        double q[16], Mi[6], Di[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) q[i * 4 + j] = lane_bcast(Z[0], 16 * i + j);
        ok &= ldl4(q, Mi, Di);
        double me = msel == 6 ? 1.0 : Mi[0];
#pragma unroll
        for (int i = 1; i < 6; i++) me = msel == i ? Mi[i] : me;
        const double m_gc = m_lower ? me : 0.0, m_cg = m_upper ? me : 0.0;
        double dg = Di[3];
        dg = g == 2 ? Di[2] : dg; dg = g == 1 ? Di[1] : dg; dg = g == 0 ? Di[0] : dg;
        const double md = dg * m_gc;
        const double K0 = mfma4(m_cg, Z[0], 0.0);
        const double Kd = dg * K0;
        const double tsel = mfma4(m_gc, c < 4 ? md : Kd, 0.0);
        const double hcm = hc * m4;
        const double Kb = __builtin_fma(hcm, m_gc, K0 * m4c);
        d4 S = Z;
        S[1] *= m4c; S[2] *= m4c; S[3] *= m4c;
        S = __builtin_amdgcn_mfma_f64_16x16x4f64(-Kd, Kb, S, 0, 0, 0);
        rec[to] = tsel;
        const double hc4 = __builtin_fma(hc, m4, m4c);
        d4 Gh;
        Gh[0] = __builtin_fma(-hc, hc4 * tsel, C[0]);
        Gh[1] = __builtin_fma(m4, C[1], S[1]); Gh[2] = __builtin_fma(m4, C[2], S[2]); Gh[3] = __builtin_fma(m4, C[3], S[3]);
        d4 X = __builtin_amdgcn_mfma_f64_16x16x4f64(Gh[0], Mt[0], zero, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f64_16x16x4f64(Gh[1], Mt[1], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f64_16x16x4f64(Gh[2], Mt[2], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f64_16x16x4f64(Gh[3], Mt[3], X, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            rec[pdo[r]] = X[r];                        // G^ t~ (column 13) for the vector sweep
            X[r] = __builtin_fma(m13, Gh[r], X[r]);    // + g^ in column 13
        }
        d4 Cz = zero;
        Cz[0] = X[0]; // the rows u of T~ are [I 0 | t~_w]
        Qn = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[1], X[1], Cz, 0, 0, 0);
        Qn = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[2], X[2], Qn, 0, 0, 0);
        Qn = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[3], X[3], Qn, 0, 0, 0);
    }
    // (Q_m, q_m) where both halves read it
#pragma unroll
    for (int r = 0; r < 4; r++) tw[sqo[r]] = Qn[r];
    WSYNC();
    return ok ? 0 : 1;
}

// Where the halves meet: ds_m = -(Q_m + P_m)^-1 (q_m + p_m), a 13 x 13 symmetric positive definite system, rows in lanes (lane i holds
// row i).  P_m / p_m come from the meeting stage's record (packed, written by the backward half), Q_m / q_m from the workgroup scratch.
// The factor L D L' of the meeting system lives in the workgroup scratch (L strictly lower, packed by rows, and 1 / D): the Riccati wave
// factors once per iteration (predictor), every solve reads it from there -- lane i its row L[i][.] for the first triangular solve, its
// column L[.][i] for the second -- so nothing of it is carried in registers across the phases of an iteration.
__device__ __forceinline__ int tw_l(int i, int j) { return TW_L + i * (i - 1) / 2 + j; } // i > j
__device__ __noinline__ int meet_factor(ldouble *recs, ldouble *tw, int m)
{
    m = uni(m);
    const int lane = threadIdx.x & 63, i = lane < 13 ? lane : 12;
    cldouble *rm = recs + m * RS;
    int fail = 0;
    double a[13];
#pragma unroll
    for (int j = 0; j < 13; j++) {
        const int hi = i > j ? i : j, lo = i > j ? j : i, e = hi * (hi + 1) / 2 + lo;
        a[j] = rm[R_P + e] + tw[TW_Q + e];
    }
    // right-looking, column by column: lane i holds row i; column j of the current Schur complement is a[j] in the lanes i >= j
    const int rowb = TW_L + i * (i - 1) / 2;
#pragma unroll
    for (int j = 0; j < 13; j++) {
        const double dj = lane_bcast(a[j], j);
        if (!(dj > 0.0)) fail = 1;
        const double inv = fast_rcp(dj);
        const double lij = a[j] * inv;
#pragma unroll
        for (int kq = j + 1; kq < 13; kq++) a[kq] = __builtin_fma(-lij, lane_bcast(a[j], kq), a[kq]); // A[i][kq] -= L[i][j] A[kq][j]
        if (lane > j && lane < 13) tw[rowb + j] = lij;
        if (lane == j) tw[TW_DINV + j] = inv;
    }
    WSYNC();
    return fail;
}
// ds_m = -(Q_m + P_m)^-1 (q_m + p_m) from the stored factor; ds: 16 slots (13 + three zeros: rows 13..15 of the sweep vectors)
__device__ __noinline__ void meet_solve(ldouble *recs, ldouble *tw, int m, ldouble *ds)
{
    m = uni(m);
    const int lane = threadIdx.x & 63, i = lane < 13 ? lane : 12;
    cldouble *rm = recs + m * RS;
    double l[12], lt[13];
#pragma unroll
    for (int j = 0; j < 12; j++) l[j] = tw[j < i ? tw_l(i, j) : TW_ZERO];        // L[i][j], j < i
#pragma unroll
    for (int j = 1; j < 13; j++) lt[j] = tw[(j > i && lane < 13) ? tw_l(j, i) : TW_ZERO]; // L[j][i], j > i
    const double dinv = tw[TW_DINV + i];
    double b = -(rm[R_PV + i] + tw[TW_QV + i]);
    // L y = b
#pragma unroll
    for (int j = 0; j < 12; j++) b = __builtin_fma(-l[j], lane_bcast(b, j), b);
    b *= dinv;
    // L' x = D^-1 y
#pragma unroll
    for (int j = 12; j >= 1; j--) b = __builtin_fma(-lt[j], lane_bcast(b, j), b);
    if (lane < 16) ds[lane] = lane < 13 ? b : 0.0;
    WSYNC();
}

// first half, vector sweep of the corrector (new right-hand side phi_cc = PHIB + smu PHIC): the arrival gradient q_k, stage by stage,
//     z~ = phi_(w, x) + q_k,  E = T'' z~_w,  g^ = [phi_u - hc E_w; z~_x - E_x],  kbar = E_w -> T' column 13,  q_{k+1} = T~' (G^ t~ + g^).
// Vectors in V layout (lane 16 a + 4 b + j holds row 4 b + a).  q_k is stored in stage k's R_PV slots (multipliers), q_m in the workgroup scratch.
// Same construction as the Riccati wave's vector sweeps: per-lane byte addresses, the stage offset as the instruction's immediate,
// two operand sets refilled one stage ahead, waits that count the younger LDS operations exactly.
struct ArrAddr {
    unsigned m0, m1, m2, m3, ph, cb, cc, pu, hf, tp, pd, wk, wq;
    __device__ __forceinline__ void step(int n)
    {
        m0 += n; m1 += n; m2 += n; m3 += n; ph += n; cb += n; cc += n; pu += n; hf += n; tp += n; pd += n; wk += n; wq += n;
        asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(ph), "+v"(cb), "+v"(cc), "+v"(pu), "+v"(hf), "+v"(tp), "+v"(pd), "+v"(wk), "+v"(wq));
    }
};
struct ArrOps {
    double m0, m1, m2, m3, phb, phc, cbb, cbc, pub, puc, hf, tp, pd;
};
template <int K>
__device__ __forceinline__ void arr_gather(const ArrAddr &p, ArrOps &x)
{
    x.phb = lds_ld<K * RSB>(p.ph);
    x.phc = lds_ld<K * RSB + R_BC * 8>(p.ph);
    x.cbb = lds_ld<K * RSB>(p.cb);               // corridor parts: pos rows, 0 elsewhere (two addresses: a first-half record's second zero is in use)
    x.cbc = lds_ld<K * RSB>(p.cc);
    x.tp = lds_ld<K * RSB>(p.tp);
    x.pub = lds_ld<K * RSB>(p.pu);
    x.puc = lds_ld<K * RSB + R_BC * 8>(p.pu);
    x.hf = lds_ld<K * RSB>(p.hf);                // quad 0: hc, other quads: 1
    x.pd = lds_ld<K * RSB>(p.pd);
    x.m0 = lds_ld<K * RSB>(p.m0);
    x.m1 = lds_ld<K * RSB>(p.m1);
    x.m2 = lds_ld<K * RSB>(p.m2);
    x.m3 = lds_ld<K * RSB>(p.m3);
}
template <int YOUNGER>
__device__ __forceinline__ void arr_wait(ArrOps &x)
{
    asm volatile("s_waitcnt lgkmcnt(%13)"
                 : "+v"(x.m0), "+v"(x.m1), "+v"(x.m2), "+v"(x.m3), "+v"(x.phb), "+v"(x.phc), "+v"(x.cbb), "+v"(x.cbc), "+v"(x.pub), "+v"(x.puc),
                   "+v"(x.hf), "+v"(x.tp), "+v"(x.pd)
                 : "n"(YOUNGER));
}
template <int K>
__device__ __forceinline__ void arr_step(const ArrAddr &p, const ArrOps &x, double smu, bool q0, double &qk)
{
    lds_st<K * RSB>(p.wq, qk);                                              // q_k for y_k = -(Q_k ds_k + q_k)
    const double zt = __builtin_fma(smu, x.phc + x.cbc, x.phb + x.cbb) + qk; // z~ (rows 13..15: junk that meets zero columns)
    double dq = q0 ? zt : 0.0;                                               // z~_w[a] into every lane of row a
    dq += quad_rot<1>(dq);
    dq += quad_rot<2>(dq);
    const double E = mfma4(x.tp, dq, 0.0);
    const double base = q0 ? __builtin_fma(smu, x.puc, x.pub) : zt;
    const double gh = __builtin_fma(-x.hf, E, base);
    const double v = gh + x.pd;
    FRP_SB();
    lds_st<K * RSB>(p.wk, E); // kbar (rows 0..3); behind VALU results computed from E (see back_step)
    d4 A;
    A[0] = x.m0; A[1] = x.m1; A[2] = x.m2; A[3] = x.m3;
    qk = matvec4s(A, v, 0.0);
}
template <int K, int L> // stages K .. L-1 above the addresses; X holds the operands of stage K
__device__ __forceinline__ void arr_tail(const ArrAddr &p, ArrOps &X, ArrOps &Y, double smu, bool q0, double &qk)
{
    if constexpr (K + 1 < L) {
        arr_gather<K + 1>(p, Y); FRP_SB(); arr_step<K>(p, X, smu, q0, qk); arr_wait<2>(Y);
        arr_tail<K + 1, L>(p, Y, X, smu, q0, qk);
    } else {
        arr_step<K>(p, X, smu, q0, qk);
    }
}
__device__ __noinline__ void sweep_arrive_vec(ldouble *recs, ldouble *xs, ldouble *tw, int m, double smu)
{
    m = uni(m); smu = uni(smu);
    const int lane = threadIdx.x & 63, a = lane >> 4, b = (lane >> 2) & 3;
    const int idx = 4 * b + a;
    const bool q0 = b == 0;
    ArrAddr p;
    p.m0 = lds_addr(recs + tab(T4_MAT + 0, lane)); p.m1 = lds_addr(recs + tab(T4_MAT + 1, lane));
    p.m2 = lds_addr(recs + tab(T4_MAT + 2, lane)); p.m3 = lds_addr(recs + tab(T4_MAT + 3, lane));
    p.ph = lds_addr(recs + R_PHIB + 4 + idx);                                    // (w, x) rows: z index 4 + idx; rows 13..15 read finite junk
    p.cb = lds_addr(recs + ((idx >= 4 && idx <= 6) ? R_CB + idx - 4 : R_ZERO));  // corridor parts on the pos rows
    p.cc = lds_addr(recs + ((idx >= 4 && idx <= 6) ? R_CC + idx - 4 : R_ZERO));
    p.pu = lds_addr(recs + R_PHIB + a);                                          // phi_u[a] (used by the quad-0 lanes)
    p.hf = lds_addr(recs + (q0 ? R_HC : R_ONE));
    p.tp = lds_addr(recs + R_T + lane);
    p.pd = lds_addr(recs + (idx <= 12 ? R_PD + idx : R_ZERO));
    p.wk = lds_addr(recs + (q0 ? R_T + 16 * a + 13 : R_DUMP));
    p.wq = lds_addr(recs + (idx <= 12 ? R_PV + idx : R_DUMP));
    double qk = (idx >= 4 && idx <= 12) ? -TW_RHO * xs[X_DX0 + idx - 4] : 0.0;
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): nothing of the compiler's own LDS traffic is left in flight
    ArrOps A, B;
    arr_gather<0>(p, A);
    arr_wait<0>(A);
    int left = m; // stages to go; A holds the operands of the next one, the addresses sit on it
    for (; left >= 5; left -= 4) {
        arr_gather<1>(p, B); FRP_SB(); arr_step<0>(p, A, smu, q0, qk); arr_wait<2>(B);
        arr_gather<2>(p, A); FRP_SB(); arr_step<1>(p, B, smu, q0, qk); arr_wait<2>(A);
        arr_gather<3>(p, B); FRP_SB(); arr_step<2>(p, A, smu, q0, qk); arr_wait<2>(B);
        arr_gather<4>(p, A); FRP_SB(); arr_step<3>(p, B, smu, q0, qk); arr_wait<2>(A);
        p.step(4 * RSB);
    }
    switch (left) {
    case 4: arr_tail<0, 4>(p, A, B, smu, q0, qk); break;
    case 3: arr_tail<0, 3>(p, A, B, smu, q0, qk); break;
    case 2: arr_tail<0, 2>(p, A, B, smu, q0, qk); break;
    default: arr_tail<0, 1>(p, A, B, smu, q0, qk); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (idx <= 12 && (lane & 3) == 0) tw[TW_QV + idx] = qk;
    WSYNC();
}

// first half, back-substitution (both passes): from v = [ds_m; 1] down to stage 0,
//     [u; x]_k = T~ v (+ t~ through row 13),  w_k = -T' [hc u; x; 1],  dz_k -> the record,  v <- [w_k; x_k; 1].
// (m >= 2; the addresses sit on the LOWEST stage a pass touches, see sweep_backvec)
struct BsAddr {
    unsigned m0, m1, m2, m3, ts, hf, wu, ww, wx;
    __device__ __forceinline__ void step(int n)
    {
        m0 += n; m1 += n; m2 += n; m3 += n; ts += n; hf += n; wu += n; ww += n; wx += n;
        asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(ts), "+v"(hf), "+v"(wu), "+v"(ww), "+v"(wx));
    }
};
struct BsOps {
    double m0, m1, m2, m3, ts, hf;
};
template <int K>
__device__ __forceinline__ void bs_gather(const BsAddr &p, BsOps &x)
{
    x.m0 = lds_ld<K * RSB>(p.m0);
    x.m1 = lds_ld<K * RSB>(p.m1);
    x.m2 = lds_ld<K * RSB>(p.m2);
    x.m3 = lds_ld<K * RSB>(p.m3);
    x.hf = lds_ld<K * RSB>(p.hf); // quad 0: hc, other quads: 1
    x.ts = lds_ld<K * RSB>(p.ts);
}
template <int YOUNGER>
__device__ __forceinline__ void bs_wait(BsOps &x)
{
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(x.m0), "+v"(x.m1), "+v"(x.m2), "+v"(x.m3), "+v"(x.ts), "+v"(x.hf) : "n"(YOUNGER));
}
template <int K>
__device__ __forceinline__ void bs_step(const BsAddr &p, const BsOps &x, bool q0, double &v)
{
    d4 A;
    A[0] = x.m0; A[1] = x.m1; A[2] = x.m2; A[3] = x.m3;
    const double ux = matvec4s(A, v, 0.0);      // rows 0..3 u, 4..12 x, 13: 1
    const double mm = x.hf * ux;
    FRP_SB();
    const double d = mfma4(x.ts, mm, 0.0);
    FRP_SB();
    const double r = -d - quad_rot<1>(d);
    const double wk = r + quad_rot<2>(r);       // w_k[a] in every lane of row a
    v = q0 ? wk : ux;
    FRP_SB();
    lds_st<K * RSB>(p.wu, ux);
    lds_st<K * RSB>(p.ww, wk);
    lds_st<K * RSB>(p.wx, ux);
}
template <int K> // stages K .. 0 above the addresses; X holds the operands of stage K
__device__ __forceinline__ void bs_tail(const BsAddr &p, BsOps &X, BsOps &Y, bool q0, double &v)
{
    if constexpr (K > 0) {
        bs_gather<K - 1>(p, Y); FRP_SB(); bs_step<K>(p, X, q0, v); bs_wait<3>(Y);
        bs_tail<K - 1>(p, Y, X, q0, v);
    } else {
        bs_step<0>(p, X, q0, v); // stage 0
    }
}
__device__ __noinline__ void sweep_backsub(ldouble *recs, cldouble *ds, int m)
{
    m = uni(m);
    const int lane = threadIdx.x & 63, a = lane >> 4, b = (lane >> 2) & 3;
    const int idx = 4 * b + a;
    const bool q0 = b == 0;
    ldouble *base = recs + (m - 2) * RS;
    BsAddr p;
    p.m0 = lds_addr(base + tab(T4_MA + 0, lane)); p.m1 = lds_addr(base + tab(T4_MA + 1, lane));
    p.m2 = lds_addr(base + tab(T4_MA + 2, lane)); p.m3 = lds_addr(base + tab(T4_MA + 3, lane));
    p.ts = lds_addr(base + tab(T4_TS, lane));
    p.hf = lds_addr(base + (q0 ? R_HC : R_ONE));
    p.wu = lds_addr(base + (q0 ? R_DZ + a : R_DUMP));                            // du = u_k (rows 0..3 of T~ v)
    p.ww = lds_addr(base + (q0 ? R_DZ + 4 + a : R_DUMP));                        // dw_k
    p.wx = lds_addr(base + ((idx >= 4 && idx <= 12) ? R_DZ + 4 + idx : R_DUMP)); // dx_k
    double v = idx == 13 ? 1.0 : ds[idx];
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): nothing of the compiler's own LDS traffic is left in flight
    BsOps A, B;
    bs_gather<1>(p, A); // stage m-1
    bs_wait<0>(A);
    bs_gather<0>(p, B); // stage m-2
    FRP_SB();
    bs_step<1>(p, A, q0, v);
    bs_wait<3>(B);
    int s_ = m - 2; // B holds stage s_, the addresses sit on it
    for (; s_ >= 4; s_ -= 4) {
        p.step(-4 * RSB);
        bs_gather<3>(p, A); FRP_SB(); bs_step<4>(p, B, q0, v); bs_wait<3>(A);
        bs_gather<2>(p, B); FRP_SB(); bs_step<3>(p, A, q0, v); bs_wait<3>(B);
        bs_gather<1>(p, A); FRP_SB(); bs_step<2>(p, B, q0, v); bs_wait<3>(A);
        bs_gather<0>(p, B); FRP_SB(); bs_step<1>(p, A, q0, v); bs_wait<3>(B);
    }
    p.step(-s_ * RSB);
    switch (s_) {
    case 3: bs_tail<3>(p, B, A, q0, v); break;
    case 2: bs_tail<2>(p, B, A, q0, v); break;
    case 1: bs_tail<1>(p, B, A, q0, v); break;
    default: bs_tail<0>(p, B, A, q0, v); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WSYNC();
}

// ================================================================== wave 1: model (lane == stage)
// Owns the iterate z and the equality multipliers y of its stage.  Heun step + compact Jacobian -> the stage record,
// equality residual, and the M'y part of the stationarity residual.  d needs the next stage's [w; x] and gm its
// multipliers: lane k+1 hands them over.  (The exact Hessian of the same step is evaluated by wave 3 at the same time.)
struct ModelState {
    double z[NZ], y[NS], fext[3];
};

// Register budget (168 per lane with z and y resident): the phase is cut into sections that hand data to each other
// through the T' slots of the stage record, which are dead between the last forward sweep and the next factorisation:
// J1 (21) while J2 is formed, y (13) during the whole linearisation.
// The Hessian lanes' inputs: RT_HU (rates, T: 4), RT_HVE (v, e: 6), RT_HY (y_p, y_v: 6) before the step, RT_HYP (y+ rows 4..9: 6).
#if !defined(FRP_QL) && !defined(FRP_QP)
constexpr int RT_J1 = R_T, RT_Y = R_T + 24;
// after the last forward sweep of an iteration wave 1 leaves here what wave 0 needs to rebuild its Hessian inputs for the
// next one (so that nothing of it occupies registers across the sweeps): (rates, T, v, e) before the step, (y_p, y_v) too
// (placed clear of RT_Y -- written at the start of the model phase while the Hessian lanes read these -- and of the columns 14, 15
// of T', which hold part of B~ in a first-half record)
constexpr int RT_HU = R_T + 48, RT_HVE = R_T + 52, RT_HY = R_T + 37, RT_HYP = R_DUMP /* (in registers) */, RT_YV = R_DUMP;
static_assert(RT_HY >= RT_Y + 13 && RT_HY + 6 <= R_T + 46 && RT_HVE + 6 <= R_T + 62, "scratch in the T' slots");
constexpr int RX_MTRIG = R_PV, RX_HTRIG = R_PD; // trig hand-over of the model wave / of the Riccati wave's Hessian lanes (12 each)
#else
// QP: the y+ lanes (bounds wave) read Kbar_x and kbar of T' in the step phase, while the model wave leaves the Hessian's inputs: those go
// to the T' slots the y+ lanes do not read -- R (columns 0..3) and the zero columns 14, 15 of the rows 0..2.  The y+ lanes' own scratch
// (block hand-over RT_YV, 9; y+ rows 4..9 for the Hessian lanes RT_HYP, 6) takes Kbar_x slots AFTER their last read of T' (same wave).
constexpr int RT_HU = R_T + 0, RT_HVE = R_T + 14, RT_HY = R_T + 30, RT_YV = R_T + 4, RT_HYP = R_T + 20;
#ifndef FRP_QS
constexpr int RT_J1 = R_T; // (the lane == stage model phase: not built with QP)
#endif
#ifndef FRP_QL
constexpr int RT_Y = R_T + 36; // (bisection builds: T' is a region of its own, dead in the evaluation phase)
constexpr int RX_MTRIG = R_PV, RX_HTRIG = R_PD;
#else
// QL: T' shares the Hessian's slots, which the element-wise waves fill during the evaluation phase -- the model wave's scratch of that
// phase has slots outside the overlay (y in the P d slots, dead from the corrector's backward sweep to the next factorisation; the trig
// hand-over at the record's end); what the Riccati wave reads at the start of the evaluation (RT_H*) and its trig hand-over sit inside HD,
// which only that wave writes in the phase (the hand-over overlaps RT_HYP / RT_HY: written after they were read, by the same wave)
constexpr int RT_Y = R_PD; // (QS: unused -- y stays in the model wave's registers, FRP_NO_YPARK)
#ifndef FRP_Q4_PARK // 1: the 12 slots at the record's end are the corridor lanes' parking (see PARK in solve_one); the model wave's trig hand-over uses the d slots instead --
#define FRP_Q4_PARK 1 // dead from that wave's commit (they carried y+) to its own store of d, a few hundred instructions behind the hand-over
#endif
#if !defined(FRP_QS) && FRP_Q4_PARK
constexpr int RX_HTRIG = R_T + 20, RX_MTRIG = R_D;
#else
constexpr int RX_HTRIG = R_T + 20, RX_MTRIG = RQ_MTRIG;
#endif
static_assert(RX_HTRIG + 12 <= R_HD + REC_HD_SIZE && RT_HY + 6 <= R_HD + REC_HD_SIZE, "the Riccati wave's scratch inside HD");
#endif
#endif
// where the lane == stage model phase parks J1 = [F_vv (9) | F_ve (9) | g_T (3)] of the first RK2 point while J2 is formed.  QS: T' shares the Hessian's slots, which
// the other waves fill during this phase -- the scratch is what THIS wave writes last: the M'y slots (PHIC, written at the end of the phase) and the d slots (written
// behind the J2 J1 products; they carried y+ until this wave's commit)
#ifndef FRP_QS
constexpr int RT_J1V = RT_J1, RT_J1E = RT_J1 + 9, RT_J1T = RT_J1 + 18;
#else
constexpr int RT_J1V = R_PHIC, RT_J1E = R_D, RT_J1T = R_D + 9;
#endif
template <int NP>
__device__ __forceinline__ void model_phase(ldouble *recs, ldouble *xs, ModelState &st, int N, double &l_eq)
{
    const int lane = threadIdx.x & 63;
    const int k = lane;
    const double *zk = st.z;
    l_eq = 0.0;
    ldouble *rec = recs + (k < N ? k : 0) * RS;
    const bool dyn = k < N - 1;
    SEG_DECL();
    // ---- part 1: the step and its linearisation (no multipliers involved)
#ifndef FRP_NO_YPARK
    if (k < N) {
#pragma unroll
        for (int i = 0; i < NS; i++) rec[RT_Y + i] = st.y[i];
    }
#endif
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double dx = xs[X_XINIT + i] - zk[8 + i];
            xs[X_DX0 + i] = dx;
            l_eq = fmax(l_eq, fabs(dx));
        }
    }
    SEG(0);
    double xn[9]; // x+ = RK2 step of (p, v, e)
#pragma unroll
    for (int i = 0; i < 9; i++) xn[i] = 0.0;
    if (dyn) {
        double a1[3], vt[3], et[3];
        {
            AccJac J1;
            const Trig tg1 = make_trig(zk + 14);
            accel_t<true>(zk + 11, tg1, zk[3], st.fext, a1, &J1);
#pragma unroll
            for (int i = 0; i < 9; i++) {
                rec[RT_J1V + i] = J1.Fvv[i];
                rec[RT_J1E + i] = J1.Fve[i];
                rec[R_LIN + i] = (i % 4 == 0 ? DT : 0.0) + 0.5 * DT * DT * J1.Fvv[i]; // Apv (i % 4 == 0: the diagonal of a row-major 3 x 3)
                rec[R_LIN + 9 + i] = 0.5 * DT * DT * J1.Fve[i];                       // Ape
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                rec[RT_J1T + i] = J1.gT[i];
                rec[R_LIN + 36 + i] = 0.5 * DT * DT * J1.gT[i];                        // BpT
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        SEG(1);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            vt[i] = zk[11 + i] + DT * a1[i];
            et[i] = zk[14 + i] + DT * zk[i];
        }
        AccJac J2;
        double a2[3];
        {
            const Trig tg2 = make_trig(et);
            accel_t<true>(vt, tg2, zk[3], st.fext, a2, &J2);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            xn[i] = zk[8 + i] + 0.5 * DT * (zk[11 + i] + vt[i]);
            xn[3 + i] = zk[11 + i] + 0.5 * DT * (a1[i] + a2[i]);
            xn[6 + i] = et[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        SEG(2);
        // products J2 J1, one column j of J1 at a time (read back from the record)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double f0 = rec[RT_J1V + 0 + j], f1 = rec[RT_J1V + 3 + j], f2 = rec[RT_J1V + 6 + j];   // J1.Fvv[:, j]
            const double e0 = rec[RT_J1E + 0 + j], e1 = rec[RT_J1E + 3 + j], e2 = rec[RT_J1E + 6 + j]; // J1.Fve[:, j]
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double sv = J2.Fvv[i * 3 + j] + DT * (J2.Fvv[i * 3 + 0] * f0 + J2.Fvv[i * 3 + 1] * f1 + J2.Fvv[i * 3 + 2] * f2);
                const double se = J2.Fve[i * 3 + j] + DT * (J2.Fvv[i * 3 + 0] * e0 + J2.Fvv[i * 3 + 1] * e1 + J2.Fvv[i * 3 + 2] * e2);
                const double fij = i == 0 ? f0 : (i == 1 ? f1 : f2), eij = i == 0 ? e0 : (i == 1 ? e1 : e2);
                rec[R_LIN + 18 + i * 3 + j] = (i == j ? 1.0 : 0.0) + 0.5 * DT * (fij + sv); // Avv
                rec[R_LIN + 27 + i * 3 + j] = 0.5 * DT * (eij + se);                         // Ave
                rec[R_LIN + 42 + i * 3 + j] = 0.5 * DT * DT * J2.Fve[i * 3 + j];             // Bvw
            }
        }
        {
            const double g0 = rec[RT_J1T], g1 = rec[RT_J1T + 1], g2 = rec[RT_J1T + 2]; // J1.gT
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double sT = J2.gT[i] + DT * (J2.Fvv[i * 3 + 0] * g0 + J2.Fvv[i * 3 + 1] * g1 + J2.Fvv[i * 3 + 2] * g2);
                const double gi = i == 0 ? g0 : (i == 1 ? g1 : g2);
                rec[R_LIN + 39 + i] = 0.5 * DT * (gi + sT); // BvT
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    SEG(3);
    // ---- d = prev(z_k) - s_{k+1}: the next stage's [w; x] comes from lane k+1
    {
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double zn = dpp_move<0x130>(st.z[4 + i]); // wave_shl:1, lane k <- lane k+1
            const double d = zk[i] - zn;
            if (k < N) rec[R_D + i] = dyn ? d : 0.0; // (the last stage has no successor: d = 0; the slots carried y+ through the step phase)
            dmax = fmax(dmax, fabs(d));
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double zn = dpp_move<0x130>(st.z[8 + i]);
            const double d = xn[i] - zn;
            if (k < N) rec[R_D + 4 + i] = dyn ? d : 0.0;
            dmax = fmax(dmax, fabs(d));
        }
        if (dyn) l_eq = fmax(l_eq, dmax);
    }
    __builtin_amdgcn_sched_barrier(0);
    SEG(4);
    // ---- part 2: gm = M' y_{k+1} - [0; y_k], the linearisation read back from the record this lane has just written
    {
#ifndef FRP_NO_YPARK
        if (k < N) {
#pragma unroll
            for (int i = 0; i < NS; i++) st.y[i] = rec[RT_Y + i];
        }
#endif
        double yn[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) yn[i] = dpp_move<0x130>(st.y[i]);
        if (k < N) {
            double gm[NZ];
#pragma unroll
            for (int i = 0; i < 4; i++) gm[i] = 0.0;
#pragma unroll
            for (int i = 0; i < NS; i++) gm[4 + i] = -st.y[i];
            if (dyn) {
                const double *yw = yn, *yp = yn + 4, *yv = yn + 7, *ye = yn + 10;
#pragma unroll
                for (int i = 0; i < 4; i++) gm[i] += yw[i];
                double gT = 0.0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    gm[i] += DT * ye[i];
                    gm[8 + i] += yp[i];
                    gm[14 + i] += ye[i];
                    double apv[3], ape[3], avv[3], ave[3], bvw[3]; // row i of the five 3 x 3 blocks: one batch of LDS reads
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        apv[j] = rec[R_LIN + i * 3 + j]; ape[j] = rec[R_LIN + 9 + i * 3 + j]; avv[j] = rec[R_LIN + 18 + i * 3 + j];
                        ave[j] = rec[R_LIN + 27 + i * 3 + j]; bvw[j] = rec[R_LIN + 42 + i * 3 + j];
                    }
                    const double bpt = rec[R_LIN + 36 + i], bvt = rec[R_LIN + 39 + i];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        gm[j] += bvw[j] * yv[i];
                        gm[11 + j] += apv[j] * yp[i] + avv[j] * yv[i];
                        gm[14 + j] += ape[j] * yp[i] + ave[j] * yv[i];
                    }
                    gT += bpt * yp[i] + bvt * yv[i];
                }
                gm[3] += gT;
            }
#pragma unroll
            // (PHIC is free from the step phase to the affine phase; the Newton step's slots are NOT: wave 0 starts the
            // predictor's forward sweep, which writes them, while the other waves may still be summing the residual)
            for (int i = 0; i < NZ; i++) rec[R_PHIC + i] = gm[i];
        }
    }
    SEG(5);
    SEGM_FLUSH();
}

// ================================================================== wave 3, lanes of face group 0: exact Hessian (lane == stage)
// Keeps its own copy of what the Hessian of y_{k+1}' c(z_k) depends on: rates, thrust, velocity, attitude of the
// stage and the pos / vel multipliers of the next one (updated from the same dz / y+ as the owners' copies).
struct HessState {
    double u[4], ve[6], y6[6], fext[3]; // u = (rates, T); ve = (v, e); y6 = (y_p, y_v) of stage k+1
};
__device__ __forceinline__ void hessian_phase(ldouble *rec, const HessState &hs, bool dyn, int hess)
{
    if (dyn && hess) {
        double x[9];
        x[0] = x[1] = x[2] = 0.0; // the position does not enter
#pragma unroll
        for (int i = 0; i < 6; i++) x[3 + i] = hs.ve[i];
        rk2_hessian(x, hs.u, hs.fext, hs.y6, hs.y6 + 3, [&](int i, int j, double val) {
            if (hd_index(i, j) >= 0) rec[R_HD + hd_index(i, j)] = val;
        });
    } else {
        // the Hessian slots are overwritten by P every iteration: stages without a dynamics Hessian clear them again
#pragma unroll
        for (int i = 0; i < REC_HD_SIZE; i++) rec[R_HD + i] = 0.0;
    }
}

// ================================================================== three lanes per stage (NP = 20)
// The evaluation phase is the part of an iteration in which the Riccati wave waits for the helpers, and with lane ==
// stage the model and the Hessian are 1.5-2 k instructions each on 20 of 64 lanes.  With NP = 20 three lanes share a stage
// (lane = sub * NP + k, like the bound rows): every lane keeps a copy of the stage's state and evaluates the part of the
// Jacobian / Hessian that belongs to ONE attitude angle -- column `sub` of the blocks d(.)/d(v, e) and of the products with
// them -- with the same instruction stream for the three angles:
//   d/d(angle j) of a product of sines / cosines that contains the angle once = the product with (s_j, c_j) -> (c_j, -s_j),
// so a lane differentiates by substituting ITS angle in the trig set (six selects) and evaluating zB again (TrigK).  What a lane
// needs of the other two (their sine / cosine, s_j and a_j of the Hessian, a column of Ke) goes through a dozen scratch
// slots of the stage record that are dead in this phase.  Products with J = d acc / d v use its structure,
// J c = d (zB (zB . c) - c), instead of nine multiply-adds per column.
__device__ __forceinline__ double pick3(int sub, double a0, double a1, double a2) { return sub == 0 ? a0 : (sub == 1 ? a1 : a2); }
__device__ __forceinline__ int pick3i(int sub, int a0, int a1, int a2) { return sub == 0 ? a0 : (sub == 1 ? a1 : a2); }
// zB = (cy sp cr + sy sr, sy sp cr - cy sr, cp cr) and its derivatives.  A derivative by one angle replaces that angle's
// (sin, cos) by (cos, -sin) in the terms that contain it and removes the terms that do not: the "sr" terms carry no pitch
// (factor kB), the "cp cr" term no yaw (factor kC).  The rule composes, so second derivatives are two substitutions.
struct TrigK {
    double sr, cr, sp, cp, sy, cy, kB, kC;
};
__device__ __forceinline__ TrigK trigk(const Trig &t)
{
    TrigK d;
    d.sr = t.sr; d.cr = t.cr; d.sp = t.sp; d.cp = t.cp; d.sy = t.sy; d.cy = t.cy; d.kB = 1.0; d.kC = 1.0;
    return d;
}
__device__ __forceinline__ void zb_of(const TrigK &t, double z[3])
{
    const double spcr = t.sp * t.cr;
    z[0] = t.cy * spcr + t.kB * (t.sy * t.sr);
    z[1] = t.sy * spcr - t.kB * (t.cy * t.sr);
    z[2] = t.kC * (t.cp * t.cr);
}
__device__ __forceinline__ void zb_of(const Trig &t, double z[3])
{
    z[0] = t.cy * t.sp * t.cr + t.sy * t.sr;
    z[1] = t.sy * t.sp * t.cr - t.cy * t.sr;
    z[2] = t.cp * t.cr;
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ TrigK dtrig_lane(const TrigK &t, int sub) // derivative with respect to the lane's own angle
{
    TrigK d;
    d.sr = sub == 0 ? t.cr : t.sr; d.cr = sub == 0 ? -t.sr : t.cr;
    d.sp = sub == 1 ? t.cp : t.sp; d.cp = sub == 1 ? -t.sp : t.cp;
    d.sy = sub == 2 ? t.cy : t.sy; d.cy = sub == 2 ? -t.sy : t.cy;
    d.kB = sub == 1 ? 0.0 : t.kB; d.kC = sub == 2 ? 0.0 : t.kC;
    return d;
}
template <int L>
__device__ __forceinline__ TrigK dtrig(const TrigK &t) // derivative with respect to angle L (roll, pitch, yaw)
{
    TrigK d = t;
    if (L == 0) { d.sr = t.cr; d.cr = -t.sr; }
    if (L == 1) { d.sp = t.cp; d.cp = -t.sp; d.kB = 0.0; }
    if (L == 2) { d.sy = t.cy; d.cy = -t.sy; d.kC = 0.0; }
    return d;
}
// sines / cosines of e and e + dt rates: every lane evaluates its own angle of both, the stage's three lanes swap them
// through 12 scratch slots (layout [s1(3) c1(3) s2(3) c2(3)])
__device__ __forceinline__ void trig_shared(ldouble *sc, int sub, const double e[3], const double et[3], Trig &t1, Trig &t2)
{
    double s1, c1, s2, c2;
    sincos_bounded(pick3(sub, e[0], e[1], e[2]), &s1, &c1);
    sincos_bounded(pick3(sub, et[0], et[1], et[2]), &s2, &c2);
    sc[sub] = s1; sc[3 + sub] = c1; sc[6 + sub] = s2; sc[9 + sub] = c2;
    WSYNC();
    t1.sr = sc[0]; t1.sp = sc[1]; t1.sy = sc[2]; t1.cr = sc[3]; t1.cp = sc[4]; t1.cy = sc[5];
    t2.sr = sc[6]; t2.sp = sc[7]; t2.sy = sc[8]; t2.cr = sc[9]; t2.cp = sc[10]; t2.cy = sc[11];
    WSYNC();
}

// wave 1, three lanes per stage: Heun step, column `sub` of the compact linearisation, d, M'y (see model_phase).
// Register budget: z and y are this wave's persistent state (60 VGPRs); y waits in the record for the whole phase (the
// neighbour stage and the lane-dependent entries are read from there anyway), and every block below ends with the stores of
// what it produced, so that little more than the trig sets and the step itself is live from block to block.
template <int NP>
__device__ __forceinline__ void model_phase3(ldouble *recs, ldouble *xs, ModelState &st, int N, double &l_eq, int tw_m = 0)
{
    // (`sub` opaque per call: what is derived from it -- masks, factors, addresses -- is loop invariant, and hoisted out of the
    // interior-point loop it is spilled and reloaded on this wave's critical phase; see ROW_PICK)
    const int lane = threadIdx.x & 63, k = lane % NP, sub = opq(lane / NP);
    const bool act = k < N && sub < 3, dyn = act && k < N - 1;
    const double *z = st.z;
    l_eq = 0.0;
    ldouble *rec = recs + (act ? k : 0) * RS;
#ifdef FRP_PROFILE_W1 // (profile build) blocks of this phase: slots 18.. of the segment counters (park + step, d, linearisation + M'y)
    long long m3t_ = clock64();
#define M3_SEG(i) do { const long long tn_ = clock64(); if (lane == 0) atomicAdd((unsigned long long *)&g_prof_seg[18 + (i)], (unsigned long long)(tn_ - m3t_)); m3t_ = tn_; } while (0)
#else
#define M3_SEG(i)
#endif
    if (act && sub == 0) {
#pragma unroll
        for (int i = 0; i < NS; i++) rec[RT_Y + i] = st.y[i];
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double dx = xs[X_XINIT + i] - z[8 + i];
            xs[X_DX0 + i] = dx;
            l_eq = fmax(l_eq, fabs(dx));
        }
    }
    WSYNC();
    // ---- block 1: the step
    Trig t1, t2;
    double zb1[3], zb2[3], vt[3], xn[9], a1s = 0.0, a2s = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) xn[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) zb1[i] = zb2[i] = vt[i] = 0.0;
    t1.sr = t1.cr = t1.sp = t1.cp = t1.sy = t1.cy = 0.0; t2 = t1;
    if (dyn) {
        const double *w = z, T = z[3], *pp = z + 8, *v = z + 11, *e = z + 14;
        double et[3], a1[3];
        // (three-wave workgroups: the model wave is short of registers -- the external force comes from the workgroup scratch, where
        // the Hessian lanes read it too, instead of six registers of persistent state)
        const double fx[3] = {QW ? xs[X_FEXT + k] : st.fext[0], QW ? xs[X_FEXT + NP + k] : st.fext[1], QW ? xs[X_FEXT + 2 * NP + k] : st.fext[2]};
#pragma unroll
        for (int i = 0; i < 3; i++) et[i] = e[i] + DT * w[i];
        trig_shared(rec + RX_MTRIG, sub, e, et, t1, t2);
        zb_of(t1, zb1); zb_of(t2, zb2);
        a1s = T * (1.0 / MASS) + DRAG * dot3(zb1, v);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            a1[i] = a1s * zb1[i] - DRAG * v[i] + fx[i] - (i == 2 ? GRAV : 0.0);
            vt[i] = v[i] + DT * a1[i];
        }
        a2s = T * (1.0 / MASS) + DRAG * dot3(zb2, vt);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double a2 = a2s * zb2[i] - DRAG * vt[i] + fx[i] - (i == 2 ? GRAV : 0.0);
            xn[i] = pp[i] + 0.5 * DT * (v[i] + vt[i]);
            xn[3 + i] = v[i] + 0.5 * DT * (a1[i] + a2);
            xn[6 + i] = et[i];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    M3_SEG(0);
    // ---- block 2: d = prev(z_k) - s_{k+1}: the next stage's [w; x] comes from lane + 1 (same sub); stored by sub 0
    {
        double dmax = 0.0, dv[NS];
#pragma unroll
        for (int i = 0; i < 4; i++) dv[i] = z[i] - dpp_move<0x130>(st.z[4 + i]); // wave_shl:1
#pragma unroll
        for (int i = 0; i < 9; i++) dv[4 + i] = xn[i] - dpp_move<0x130>(st.z[8 + i]);
#pragma unroll
        for (int i = 0; i < NS; i++) dmax = fmax(dmax, fabs(dv[i]));
        if (dyn) l_eq = fmax(l_eq, dmax);
        if (act && sub == 0) { // (the last stage has no successor: d = 0; the slots carried y+ through the step phase)
#pragma unroll
            for (int i = 0; i < NS; i++) rec[R_D + i] = dyn ? dv[i] : 0.0;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    M3_SEG(1);
    // ---- block 3: column `sub` of the linearisation and the entries (sub, 4 + sub, 8 + sub, 11 + sub, 14 + sub) of
    // gm = M' y_{k+1} - [0; y_k] (3 and 7 on sub 0)
    if (act) {
        cldouble *yk = rec + RT_Y, *yn = rec + RS + RT_Y;
        double g_u = 0.0, g_T = 0.0;
        double g_w = -yk[sub], g_p = -yk[4 + sub], g_v = -yk[7 + sub], g_e = -yk[10 + sub];
        const double g_w3 = -yk[3];
        if (dyn) {
            const double *v = z + 11;
            // column `sub` of F_ve = d acc / d e and of F_vv = d acc / d v = d (zB zB' - I) at both points
            double d1[3], d2[3], fe1[3], fe2[3], fv1[3], fv2[3];
            zb_of(dtrig_lane(trigk(t1), sub), d1); zb_of(dtrig_lane(trigk(t2), sub), d2);
            const double dzv1 = dot3(d1, v), dzv2 = dot3(d2, vt);
            const double zb1j = pick3(sub, zb1[0], zb1[1], zb1[2]), zb2j = pick3(sub, zb2[0], zb2[1], zb2[2]);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                fe1[i] = a1s * d1[i] + DRAG * zb1[i] * dzv1;
                fe2[i] = a2s * d2[i] + DRAG * zb2[i] * dzv2;
                fv1[i] = DRAG * zb1[i] * zb1j - (sub == i ? DRAG : 0.0);
                fv2[i] = DRAG * zb2[i] * zb2j - (sub == i ? DRAG : 0.0);
            }
            const double q1 = dot3(zb2, fv1), q2 = dot3(zb2, fe1), q3 = dot3(zb2, zb1) * (1.0 / MASS);
            const double yp[3] = {yn[4], yn[5], yn[6]}, yv[3] = {yn[7], yn[8], yn[9]};
            const double ywj = yn[sub], yej = yn[10 + sub], ypj = yn[4 + sub];
            ldouble *lc = rec + R_LIN + sub;
            g_u = ywj + DT * yej;
            g_p += ypj;
            g_e += yej;
            g_T = yn[3];
            double bptj = 0.0, bvtj = 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double sv = fv2[i] + DT * DRAG * (zb2[i] * q1 - fv1[i]); // (F_vv2 (I + dt F_vv1))[i][sub]
                const double se = fe2[i] + DT * DRAG * (zb2[i] * q2 - fe1[i]); // (F_ve2 + dt F_vv2 F_ve1)[i][sub]
                const double apv = (sub == i ? DT : 0.0) + 0.5 * DT * DT * fv1[i];
                const double ape = 0.5 * DT * DT * fe1[i];
                const double avv = (sub == i ? 1.0 : 0.0) + 0.5 * DT * (fv1[i] + sv);
                const double ave = 0.5 * DT * (fe1[i] + se);
                const double bvw = 0.5 * DT * DT * fe2[i];
                lc[0 + 3 * i] = apv; lc[9 + 3 * i] = ape; lc[18 + 3 * i] = avv; lc[27 + 3 * i] = ave; lc[42 + 3 * i] = bvw;
                const double g1 = zb1[i] * (1.0 / MASS), g2 = zb2[i] * (1.0 / MASS);
                const double bpt = 0.5 * DT * DT * g1, bvt = 0.5 * DT * (g1 + g2 + DT * DRAG * (zb2[i] * q3 - g1));
                bptj = sub == i ? bpt : bptj; bvtj = sub == i ? bvt : bvtj;
                g_u += bvw * yv[i];
                g_v += apv * yp[i] + avv * yv[i];
                g_e += ape * yp[i] + ave * yv[i];
                g_T += bpt * yp[i] + bvt * yv[i];
            }
            rec[R_LIN + 36 + sub] = bptj;
            rec[R_LIN + 39 + sub] = bvtj;
        }
        rec[R_PHIC + sub] = g_u; rec[R_PHIC + 4 + sub] = g_w; rec[R_PHIC + 8 + sub] = g_p;
        rec[R_PHIC + 11 + sub] = g_v; rec[R_PHIC + 14 + sub] = g_e;
        if (sub == 0) { rec[R_PHIC + 3] = g_T; rec[R_PHIC + 7] = g_w3; }
#pragma unroll
        for (int i = 0; i < NS; i++) st.y[i] = yk[i];
    }
    M3_SEG(2);
    // ---- block 4 (twisted solve): the stages of the first half keep the INVERTED transition [u; x]_k = T~ [w+; x+] + t~ in place of the
    // linearisation (layout: ma_src).  A = [I Apv Ape; 0 Avv Ave; 0 0 I] inverts in closed form around the 3 x 3 block Avv; lane `sub`
    // forms row `sub` of every block:  A~vv = Avv^-1,  A~ve = -Avv^-1 Ave,  A~pv = -Apv Avv^-1,  A~pe = -Ape - A~pv Ave,
    // B~ = -A~ B,  t~ = [-d_w; A~ (B d_w - d_x)].
    if (tw_m) {
        // Three passes, each: read, fence, store -- the three lanes of a stage read each other's slots, and what is live between the
        // passes (beside this wave's 60 registers of persistent state) is one row of each inverse block.
        WSYNC(); // the three lanes' columns of the linearisation are in the record; the y slots (which share two of B~'s) are read
        const bool inv = act && k < tw_m;
        ldouble *L = rec + R_LIN;
        const int s3 = 3 * sub;
        double vi[3], tpv[3], tpe[3], tve[3];
        // pass 1: Avv^-1 (adjugate), rows `sub` of A~vv and A~pv
        if (inv) {
            double v[9];
#pragma unroll
            for (int i = 0; i < 9; i++) v[i] = L[18 + i];
            const double ap0 = L[s3], ap1 = L[s3 + 1], ap2 = L[s3 + 2];
            const double c00 = v[4] * v[8] - v[5] * v[7], c01 = v[5] * v[6] - v[3] * v[8], c02 = v[3] * v[7] - v[4] * v[6];
            const double id = fast_rcp(v[0] * c00 + v[1] * c01 + v[2] * c02);
            const double V0[3] = {c00 * id, (v[2] * v[7] - v[1] * v[8]) * id, (v[1] * v[5] - v[2] * v[4]) * id};
            const double V1[3] = {c01 * id, (v[0] * v[8] - v[2] * v[6]) * id, (v[2] * v[3] - v[0] * v[5]) * id};
            const double V2[3] = {c02 * id, (v[1] * v[6] - v[0] * v[7]) * id, (v[0] * v[4] - v[1] * v[3]) * id};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                vi[j] = pick3(sub, V0[j], V1[j], V2[j]);
                tpv[j] = -(ap0 * V0[j] + ap1 * V1[j] + ap2 * V2[j]);
            }
        }
        WSYNC();
        if (inv) {
#pragma unroll
            for (int j = 0; j < 3; j++) { L[s3 + j] = tpv[j]; L[18 + s3 + j] = vi[j]; }
        }
        // pass 2: rows `sub` of A~ve = -Avv^-1 Ave and A~pe = -Ape - A~pv Ave
        if (inv) {
            double E[9];
#pragma unroll
            for (int i = 0; i < 9; i++) E[i] = L[27 + i];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                tpe[j] = -L[9 + s3 + j] - (tpv[0] * E[j] + tpv[1] * E[3 + j] + tpv[2] * E[6 + j]);
                tve[j] = -(vi[0] * E[j] + vi[1] * E[3 + j] + vi[2] * E[6 + j]);
            }
        }
        WSYNC();
        if (inv) {
#pragma unroll
            for (int j = 0; j < 3; j++) { L[9 + s3 + j] = tpe[j]; L[27 + s3 + j] = tve[j]; }
        }
        // pass 3: rows `sub` of B~ = -A~ B (B = [0 BpT; Bvw BvT; dt I 0]) and of t~ = A~ (B d_w - d_x); t~_u = -d_w
        double o_bp[4], o_bv[4], o_t[3], o_tu[4];
        if (inv) {
            double Bv[12], dw[4], rv[3], re[3];
#pragma unroll
            for (int l = 0; l < 3; l++) {
                Bv[4 * l + 0] = L[42 + 3 * l]; Bv[4 * l + 1] = L[43 + 3 * l]; Bv[4 * l + 2] = L[44 + 3 * l]; Bv[4 * l + 3] = L[39 + l];
            }
            const double bpt = L[36 + sub];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                o_bv[c] = -(vi[0] * Bv[c] + vi[1] * Bv[4 + c] + vi[2] * Bv[8 + c] + (c < 3 ? tve[c] * DT : 0.0));
                o_bp[c] = -((c == 3 ? bpt : 0.0) + tpv[0] * Bv[c] + tpv[1] * Bv[4 + c] + tpv[2] * Bv[8 + c] + (c < 3 ? tpe[c] * DT : 0.0));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { dw[i] = rec[R_D + i]; o_tu[i] = -dw[i]; }
#pragma unroll
            for (int l = 0; l < 3; l++) {
                rv[l] = Bv[4 * l] * dw[0] + Bv[4 * l + 1] * dw[1] + Bv[4 * l + 2] * dw[2] + Bv[4 * l + 3] * dw[3] - rec[R_D + 7 + l];
                re[l] = DT * dw[l] - rec[R_D + 10 + l];
            }
            const double rp = bpt * dw[3] - rec[R_D + 4 + sub];
            o_t[0] = rp + tpv[0] * rv[0] + tpv[1] * rv[1] + tpv[2] * rv[2] + tpe[0] * re[0] + tpe[1] * re[1] + tpe[2] * re[2];
            o_t[1] = vi[0] * rv[0] + vi[1] * rv[1] + vi[2] * rv[2] + tve[0] * re[0] + tve[1] * re[1] + tve[2] * re[2];
            o_t[2] = pick3(sub, re[0], re[1], re[2]);
        }
        WSYNC();
        if (inv) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                rec[R_LIN + 36 + 4 * sub + c] = o_bp[c];
                rec[pick3i(sub, lbv_slot(c), lbv_slot(4 + c), lbv_slot(8 + c))] = o_bv[c];
            }
            rec[R_D + 4 + sub] = o_t[0]; rec[R_D + 7 + sub] = o_t[1]; rec[R_D + 10 + sub] = o_t[2];
            if (sub == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) rec[R_D + i] = o_tu[i];
            }
        }
    }
}

// wave 0, three lanes per stage: exact Hessian of y_{k+1}' c(z_k), rows / columns of angle `sub` (see rk2_hessian_core)
static_assert(hd_pack(0, 0) == 0 && hd_pack(1, 1) == 10 && hd_pack(2, 2) == 19 && hd_pack(3, 7) == 27 && hd_pack(4, 7) == 30 &&
              hd_pack(5, 7) == 33 && hd_pack(6, 7) == 36 && hd_pack(7, 7) == 39 && hd_pack(8, 8) == 42 && hd_pack(9, 9) == 44, "packed Hessian rows");
__device__ __forceinline__ void hessian_phase3(ldouble *rec, const HessState &hs, int sub, bool dyn, int hess)
{
    sub = opq(sub); // (see model_phase3)
    if (dyn && hess) {
        const double *w = hs.u, T = hs.u[3], *v = hs.ve, *e = hs.ve + 3, *yp = hs.y6, *yv = hs.y6 + 3;
        double et[3];
#pragma unroll
        for (int i = 0; i < 3; i++) et[i] = e[i] + DT * w[i];
        ldouble *sc = rec + RX_HTRIG;
        Trig t1, t2;
        trig_shared(sc, sub, e, et, t1, t2);
        double zb1[3], zb2[3], vt[3], beta[3], gam1[3];
        zb_of(t1, zb1); zb_of(t2, zb2);
        const double a1s = T * (1.0 / MASS) + DRAG * dot3(zb1, v);
#pragma unroll
        for (int i = 0; i < 3; i++) vt[i] = v[i] + DT * (a1s * zb1[i] - DRAG * v[i] + hs.fext[i] - (i == 2 ? GRAV : 0.0));
        const double a2s = T * (1.0 / MASS) + DRAG * dot3(zb2, vt);
#pragma unroll
        for (int i = 0; i < 3; i++) beta[i] = 0.5 * DT * yv[i];
        const double s2 = dot3(beta, zb2);
#pragma unroll
        for (int i = 0; i < 3; i++) gam1[i] = (0.5 * DT * DT * yp[i] + 0.5 * DT * yv[i]) + DT * (DRAG * zb2[i] * s2 - DRAG * beta[i]);
        const double s1 = dot3(gam1, zb1);
        // first derivatives of zB: all three directions at the first point (the Jacobian of acc1 enters K in full), the own one at the second
        double D1a[3][3], d1[3], d2[3], dzv1[3];
        zb_of(dtrig<0>(trigk(t1)), D1a[0]); zb_of(dtrig<1>(trigk(t1)), D1a[1]); zb_of(dtrig<2>(trigk(t1)), D1a[2]);
#pragma unroll
        for (int c = 0; c < 3; c++) d1[c] = pick3(sub, D1a[0][c], D1a[1][c], D1a[2][c]);
#pragma unroll
        for (int l = 0; l < 3; l++) dzv1[l] = dot3(D1a[l], v);
        const TrigK t1j = dtrig_lane(trigk(t1), sub), t2j = dtrig_lane(trigk(t2), sub);
        zb_of(t2j, d2);
        const double sj1 = dot3(gam1, d1), aj1 = DRAG * pick3(sub, dzv1[0], dzv1[1], dzv1[2]);
        const double sj2 = dot3(beta, d2), aj2 = DRAG * dot3(d2, vt);
        sc[sub] = sj1; sc[3 + sub] = aj1; sc[6 + sub] = sj2; sc[9 + sub] = aj2;
        WSYNC();
        double sj1v[3], aj1v[3], sj2v[3], aj2v[3];
#pragma unroll
        for (int l = 0; l < 3; l++) { sj1v[l] = sc[l]; aj1v[l] = sc[3 + l]; sj2v[l] = sc[6 + l]; aj2v[l] = sc[9 + l]; }
        WSYNC();
        // row `sub` of the two (e, e) blocks
        double hee1[3], hee2[3];
        {
            double dd[3];
            zb_of(dtrig<0>(t1j), dd); hee1[0] = DRAG * dot3(dd, v) * s1 + aj1 * sj1v[0] + aj1v[0] * sj1 + a1s * dot3(gam1, dd);
            zb_of(dtrig<1>(t1j), dd); hee1[1] = DRAG * dot3(dd, v) * s1 + aj1 * sj1v[1] + aj1v[1] * sj1 + a1s * dot3(gam1, dd);
            zb_of(dtrig<2>(t1j), dd); hee1[2] = DRAG * dot3(dd, v) * s1 + aj1 * sj1v[2] + aj1v[2] * sj1 + a1s * dot3(gam1, dd);
            zb_of(dtrig<0>(t2j), dd); hee2[0] = DRAG * dot3(dd, vt) * s2 + aj2 * sj2v[0] + aj2v[0] * sj2 + a2s * dot3(beta, dd);
            zb_of(dtrig<1>(t2j), dd); hee2[1] = DRAG * dot3(dd, vt) * s2 + aj2 * sj2v[1] + aj2v[1] * sj2 + a2s * dot3(beta, dd);
            zb_of(dtrig<2>(t2j), dd); hee2[2] = DRAG * dot3(dd, vt) * s2 + aj2 * sj2v[2] + aj2v[2] * sj2 + a2s * dot3(beta, dd);
        }
        const double hTe1 = sj1 * (1.0 / MASS), hTe2 = sj2 * (1.0 / MASS);
        double hve1[3], hve2[3], Kv[3], Ke[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            hve1[i] = DRAG * (d1[i] * s1 + zb1[i] * sj1);
            hve2[i] = DRAG * (d2[i] * s2 + zb2[i] * sj2);
        }
        // K = J1' H2ve (+ h2Te on the T row), column `sub`: J1 = [F_vv1 F_ve1 g_T1] of the first acceleration
        const double zh = dot3(zb1, hve2);
        const double KT = hTe2 + DT * (1.0 / MASS) * zh;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Kv[i] = hve2[i] + DT * DRAG * (zb1[i] * zh - hve2[i]);
            Ke[i] = DT * (a1s * dot3(D1a[i], hve2) + DRAG * zh * dzv1[i]);
        }
        sc[3 * sub] = Ke[0]; sc[3 * sub + 1] = Ke[1]; sc[3 * sub + 2] = Ke[2];
        WSYNC();
        double Ker[3]; // Ke[sub][l]: entry `sub` of the columns of the three lanes
#pragma unroll
        for (int l = 0; l < 3; l++) Ker[l] = sc[3 * l + sub];
        WSYNC();
        // packed upper triangle over (w 0..2, T 3, v 4..6, e 7..9): row a < 3 starts at {0, 10, 19}[a] for column a, row 3 + i at
        // 27 + 3 i for column 7, row 7 + a at {39, 42, 44}[a] for column 7 + a
        ldouble *hd = rec + R_HD;
        ldouble *hw = hd + (sub == 0 ? 0 : (sub == 1 ? 9 : 17));  // + column index
        ldouble *he = hd + (sub == 0 ? 39 : (sub == 1 ? 41 : 42)); // + l
        ldouble *dump = rec + R_DUMP;
#pragma unroll
        for (int l = 0; l < 3; l++) {
            ldouble *pw = l >= sub ? hw + l : dump, *pe = l >= sub ? he + l : dump;
            *pw = DT * DT * hee2[l];                             // (w_sub, w_l)
            *pe = hee1[l] + Ker[l] + Ke[l] + hee2[l];            // (e_sub, e_l)
            hw[7 + l] = DT * (Ke[l] + hee2[l]);                  // (w_sub, e_l)
            hw[4 + l] = DT * Kv[l];                              // (w_sub, v_l)
            hd[30 + 3 * l + sub] = hve1[l] + Kv[l];              // (v_l, e_sub)
        }
        hw[3] = DT * KT;                                         // (w_sub, T)
        hd[27 + sub] = hTe1 + KT;                                // (T, e_sub)
    } else if (sub < 3) {
        // the Hessian slots are overwritten by P every iteration: stages without a dynamics Hessian clear them again
#pragma unroll
        for (int i = 0; i < REC_HD_SIZE / 3; i++) rec[R_HD + 15 * sub + i] = 0.0;
    }
}

// Live corridor rows of a stage when the caller gives no face counts: trailing all-zero rows are padding
// (forces_normal.cpp:127-135).  The loop is lane-divergent (face counts differ per stage), so it lives in a function of its
// own that must compile WITHOUT scratch: the gfx950 backend can place a VGPR spill at the exit of a lane-divergent loop,
// where it executes with an empty EXEC mask and saves nothing (tests/test_capi_cpu.py checks the scratch size).
__device__ __noinline__ int count_live_faces(const double *pk, int M)
{
    int nf = M;
    while (nf > 0) {
        const double *r = pk + NPRE + 3 * (nf - 1);
        if (r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0 && pk[NPRE + 3 * M + nf - 1] >= -HU) nf--;
        else break;
    }
    return nf;
}

// ================================================================== the solve, one role per wave
struct Shared {
    ldouble *recs, *xs, *tw;
    Ctl *ctl;
    int cukey; // this workgroup's CU in the per-CU words of the workspace (KernelArgs::cu_slots), or -1
    int late;  // (Q4, head start) 1: another workgroup was on this CU first -- it may be about to mark the CU: the first claim waits a moment
    int rsimd; // (Q4) the SIMD this workgroup's Riccati wave claimed (0..3: distinct for the four workgroups of a CU), or -1
};

// The next problem of this workgroup: queue position from the device counter, mapped through the launch order (longest
// expected solve first, see order_keys_kernel) by the one lane that claims it; a.B = the queue is exhausted.
// Q4 variants: a solve that reaches iteration a.iso_it has marked its CU (bits 8.. of the CU's word count such solves); the other
// workgroups of that CU finish the solves they have and wait here -- a problem alone on a CU iterates a quarter faster, and the launch
// ends with its longest solve.  (The marked solve itself never waits: it takes its mark back before it claims.)
__device__ __forceinline__ int claim_next(const KernelArgs &a, int cukey = -1)
{
    if (QW && cukey >= 0 && a.iso_it > 0) {
        while (__hip_atomic_load(a.cu_slots + cukey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 256) __builtin_amdgcn_s_sleep(64);
    }
    const int p = atomicAdd(a.counter, 1);
    if (p >= a.B) {
        // (head-start problems nobody took -- fewer first arrivals than KernelArgs::head_start -- are ordinary problems for whoever comes here)
        if (QW && a.head_start > 0) {
            const int h = atomicAdd(a.counter + 2, 1);
            if (h < a.head_start) return a.order ? a.order[h] : h;
        }
        return a.B;
    }
    return a.order ? a.order[p] : p;
}

__device__ __forceinline__ void publish(ldouble *xs, int wave, int lane, int slot, double v)
{
    if (lane == 0) xs[X_RED + wave * 16 + slot] = v;
}
__device__ __forceinline__ double red(cldouble *xs, int wave, int slot) { return xs[X_RED + wave * 16 + slot]; }

// TW: the twisted solve (a.twist stages eliminated forward by the model wave while the Riccati wave runs the rest backward); a kernel
// variant of its own, so that the plain solve carries none of it
template <int NP, int FL, bool FREG, int ROLE, bool TW>
__device__ __forceinline__ void solve_one(const KernelArgs &a, const int b, const Shared &sh)
{
    constexpr int H = RowMap<NP>::H, R = RowMap<NP>::R;
    constexpr int wave = ROLE; // == threadIdx.x >> 6: every role is compiled on its own, so only ITS state occupies registers
    // Who does what.  Plain: 0 Riccati, 1 model, 2 bounds, 3 faces (+ some bound rounds).  Q4 (three waves): 0 Riccati, 1 model + faces
    // (+ the bound rounds [RQ, R), none by default), 2 the bound rounds [0, RQ).  Measured (profile build, one problem alone, cycles of
    // the three element-wise phases' long pole): faces + 5 bound rounds on one wave 13.6 / 7.8 / 7.3 k (134 registers of persistent state:
    // ~110 scratch accesses per iteration), faces on the Riccati wave 14.8 / 4.0 / 7.7 k (its state does not fit the callee-saved registers
    // of the sweeps, and that wave runs at low priority), against 7.1 / 3.2 / 2.2 k of the four-wave split.
    // The element-wise partial results are published per wave in X_RED and combined by everybody in the fixed order (WB, WF);
    // WEQ = the row the model wave publishes the equality norm in.
#ifndef FRP_Q4_FWAVE // Q4: the wave that owns the corridor rows (0 = the Riccati wave, 1 = the model wave, 2 = the wave of the bound rounds)
#define FRP_Q4_FWAVE 1
#endif
#ifndef FRP_Q4_RM // Q4: bound rounds on the model wave (the LAST ones: rows 15, 16 carry no rate coupling and need two cost parameters)
#define FRP_Q4_RM 0
#endif
    constexpr int WEQ = QW ? 3 : 1;
    constexpr bool IS_M = wave == 1, IS_F = wave == (QW ? FRP_Q4_FWAVE : 3);   // model; corridor rows
    // CONTRIB(w): wave w owns bound rounds and / or corridor rows and publishes element-wise partial results
    auto contrib = [](int w) constexpr { return QW ? (w == 2 || w == FRP_Q4_FWAVE || (w == 1 && FRP_Q4_RM > 0)) : (w == 2 || w == 3); };
    constexpr bool IS_B = contrib(wave);
    constexpr bool IS_BO = IS_B && !IS_F;                                      // ... bound rounds only
    static_assert(!QP || ((NP == 20 || QS) && !TW), "P in global memory: the y+ rows are dealt over the three lanes of a stage (Q4) or formed by the lane of the stage (Q30)");
    static_assert(!QS || (!QW && H == 2 && !TW), "Q30: four-wave workgroups, two lanes per stage in the element-wise phases, the lane == stage model phase");
    static_assert(!QW || (NP == 20 && !TW && wave < 3), "Q4: the three-lanes-per-stage model phase, plain solve");
    const int lane = threadIdx.x & 63;
    const int N = a.N, M = a.M, MF = a.MF, np = NPRE + 4 * M;
    ldouble *recs = sh.recs, *xs = sh.xs;
    // combined by every wave in the fixed order 2, 3 (plain) / 2, 1, 0 (Q4, the waves that contribute)
    auto rmax = [&](int slot) {
        double v = red(xs, 2, slot);
        if constexpr (contrib(3)) v = fmax(v, red(xs, 3, slot));
        if constexpr (contrib(1)) v = fmax(v, red(xs, 1, slot));
        if constexpr (contrib(0)) v = fmax(v, red(xs, 0, slot));
        return v;
    };
    auto rmin = [&](int slot) {
        double v = red(xs, 2, slot);
        if constexpr (contrib(3)) v = fmin(v, red(xs, 3, slot));
        if constexpr (contrib(1)) v = fmin(v, red(xs, 1, slot));
        if constexpr (contrib(0)) v = fmin(v, red(xs, 0, slot));
        return v;
    };
    auto rsum = [&](int slot) {
        double v = red(xs, 2, slot);
        if constexpr (contrib(3)) v += red(xs, 3, slot);
        if constexpr (contrib(1)) v += red(xs, 1, slot);
        if constexpr (contrib(0)) v += red(xs, 0, slot);
        return v;
    };
    double *pws = nullptr; // Q4: this workgroup's NP blocks of packed P in global memory
    if constexpr (QP) pws = uni(a.pws + (size_t)blockIdx.x * (NP * PG));
    const int tw_m = TW ? uni(a.twist) : 0; // (validated by the launcher: 2 <= tw_m <= N - 2)
    ldouble *tw = sh.tw;
    // three lanes per stage (H == 3: NP = 20) also on the Riccati wave's Hessian and on the model wave (model_phase3, hessian_phase3)
    constexpr bool H3 = (H == 3);
    const int k = (wave == 1 && !H3) ? lane : lane % NP, half = lane / NP; // stage of this lane; row / face group
    const bool kact = k < N && ((wave == 1 && !H3) || half < H);
    const bool hact = kact && (H3 || half == 0); // lanes that carry a copy of the stage's model state (waves 0, 1)
    const bool own0 = kact && half == 0;         // one lane per stage: writes that must happen once
    const double *pk = a.params + ((size_t)b * N + (k < N ? k : 0)) * np;
    const int hess = a.hessian ? 1 : 0;
    const int model = a.models ? a.models[b] : a.model; // normal / final objective of THIS problem (switch_to_final, nmpc_solver.cpp:381)

    PROF_T0();
    // ---------------------------------------------------------------- per-wave state
    ModelState ms;                         // wave 1
    double bz[R], bzp[R], bsl[R], bsu[R], bll[R], blu[R], bcl[R], bcu[R], pc[NPRE]; // wave 2
    double fa0[FL], fa1[FL], fa2[FL], fbb[FL], fs[FL], fl_[FL], fcr[FL], fpos[3];    // wave 3 (fa*, fbb only when FREG)
    int nfk = 0;
#pragma unroll
    for (int i = 0; i < NZ; i++) ms.z[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NS; i++) ms.y[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NPRE; i++) pc[i] = 0.0;
    ms.fext[0] = ms.fext[1] = ms.fext[2] = 0.0;
    // wave 0: what its Hessian lanes carry from the step phase to the next evaluation -- the Newton step of (rates, T), (v, e) and
    // y+ (p, v rows) of the next stage; the values before the step come from the model wave through the record (RT_HZ, RT_HY),
    // so that nothing of the Hessian's inputs is live across the sweeps
#ifndef FRP_QP_YWAVE // QP: the wave that forms y+ (0 = the Riccati wave, fetching S_xx in the step phase; 2 = the bounds wave, prefetching behind barrier D)
#define FRP_QP_YWAVE 0
#endif
    constexpr int WY = FRP_QP_YWAVE;
    double sx[16];       // QP: this lane's block of S_xx
#pragma unroll
    for (int i = 0; i < 16; i++) sx[i] = 0.0;
    auto fetch_sx = [&]() {
        typedef double gd2 __attribute__((ext_vector_type(2)));
        const int aq = opq(half) < 3 ? opq(half) : 2;
        // (a GLOBAL pointer by type: through the generic one these were FLAT loads -- the one trip to memory on the path of the step phase took the flat route and counted on the LDS counter too)
        const __attribute__((address_space(1))) gd2 *src = (const __attribute__((address_space(1))) gd2 *)((const char *)pws + ((unsigned)opq(k) * (PG * 8) + (unsigned)aq * 128u));
#pragma unroll
        for (int q = 0; q < 8; q++) { const gd2 v = src[q]; sx[2 * q] = v.x; sx[2 * q + 1] = v.y; }
    };
    double hyp[6], hap = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) hyp[i] = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) { bz[r] = bzp[r] = 0.0; bsl[r] = bsu[r] = bll[r] = blu[r] = 1.0; bcl[r] = bcu[r] = 0.0; }
#pragma unroll
    for (int t = 0; t < FL; t++) { fa0[t] = fa1[t] = fa2[t] = fbb[t] = 0.0; fs[t] = fl_[t] = 1.0; fcr[t] = 0.0; }
    fpos[0] = fpos[1] = fpos[2] = 0.0;
    // (corridor rows on the model wave: the position is z[8..10] of that wave's own state -- updated by the same operations -- not a copy)
    // (measured and dropped: the rows' slacks and multipliers in the record between the phases -- the model wave's trig hand-over slots,
    // the trig sets through ds_bpermute instead -- frees 8 registers and changes nothing: 0.855 vs 0.853 ms; what the allocator keeps
    // in scratch on this wave is two row constants, the second-order terms between the barriers D and F and three integers)
    constexpr bool FPZ = IS_M && IS_F;
    double *const fpq = FPZ ? ms.z + 8 : fpos;
    // Three-wave workgroups, rows in registers: the merged model + corridor wave is a few registers short in its model phase, and what the allocator sent to
    // scratch for it -- one constant of every row all iteration long, the rows' second-order terms from barrier D to the commit -- came back one dependent
    // round trip to memory at a time: six in the evaluation phase, two in the affine phase, four each in the step phase and the commit (the disassembly:
    // scratch_load, s_waitcnt vmcnt(0), use, next scratch_load), on the wave that is the long pole of every one of them.  Those values live in LDS instead:
    // four slots per lane at the record's end (lane (k, half): slots 4 half .. 4 half + 3 of stage k -- b of its two rows, then their second-order terms),
    // written and read by their own lane only.
#if defined(FRP_QL) && !defined(FRP_QS) && FRP_Q4_PARK
    constexpr bool PARK = QW && FREG && IS_F && FL <= 2;
    constexpr int R_PRK = RQ_MTRIG;
#else
    constexpr bool PARK = false;
    constexpr int R_PRK = 0;
#endif
    ldouble *const prk = recs + (k < NP ? k : 0) * RS + R_PRK + 4 * (half < 3 ? half : 0);

    // face t of this lane is row j = t * H + half of its stage; its constants come from registers or from the parameters
    // (re-reading variants: two per-lane base pointers -- row `half` of A and of b -- made opaque once per phase by face_bases(),
    // every row of the lane then an immediate offset from them.  Left to itself the compiler keeps a 64-bit address per
    // row for the whole solve: 40 registers at ten rows per lane, i.e. 40 spills)
    // (the row group clamped to the last one: the lanes 3 NP .. 63 of the three-lanes-per-stage mapping are inactive, but the chunked
    // fetch below is unconditional -- with the unclamped group their last row would lie one row behind the stage's block)
    const int halfc = half < H ? half : H - 1;
    // (pointers into GLOBAL memory by type: behind the opaque asm of face_bases() a generic pointer's loads are FLAT instructions, which count on the LDS counter as well --
    // every wait for a row constant then drains the wave's LDS reads, and the other way round)
#ifdef FRP_ROWS_FLAT // (experiment knob: the generic pointers of rounds 2-5)
    typedef const double gcdouble;
#else
    typedef const __attribute__((address_space(1))) double gcdouble;
#endif
    gcdouble *pkA = (gcdouble *)(pk + NPRE + 3 * halfc), *pkB = (gcdouble *)(pk + NPRE + 3 * M + halfc);
    auto face_bases = [&]() {
        if constexpr (!FREG) {
            pkA = (gcdouble *)(pk + NPRE + 3 * halfc); pkB = (gcdouble *)(pk + NPRE + 3 * M + halfc);
            asm volatile("" : "+v"(pkA), "+v"(pkB));
        }
    };
    auto face_consts = [&](int t, double &a0, double &a1, double &a2, double &bb) {
        if (FREG) { a0 = fa0[t]; a1 = fa1[t]; a2 = fa2[t]; bb = PARK ? prk[t] : fbb[t]; }
        else { a0 = pkA[3 * H * t]; a1 = pkA[3 * H * t + 1]; a2 = pkA[3 * H * t + 2]; bb = pkB[H * t] + HU; }
    };
    // body(t, a0, a1, a2, bb) for the live rows of this lane.  Re-reading variants: a load inside `if (row is live)` cannot be hoisted
    // over the branch, so the rows of a lane were ten dependent trips to the L2 per phase (measured: the faces wave 14.6 k / 10.8 k /
    // 8.9 k cycles in the evaluation / affine / step phases at 30 rows per stage against 6.1 / 3.3 / 2.6 k at 6); the constants of
    // FCH rows are now fetched together, unconditionally (every row index of the lane is inside the stage's block when M covers
    // FL * H rows -- else the row-by-row form), and only their use is predicated.
#ifndef FRP_FCH // experiment knob: rows per fetch
#define FRP_FCH 5
#endif
    constexpr int FCH = (!FREG && FL % FRP_FCH == 0) ? FRP_FCH : 1;
    const bool rows_in_block = M >= FL * H;
    auto for_faces = [&](auto body) __attribute__((always_inline)) {
        if constexpr (FREG || FCH == 1) {
#pragma unroll
            for (int t = 0; t < FL; t++) {
                FRP_RSB();
                if (t * H + half < nfk) {
                    double a0, a1, a2, bb;
                    face_consts(t, a0, a1, a2, bb);
                    body(t, a0, a1, a2, bb);
                }
            }
        } else if (rows_in_block) {
#pragma unroll
            for (int t0 = 0; t0 < FL; t0 += FCH) {
                double c0[FCH], c1[FCH], c2[FCH], cb[FCH];
#pragma unroll
                for (int q = 0; q < FCH; q++) {
                    const int t = t0 + q;
                    c0[q] = pkA[3 * H * t]; c1[q] = pkA[3 * H * t + 1]; c2[q] = pkA[3 * H * t + 2]; cb[q] = pkB[H * t];
                }
#pragma unroll
                for (int q = 0; q < FCH; q++) {
                    const int t = t0 + q;
                    if (t * H + half < nfk) body(t, c0[q], c1[q], c2[q], cb[q] + HU);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < FL; t++) {
                if (t * H + half < nfk) {
                    double a0, a1, a2, bb;
                    face_consts(t, a0, a1, a2, bb);
                    body(t, a0, a1, a2, bb);
                }
            }
        }
    };

    // The 17 bound pairs of a stage are shared by waves 2 and 3: rounds [RB0, RB1) of the row mapping each (wave 3 also
    // owns the corridor rows), so that neither is the long pole of the element-wise phases.
    // (measured: a corridor row costs about as much as 1.5 bound pairs because of its cross-lane sums)
#ifndef FRP_RSPLIT_ADJ // experiment knob: bound rounds moved from the faces wave to the bounds wave
#define FRP_RSPLIT_ADJ 0
#endif
    constexpr int RSPLIT0 = R - (FL <= 2 ? R / 3 : (FL <= 5 ? R / 6 : 0));
#ifndef FRP_RSPLIT_ADJ32 // (same, for the N <= 32 variants only)
#define FRP_RSPLIT_ADJ32 0
#endif
    constexpr int RSPLIT_ADJ_ = FRP_RSPLIT_ADJ + ((NP == 32 || NP == 30) ? FRP_RSPLIT_ADJ32 : 0);
    constexpr int RSPLIT = RSPLIT0 + RSPLIT_ADJ_ <= R ? RSPLIT0 + RSPLIT_ADJ_ : R;
    constexpr int RQ = R - FRP_Q4_RM;
    constexpr int RB0 = QW ? (wave == 1 ? RQ : 0) : (IS_F ? RSPLIT : 0), RB1 = QW ? (wave == 1 ? R : RQ) : (IS_F ? R : RSPLIT);
    // evaluation: residuals, barrier Hessian / predictor rhs of this wave's bound rows -> record; norms
    auto bounds_eval = [&](double &l_in, double &l_rc, double &l_gap) {
        if (!kact) return;
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        ldouble *rec = recs + k * RS;
        const int halfp = opq(half);
#pragma unroll
        for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
            const int ib = r * H, i = ib + half;
            if (i >= NZ) continue;
            const double hd = ROW_PICK(cq.hd(i));
            const double qi = ROW_PICK(cq.q(i));
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
            const double zi = bz[r];
            double cg = hd * zi + qi; // cost gradient
            if (ib < 8) cg += (i < 8 ? cq.hc() : 0.0) * bzp[r];
            const double sl = bsl[r], su = bsu[r], ll = bll[r], lu = blu[r];
            const double vl = lb - zi, vu = zi - ub;
            const double rl = vl + sl, ru = vu + su;
            l_in = fmax(l_in, fmax(fmax(vl, vu), fmax(fabs(rl), fabs(ru))));
            l_rc = fmax(l_rc, fmax(sl * ll, su * lu));
            l_gap += sl * ll + su * lu;
            const double sgl = ll * fast_rcp(sl), sgu = lu * fast_rcp(su);
            rec[R_PHIB + i] = cg + lu - ll;          // cost and bound part of the stationarity residual
            rec[R_PHID + i] = hd + sgl + sgu;
            rec[R_PHI + i] = cg + sgu * ru - sgl * rl;
        }
    };
    // affine step: lengths, second-order terms (kept in registers), corrector rhs of the bound rows -> record
    auto bounds_affine = [&](double &m_p, double &m_d, double &s_sdl, double &s_lds, double &s_dsdl) {
        if (!kact) return;
        const CostQ cq = make_cost(pc, stage_class(k, N), model);
        ldouble *rec = recs + k * RS;
        const int halfp = opq(half);
#pragma unroll
        for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
            const int ib = r * H, i = ib + half;
            if (i >= NZ) continue;
            const double hd = ROW_PICK(cq.hd(i));
            const double qi = ROW_PICK(cq.q(i));
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
            const double zi = bz[r], dzi = rec[R_DZ + i];
            double pb = hd * zi + qi; // cost gradient
            if (ib < 8) pb += (i < 8 ? cq.hc() : 0.0) * bzp[r];
            // one constraint of the affine step (smu = 0, corr = 0): t1 = (l r_in - corr)/s, sinv = 1/s
            auto cstep = [&](double s, double l, double gdz, double viol, double &cr, double &t1, double &sinv) {
                const double u = fast_rcp(s * l);
                sinv = u * l;
                const double linv = u * s;
                const double rin = viol + s;
                const double ds = -rin - gdz;
                const double dl = -l * (1.0 + ds * sinv); // (-(s l) - l ds) / s
                m_p = fmax(m_p, -ds * sinv);
                m_d = fmax(m_d, -dl * linv);
                s_sdl += s * dl; s_lds += l * ds;
                cr = ds * dl;
                s_dsdl += cr;
                t1 = (l * rin - cr) * sinv;
            };
            double tl, tu, sil, siu;
            cstep(bsl[r], bll[r], -dzi, lb - zi, bcl[r], tl, sil);
            cstep(bsu[r], blu[r], dzi, zi - ub, bcu[r], tu, siu);
            rec[R_PHIB + i] = pb + tu - tl;
            rec[R_PHIC + i] = siu - sil;
        }
    };

    // ---------------------------------------------------------------- init
    double smin = 1e300;
    int bad_param = 0, mcount = 0;
    auto bounds_init = [&]() {
        if (!kact) return;
        const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
        pc[0] = pk[0]; pc[1] = pk[1]; pc[2] = pk[2]; pc[6] = pk[6]; pc[7] = pk[7]; pc[8] = pk[8]; pc[9] = pk[9];
        const int halfp = opq(half);
#pragma unroll
        for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
            const int ib = r * H, i = ib + half;
            if (i >= NZ) continue;
            const double lb = ROW_PICK(lower_bound(i));
            const double ub = ROW_PICK(upper_bound(i));
            bz[r] = z0[i];
            if (ib < 8) bzp[r] = z0[i < 4 ? i + 4 : (i < 8 ? i - 4 : i)];
            bsl[r] = bz[r] - lb;
            bsu[r] = ub - bz[r];
            smin = fmin(smin, fmin(bsl[r], bsu[r]));
        }
    };
    // (wave 0 is idle in the evaluation phase: it evaluates the dynamics Hessian there, from what wave 1 publishes)
    if constexpr (IS_M) {
        if (lane < 9) xs[X_XINIT + lane] = a.xinit[(size_t)b * 9 + lane];
        if (kact) {
            const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
#pragma unroll
            for (int i = 0; i < NZ; i++) ms.z[i] = z0[i];
            ms.fext[0] = pk[3]; ms.fext[1] = pk[4]; ms.fext[2] = pk[5];
            if (half == 0) {
                if constexpr (!QS) { xs[X_FEXT + k] = ms.fext[0]; xs[X_FEXT + NP + k] = ms.fext[1]; xs[X_FEXT + 2 * NP + k] = ms.fext[2]; }
                ldouble *rec = recs + k * RS;
                rec[R_HC] = -2.0 * pk[8]; // (u_i, w_i) cost coupling of this stage (constant)
                rec[R_ZERO] = 0.0; rec[R_ZERO2] = 0.0; rec[R_ONE] = 1.0; rec[R_DT] = (TW && k < tw_m) ? -DT : DT; rec[R_DUMP] = 0.0;
                if (k == N - 1) { // no dynamics behind the last stage: its M row stays zero
#pragma unroll
                    for (int i = 0; i < 64; i++) rec[R_LIN + i] = 0.0;
                }
                rec[R_CB + 0] = rec[R_CB + 1] = rec[R_CB + 2] = 0.0;
                rec[R_CC + 0] = rec[R_CC + 1] = rec[R_CC + 2] = 0.0;
#pragma unroll
                for (int i = 0; i < NZ; i++) rec[R_DZ + i] = 0.0; // (the first evaluation reads 0 * step: not a predecessor's NaN)
#pragma unroll
                for (int i = 0; i < 4; i++) rec[RT_HU + i] = ms.z[i]; // the Hessian's inputs of the first iteration (y = 0)
#pragma unroll
                for (int i = 0; i < 6; i++) { rec[RT_HVE + i] = ms.z[11 + i]; rec[RT_HY + i] = 0.0; if constexpr (QP) rec[RT_HYP + i] = 0.0; }
            }
        }
    }
    if constexpr (IS_B) bounds_init();
    if constexpr (IS_F) {
        int nf = 0;
        if (kact) {
            if (a.nfaces) nf = a.nfaces[(size_t)b * N + k];
            else nf = count_live_faces(pk, M);
            if (nf > MF || nf < 0 || nf > FL * H) { bad_param = 1; nf = 0; }
            if (half == 0) mcount = 34 + nf;
            const double *z0 = a.x0 + ((size_t)b * N + k) * NZ;
            if constexpr (!FPZ) { fpos[0] = z0[8]; fpos[1] = z0[9]; fpos[2] = z0[10]; }
#pragma unroll
            for (int t = 0; t < FL; t++) {
                const int j = t * H + half;
                if (j < nf) {
                    const double a0 = pk[NPRE + 3 * j], a1 = pk[NPRE + 3 * j + 1], a2 = pk[NPRE + 3 * j + 2];
                    const double bb = pk[NPRE + 3 * M + j] + HU;
                    if (FREG) { fa0[t] = a0; fa1[t] = a1; fa2[t] = a2; if (PARK) prk[t] = bb; else fbb[t] = bb; }
                    fs[t] = -(a0 * fpq[0] + a1 * fpq[1] + a2 * fpq[2] - bb);
                    smin = fmin(smin, fs[t]);
                }
            }
        }
        nfk = nf;
    }
    smin = wave_min(smin);
    if constexpr (IS_F) {
        const int mt = (int)wave_sum((double)mcount);
        const int bd = wave_max((double)bad_param) > 0.0 ? 1 : 0;
        if (lane == 0) { sh.ctl->mtot = mt; sh.ctl->bad = bd; }
    }
    publish(xs, wave, lane, 13, smin);
    BAR();
    const int mtot = sh.ctl->mtot;
    if (sh.ctl->bad) { // a stage has more live corridor rows than the caller sized the problem for (MF)
        if (wave == 0 && lane == 0) {
            if (QW && sh.ctl->head_first >= 0) { sh.ctl->head_first = -1; atomicSub(a.cu_slots + sh.cukey, 256); atomicSub(a.counter + 1, 1); } // (a head-start solve gives its CU back)
            sh.ctl->next = claim_next(a, sh.cukey); a.exitflag[b] = FRP_EXIT_PARAM_VALUE; a.iters[b] = 0;
        }
        if (wave == 1 && own0) {
            double *zo = a.z + ((size_t)b * N + k) * NZ;
#pragma unroll
            for (int i = 0; i < NZ; i++) zo[i] = ms.z[i];
        }
        return;
    }
    {
        // infeasible-start initialisation: uniform slack shift (see oracle/nmpc_ipm.c)
        smin = rmin(13);
        const double shift = (smin >= S_MIN) ? 0.0 : (S_MIN - smin) + fmax(0.0, -smin);
        if constexpr (IS_B) {
#pragma unroll
            for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
                bsl[r] += shift; bsu[r] += shift;
                bll[r] = a.mu0 / bsl[r]; blu[r] = a.mu0 / bsu[r];
            }
        }
        if constexpr (IS_F) {
#pragma unroll
            for (int t = 0; t < FL; t++) { fs[t] += shift; fl_[t] = a.mu0 / fs[t]; }
        }
    }

    const double inv_kmtot_r = 1.0 / (KAPPA_LAM * (double)mtot);
    if constexpr (PARK) { // (this wave keeps it in the workgroup scratch: in a register it is spilled, and the commit waited for the reload; first read: barriers later)
        if (lane == 0) xs[X_IKM] = inv_kmtot_r;
    }
#define inv_kmtot (PARK ? (double)xs[X_IKM] : inv_kmtot_r)
    int flag = FRP_EXIT_MAXIT, it = 0, nfallback = 0;
    bool iso_mine = false; // (Riccati wave) this solve holds a mark on its CU
    if constexpr (QW && wave == 0) iso_mine = uni(sh.ctl->head_first) >= 0; // (a head-start solve: marked by the kernel prologue)
    // Twisted variants: the iterations are solved the twisted way only until the residuals of an iteration fall below TW_EXACT_BELOW,
    // the ones after that (one-way switch, decided one iteration ahead) by the plain recursion
    // (the twisted solve is an inexact Newton method -- penalty on x_0 --: the end game, and with it the accuracy of the returned point,
    // is the plain recursion's; same rule and constant as oracle/nmpc_ipm.c).  Decided one iteration ahead because the model phase
    // writes the first-half records in another form.
    constexpr double TW_EXACT_BELOW = 1e-4;
    bool tw_on = true;
    int tw_it = tw_m; // stages eliminated forward in THIS iteration (0: the plain recursion)
    double theta_h = hess ? 1.0 : 0.0; // weight of the dynamics Hessian
    bool gn_retry = false;              // this iteration is being redone with the Gauss-Newton Hessian
    Norms nm = {0, 0, 0, 0, 0, 0};
    double mu = 0.0, step_cc = 0.0;
#ifndef FRP_R_PRIO
#define FRP_R_PRIO 0
#endif
#ifndef FRP_H_PRIO
#define FRP_H_PRIO 2
#endif
    // Issue priority: the element-wise waves above the Riccati waves.  A SIMD hosts one wave of each of the three resident
    // workgroups; the short VALU-dense phases of one problem (model, bounds, faces) otherwise queue behind the long sweep
    // of another and take twice their stand-alone time, while a sweep (MFMA / latency bound) barely notices the extra
    // VALU traffic: measured 1.712 vs 1.743 ms per 4096-problem launch (the opposite assignment: no difference to none).
    if constexpr (wave == 0) __builtin_amdgcn_s_setprio(FRP_R_PRIO);
    else __builtin_amdgcn_s_setprio(FRP_H_PRIO);

    // Which wave runs the forward sweeps (experiment knobs; 0 = the Riccati wave itself).  With three workgroups per CU the
    // roles are placed by SIMD and SIMD 3 hosts no Riccati wave: a sweep handed to the wave that sits there (role 3) takes
    // its issue cycles off the SIMD the Riccati wave shares with two helpers of the other resident problems.
#ifndef FRP_FWD_WAVE_P
#define FRP_FWD_WAVE_P 0
#endif
#ifndef FRP_FWD_WAVE_C
#define FRP_FWD_WAVE_C 0
#endif
    constexpr int FWD_P = NP == 20 ? FRP_FWD_WAVE_P : 0, FWD_C = NP == 20 ? FRP_FWD_WAVE_C : 0;
    PROF_DECL();
    // (Q4 with the corridor rows on the Riccati wave: that wave runs its element-wise phases at the helpers' priority, the sweeps at its own)
    constexpr bool W0H = QW && wave == 0 && IS_F;
    for (it = 0;;) {
        // ============================================================ evaluation phase
        if constexpr (W0H) __builtin_amdgcn_s_setprio(FRP_H_PRIO);
        if constexpr (wave == 0) {
            if (hact) {
                HessState hs; // the model wave's values before the last step + the step
                cldouble *rec = recs + k * RS;
#pragma unroll
                for (int i = 0; i < 4; i++) hs.u[i] = rec[RT_HU + i] + hap * rec[R_DZ + i]; // (the step is in the record until the next forward sweep)
#pragma unroll
                for (int i = 0; i < 6; i++) hs.ve[i] = rec[RT_HVE + i] + hap * rec[R_DZ + 11 + i];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double yo = (k < N - 1) ? rec[RS + RT_HY + i] : 0.0;
                    const double yp = QP ? ((k < N - 1) ? rec[RS + RT_HYP + i] : 0.0) : hyp[i]; // (QP: left in the record by the y+ lanes)
                    hs.y6[i] = yo + hap * (yp - yo);
                }
                if constexpr (QS) { hs.fext[0] = pk[3]; hs.fext[1] = pk[4]; hs.fext[2] = pk[5]; } // (no room for [3][NP] in the workgroup scratch: from the parameters, L2-resident)
                else { hs.fext[0] = xs[X_FEXT + k]; hs.fext[1] = xs[X_FEXT + NP + k]; hs.fext[2] = xs[X_FEXT + 2 * NP + k]; }
                WSYNC(); // (the neighbour lane's reads of this stage's RT_HY slots come before the scratch use of the record below)
                if constexpr (H3) hessian_phase3(recs + k * RS, hs, half, k < N - 1, hess);
                else hessian_phase(recs + k * RS, hs, k < N - 1, hess);
            }
        }
        auto model_block = [&]() __attribute__((always_inline)) {
            double l_eq;
            if constexpr (TW) { // the form of this iteration's first-half records: -dt / the B~ entry in the second zero, or dt / zero
                if (own0 && k < tw_m) {
                    ldouble *rec = recs + k * RS;
                    rec[R_DT] = tw_it ? -DT : DT;
                    if (!tw_it) rec[R_ZERO2] = 0.0;
                }
            }
            if constexpr (H3) model_phase3<NP>(recs, xs, ms, N, l_eq, tw_it);
            else model_phase<NP>(recs, xs, ms, N, l_eq);
            publish(xs, WEQ, lane, 0, wave_max(l_eq));
        };
        // (Q4, corridor rows on the model wave: their state does not fit beside the model phase's temporaries -- ~25 scratch accesses, 14.9 k
        // cycles for the phase instead of ~10 k.  Measured and dropped: evaluating the rows first, parking their 15 doubles in the global
        // workspace across the model phase and fetching them back in one batch behind it -- the barrier waits for the loads: 18.7 k.)
        // Q4, corridor rows on the model wave: which of the two runs first (FRP_Q4_FACES_FIRST: the rows' residuals only need z; with the
        // rows ahead of the model their accumulators are dead before the model's temporaries come alive -- VERDICT r05 item 1a / DESIGN 9.5-5b)
#ifndef FRP_Q4_FACES_FIRST
#define FRP_Q4_FACES_FIRST 0
#endif
        constexpr bool FACES_FIRST = QW && IS_M && IS_F && FRP_Q4_FACES_FIRST;
#ifdef FRP_PROFILE_W1 // (profile build) where the model + corridor wave spends its evaluation phase: slots 16.. of the segment counters
        long long w1t_ = clock64();
#define W1_SEG(i) do { if constexpr (IS_M) { const long long tn_ = clock64(); if (lane == 0) atomicAdd((unsigned long long *)&g_prof_seg[16 + (i)], (unsigned long long)(tn_ - w1t_)); w1t_ = tn_; } } while (0)
#else
#define W1_SEG(i)
#endif
        if constexpr (IS_M && !FACES_FIRST) { model_block(); W1_SEG(0); }
        if constexpr (IS_BO) {
            double l_in = 0.0, l_rc = 0.0, l_gap = 0.0;
            bounds_eval(l_in, l_rc, l_gap);
            publish(xs, wave, lane, 0, wave_max(l_in)); publish(xs, wave, lane, 1, wave_max(l_rc)); publish(xs, wave, lane, 2, wave_sum_mx(l_gap));
        }
        if constexpr (IS_F) {
            double l_in = 0.0, l_rc = 0.0, l_gap = 0.0;
            bounds_eval(l_in, l_rc, l_gap);
            double gp0 = 0, gp1 = 0, gp2 = 0, fp0 = 0, fp1 = 0, fp2 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
            if (kact) {
                face_bases();
                for_faces([&](int t, double a0, double a1, double a2, double bb) __attribute__((always_inline)) {
                    const double hj = a0 * fpq[0] + a1 * fpq[1] + a2 * fpq[2] - bb;
                    const double sc = fs[t], lc = fl_[t];
                    const double rc = hj + sc;
                    l_in = fmax(l_in, fmax(hj, fabs(rc)));
                    l_rc = fmax(l_rc, sc * lc);
                    l_gap += sc * lc;
                    gp0 += a0 * lc; gp1 += a1 * lc; gp2 += a2 * lc;
                    const double sg = lc * fast_rcp(sc), tt = sg * rc;
                    fp0 += a0 * tt; fp1 += a1 * tt; fp2 += a2 * tt;
                    p0 += sg * a0 * a0; p1 += sg * a0 * a1; p2 += sg * a0 * a2;
                    p3 += sg * a1 * a1; p4 += sg * a1 * a2; p5 += sg * a2 * a2;
                });
            }
            if (H > 1) {
                gp0 = xsub_sum<NP>(gp0); gp1 = xsub_sum<NP>(gp1); gp2 = xsub_sum<NP>(gp2);
                fp0 = xsub_sum<NP>(fp0); fp1 = xsub_sum<NP>(fp1); fp2 = xsub_sum<NP>(fp2);
                p0 = xsub_sum<NP>(p0); p1 = xsub_sum<NP>(p1); p2 = xsub_sum<NP>(p2);
                p3 = xsub_sum<NP>(p3); p4 = xsub_sum<NP>(p4); p5 = xsub_sum<NP>(p5);
            }
            if (kact && half == 0) {
                ldouble *rec = recs + k * RS;
                rec[R_PHIPOS + 0] = p0; rec[R_PHIPOS + 1] = p1; rec[R_PHIPOS + 2] = p2;
                rec[R_PHIPOS + 3] = p1; rec[R_PHIPOS + 4] = p3; rec[R_PHIPOS + 5] = p4;
                rec[R_PHIPOS + 6] = p2; rec[R_PHIPOS + 7] = p4; rec[R_PHIPOS + 8] = p5;
                rec[R_CB + 0] = gp0; rec[R_CB + 1] = gp1; rec[R_CB + 2] = gp2;
                rec[R_CC + 0] = fp0; rec[R_CC + 1] = fp1; rec[R_CC + 2] = fp2;
            }
            publish(xs, wave, lane, 0, wave_max(l_in)); publish(xs, wave, lane, 1, wave_max(l_rc)); publish(xs, wave, lane, 2, wave_sum_mx(l_gap));
            W1_SEG(1);
        }
        if constexpr (FACES_FIRST) { FRP_SB(); model_block(); W1_SEG(0); }
        BAR_P(0); // ------------------------------------------------------------- A
        // FRP_EARLY_FACTOR: the Riccati wave does not take part in the termination test -- ~2 k cycles of LDS reads and reductions in front of
        // its 46 k-cycle predictor, on the chain of every iteration -- but starts the factorisation at once; the helper waves, idle until
        // barrier C anyway, take the decision and leave it in Ctl::dec_*; the factorisation sweep looks for it between its passes and the
        // wave leaves the iteration like the others when it says "stop" (once per solve: a few stages of a sweep nobody needs).  What the
        // wave needs of the norms later (mu, for the centring parameter) it reads from the partial results after its sweeps; the values
        // it reports at exit it computes there.
#ifndef FRP_EARLY_FACTOR
#define FRP_EARLY_FACTOR 0
#endif
        constexpr bool EARLY = FRP_EARLY_FACTOR && !TW && FWD_P == 0;
        constexpr bool EARLY_R = EARLY && wave == 0;
        const unsigned dec_want = ((unsigned)b << 8) | (unsigned)(it & 255);
        bool tw_next = false;
        if constexpr (!EARLY_R) {
            // (workgroup-uniform scalars that live across phases go to scalar registers: the element-wise roles are at the
            // 168-VGPR cap of three workgroups per CU, and every spilled value is an L2 round trip on an in-order wavefront)
            nm.eq = uni(red(xs, WEQ, 0));
            nm.in = uni(rmax(0));
            nm.rc = uni(rmax(1));
            nm.gap = uni(rsum(2));
            nm.rs = uni(stationarity_norm<NP>(recs, N));
            mu = uni(nm.gap * (KAPPA_LAM * inv_kmtot));
            tw_next = TW && fmax(fmax(nm.eq, nm.in), fmax(nm.rs, nm.rc)) > TW_EXACT_BELOW; // (for the NEXT iteration)
            int stop = 0, stop_flag = flag;
            if (!gn_retry) {
                if (!(nm.eq == nm.eq) || !(nm.rs == nm.rs) || !(nm.gap == nm.gap)) { stop_flag = FRP_EXIT_BADFUNCEVAL; stop = 1; }
                else if (nm.eq <= a.tol_eq && nm.in <= a.tol_ineq && nm.rs <= a.tol_stat && nm.rc <= a.tol_comp) { stop_flag = FRP_EXIT_OPTIMAL; stop = 1; }
                else if (it >= a.maxit) { stop_flag = FRP_EXIT_MAXIT; stop = 1; }
                else if (mu > a.diverge_mu * fmax(1.0, a.mu0) || nm.rs > DIVERGE_RS) { stop_flag = FRP_EXIT_NOPROGRESS; stop = 1; }
            }
            if constexpr (EARLY && wave == 2) {
                if (lane == 0) { sh.ctl->dec_flag = stop_flag; sh.ctl->dec_stop = stop; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); *(volatile unsigned *)&sh.ctl->dec_tag = dec_want; }
            }
            if (stop) { flag = stop_flag; break; }
        }
        if constexpr (QW && wave == 0) { // a long solve: it takes the CU for itself (see claim_next)
            if (it == a.iso_it && a.iso_it > 0 && sh.cukey >= 0 && !iso_mine) {
                int got = 0;
                if (lane == 0) {
                    got = atomicAdd(a.counter + 1, 1) < a.iso_cap;
                    if (got) atomicAdd(a.cu_slots + sh.cukey, 256);
                    else atomicSub(a.counter + 1, 1);
                }
                iso_mine = uni(__shfl(got, 0)) != 0;
            }
        }

        // ============================================================ predictor: factorisation + forward sweep
        if constexpr (W0H) __builtin_amdgcn_s_setprio(FRP_R_PRIO);
        if (TW && !tw_it) { // (twisted variants, end game: the plain recursion -- behind ONE call, so that the twisted path keeps its registers)
            if constexpr (wave == 0) {
                const int fr = endgame_predictor(recs, xs, N, gn_retry ? 0.0 : theta_h FRP_GP_ARG(pws));
                if (lane == 0) sh.ctl->fail = fr;
            }
        } else if constexpr (TW) {
            // the two halves side by side; the Riccati wave factors and solves the meeting system; both continue from ds_m, outwards
            TW_T0();
            if constexpr (wave == 0) {
                const int fr = sweep_factor<true>(recs + tw_m * RS, xs, N - tw_m, gn_retry ? 0.0 : theta_h FRP_GP_ARG(nullptr));
                if (lane == 0) sh.ctl->fail = fr;
                TW_T1(22);
            } else if constexpr (wave == 1) {
                const int fr = sweep_arrive(recs, xs, tw, tw_m, gn_retry ? 0.0 : theta_h);
                if (lane == 0) sh.ctl->fail1 = fr;
                TW_T1(23);
            }
            BAR();
            if constexpr (wave == 0) { // the system where the halves meet: factored and solved by this wave, ds_m -> X_DS0
                TW_T1(6);
                int mf = 0;
                if (!(sh.ctl->fail | sh.ctl->fail1)) {
                    mf = meet_factor(recs, tw, tw_m);
                    TW_T1(24);
                    if (!mf) meet_solve(recs, tw, tw_m, xs + X_DS0);
                    TW_T1(25);
                }
                if (lane == 0) sh.ctl->fail2 = mf;
            }
            BAR();
            if constexpr (wave <= 1) {
                TW_T1(7);
                if (!(sh.ctl->fail | sh.ctl->fail1 | sh.ctl->fail2)) {
                    if constexpr (wave == 0) sweep_forward(recs + tw_m * RS, xs, N - tw_m);
                    else sweep_backsub(recs, xs + X_DS0, tw_m);
                }
                TW_T1(26 + wave);
            }
        } else if constexpr (wave == 0) {
            SWEEP_T0();
            lctl_t *lctl = (lctl_t *)sh.ctl;
            int fr = sweep_factor<false, EARLY>(recs, xs, N, gn_retry ? 0.0 : theta_h FRP_GP_ARG(pws), lctl, dec_want);
            if constexpr (EARLY) {
                bool stopped = (fr & 2) != 0;
                if (!stopped && !(fr & 4)) { // the sweep is through and the helpers' decision is not in yet (never observed to take that long)
                    while (uni((int)*(volatile __attribute__((address_space(3))) unsigned *)&lctl->dec_tag) != (int)dec_want) __builtin_amdgcn_s_sleep(1);
                    stopped = uni(*(volatile __attribute__((address_space(3))) int *)&lctl->dec_stop) != 0;
                }
                if (stopped) {
                    flag = uni(*(volatile __attribute__((address_space(3))) int *)&lctl->dec_flag);
                    if constexpr (QP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    break;
                }
                fr &= 1;
            }
            SWEEP_T1(0);
            if (FWD_P == 0 && !fr) sweep_forward(recs, xs, N);
            SWEEP_T1(1);
            if (lane == 0) sh.ctl->fail = fr;
            if constexpr (QP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the P stores of the factorisation sweep (long retired by now)
            if constexpr (EARLY) { nm.gap = uni(rsum(2)); mu = uni(nm.gap * (KAPPA_LAM * inv_kmtot)); } // (for the centring parameter behind barrier D)
        }
        if constexpr (FWD_P != 0 && !TW) { // the forward sweep works from the records alone: it runs on the wave whose SIMD has room
            BAR();
            if constexpr (wave == FWD_P) {
                if (!sh.ctl->fail) sweep_forward(recs, xs, N);
            }
        }
        BAR_P(1); // ------------------------------------------------------------- C
        {
            const int fr = TW ? (sh.ctl->fail | sh.ctl->fail1 | sh.ctl->fail2) : sh.ctl->fail;
            if (fr && !gn_retry && theta_h > 0.0) {
                // indefinite pivot block with the exact Hessian: the iteration is redone with the Gauss-Newton Hessian.
                // P has overwritten part of the barrier Hessian in the records, so the evaluation phase runs again.
                nfallback++;
                theta_h *= THETA_DOWN;
                gn_retry = true;
                // the factorisation has overwritten the T' slots: the model wave hands the Hessian's inputs over again (no step
                // in between), behind a barrier of its own (0.05 % of solves come here)
                if constexpr (wave == 1) {
                    if (own0) {
                        ldouble *rec = recs + k * RS;
#pragma unroll
                        for (int i = 0; i < 4; i++) rec[RT_HU + i] = ms.z[i];
#pragma unroll
                        for (int i = 0; i < 6; i++) { rec[RT_HVE + i] = ms.z[11 + i]; rec[RT_HY + i] = ms.y[4 + i]; if constexpr (QP) rec[RT_HYP + i] = 0.0; }
                    }
                }
                hap = 0.0;
                BAR();
                continue;
            }
            if (fr) { flag = FRP_EXIT_FACTORIZATION; break; }
            if (!gn_retry && hess) theta_h = fmin(1.0, theta_h + THETA_UP);
            gn_retry = false;
        }

        // ============================================================ affine step: lengths, second-order term, corrector rhs
        if constexpr (W0H) __builtin_amdgcn_s_setprio(FRP_H_PRIO);
        if constexpr (IS_BO) {
            double m_p = 0.0, m_d = 0.0, s_sdl = 0.0, s_lds = 0.0, s_dsdl = 0.0;
            bounds_affine(m_p, m_d, s_sdl, s_lds, s_dsdl);
            publish(xs, wave, lane, 3, wave_max(m_p)); publish(xs, wave, lane, 4, wave_max(m_d));
            publish(xs, wave, lane, 5, wave_sum_mx(s_sdl)); publish(xs, wave, lane, 6, wave_sum_mx(s_lds)); publish(xs, wave, lane, 7, wave_sum_mx(s_dsdl));
        }
        if constexpr (IS_F) {
            double m_p = 0.0, m_d = 0.0, s_sdl = 0.0, s_lds = 0.0, s_dsdl = 0.0;
            bounds_affine(m_p, m_d, s_sdl, s_lds, s_dsdl);
            double b0 = 0, b1 = 0, b2 = 0, c0 = 0, c1 = 0, c2 = 0;
            if (kact) {
                cldouble *rec = recs + k * RS;
                const double d8 = rec[R_DZ + 8], d9 = rec[R_DZ + 9], d10 = rec[R_DZ + 10];
                face_bases();
                for_faces([&](int t, double a0, double a1, double a2, double bb) __attribute__((always_inline)) {
                    const double s = fs[t], l = fl_[t];
                    const double gdz = a0 * d8 + a1 * d9 + a2 * d10, viol = a0 * fpq[0] + a1 * fpq[1] + a2 * fpq[2] - bb;
                    const double u = fast_rcp(s * l);
                    const double sinv = u * l, linv = u * s;
                    const double rin = viol + s;
                    const double ds = -rin - gdz;
                    const double dl = -l * (1.0 + ds * sinv);
                    m_p = fmax(m_p, -ds * sinv);
                    m_d = fmax(m_d, -dl * linv);
                    s_sdl += s * dl; s_lds += l * ds;
                    const double cr = ds * dl;
                    s_dsdl += cr;
                    if (PARK) prk[FL + t] = cr; else fcr[t] = cr;
                    const double t1 = (l * rin - cr) * sinv;
                    b0 += a0 * t1; b1 += a1 * t1; b2 += a2 * t1;
                    c0 += a0 * sinv; c1 += a1 * sinv; c2 += a2 * sinv;
                });
            }
            if (H > 1) {
                b0 = xsub_sum<NP>(b0); b1 = xsub_sum<NP>(b1); b2 = xsub_sum<NP>(b2);
                c0 = xsub_sum<NP>(c0); c1 = xsub_sum<NP>(c1); c2 = xsub_sum<NP>(c2);
            }
            if (kact && half == 0) {
                ldouble *rec = recs + k * RS;
                rec[R_CB + 0] = b0; rec[R_CB + 1] = b1; rec[R_CB + 2] = b2;
                rec[R_CC + 0] = c0; rec[R_CC + 1] = c1; rec[R_CC + 2] = c2;
            }
            publish(xs, wave, lane, 3, wave_max(m_p)); publish(xs, wave, lane, 4, wave_max(m_d));
            publish(xs, wave, lane, 5, wave_sum_mx(s_sdl)); publish(xs, wave, lane, 6, wave_sum_mx(s_lds)); publish(xs, wave, lane, 7, wave_sum_mx(s_dsdl));
        }
        BAR_P(2); // ------------------------------------------------------------- D
        double smu;
        {
            const double m_p = rmax(3), m_d = rmax(4);
            const double ap = (m_p > 1.0) ? fast_rcp(m_p) : 1.0;
            const double ad = (m_d > 1.0) ? fast_rcp(m_d) : 1.0;
            const double gap_aff = mu * (double)mtot + ad * rsum(5) + ap * rsum(6) + ap * ad * rsum(7);
            double sigma = gap_aff * fast_rcp((double)mtot * mu);
            sigma = sigma * sigma * sigma;
            if (sigma > 1.0) sigma = 1.0;
            smu = sigma * mu;
            if (smu < MU_FLOOR_FRAC * a.tol_comp) smu = MU_FLOOR_FRAC * a.tol_comp;
            smu = uni(smu);
            // reported, not iterated on (info.mu_aff / sigma / step_aff, FORCESNLPsolver_normal.h:275-289): every wave has the values;
            // the faces wave -- the one on the SIMD without a Riccati wave -- parks them in three free scratch slots, the Riccati wave
            // picks them up at exit (carried in ITS registers across the sweeps they cost 0.5 % of the launch, measured)
            if constexpr (IS_F) {
                if (lane == 0) { xs[X_RED + 3 * 16 + 14] = gap_aff * fast_rcp((double)mtot); xs[X_RED + 3 * 16 + 15] = sigma; xs[X_RED + 2 * 16 + 14] = ap; }
            }
        }

        // QP: the bounds wave -- idle until barrier E -- fetches its block of S_xx (16 doubles per lane, one 128-byte line: the lane's
        // diagonal block and the block to its right, stored by the factorisation sweep of this iteration) for the y+ rows of the step phase
        // (FRP_QP_YWAVE = 2; the default is the Riccati wave, which fetches at the start of the step phase: the bounds wave has no
        // room for 32 more registers between the barriers D and F -- 310 spilled VGPRs instead of 126)
        if constexpr (wave == WY && WY != 0 && QP) fetch_sx();
#ifndef FRP_QP_TOUCH // experiment knob: the bounds wave -- idle from here to barrier E -- reads one word of every 128-byte line of this workgroup's S_xx blocks, so
#define FRP_QP_TOUCH 0 // that the Riccati wave's fetch at the head of the step phase finds them in the CU's vector cache (one register, waited for before barrier E)
#endif
        unsigned touch_ = 0;
        if constexpr (FRP_QP_TOUCH != 0 && QP && !QS && WY == 0 && IS_BO) {
            const char *tp_ = (const char *)pws + (unsigned)(lane < 3 * N ? lane : 0) * 128u;
            asm volatile("global_load_dword %0, %1, off" : "=v"(touch_) : "v"(tp_) : "memory");
        }
        if constexpr (W0H) __builtin_amdgcn_s_setprio(FRP_R_PRIO);

        // ============================================================ corrector: vector backward sweep + forward sweep with y+
        if (TW && !tw_it) {
            if constexpr (wave == 0) endgame_corrector(recs, xs, N, smu);
        } else if constexpr (TW) {
            TW_T0();
            if constexpr (wave == 0) {
                sweep_backvec<true>(recs + tw_m * RS, xs, N - tw_m, smu);
                TW_T1(28);
            } else if constexpr (wave == 1) {
                sweep_arrive_vec(recs, xs, tw, tw_m, smu);
                TW_T1(29);
            }
            BAR();
            if constexpr (wave == 0) {
                TW_T1(12);
                meet_solve(recs, tw, tw_m, xs + X_DS0);
                TW_T1(14);
                sweep_forward(recs + tw_m * RS, xs, N - tw_m);
                TW_T1(30);
            } else if constexpr (wave == 1) {
                TW_T1(13);
                meet_solve(recs, tw, tw_m, tw + TW_DS); // (both waves solve, from the stored factor: no second barrier)
                TW_T1(15);
                sweep_backsub(recs, tw + TW_DS, tw_m);
                TW_T1(31);
            }
        } else if constexpr (wave == 0) {
            SWEEP_T0();
            sweep_backvec<false>(recs, xs, N, smu);
            SWEEP_T1(2);
            if (FWD_C == 0) sweep_forward(recs, xs, N);
            SWEEP_T1(3);
        }
        if constexpr (FWD_C != 0 && !TW) {
            BAR();
            if constexpr (wave == FWD_C) sweep_forward(recs, xs, N);
        }
        if constexpr (FRP_QP_TOUCH != 0 && QP && !QS && WY == 0 && IS_BO) asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch_) : : "memory");
        BAR_P(3); // ------------------------------------------------------------- E

        // ============================================================ step: pass A (ratios), pass B (commit)
        if constexpr (W0H) __builtin_amdgcn_s_setprio(FRP_H_PRIO);
        double m_p = 0.0, m_d = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0; // q: sums of ds l, s dl, ds dl
        double dzf[3];                            // wave 3: dz of pos (the commit reads every dz from the record again: nothing is carried across barrier F)
        // one constraint of the corrector step
        auto cstep = [&](double s, double l, double corr, double gdz, double viol, double &ds, double &dl) {
            const double u = fast_rcp(s * l);
            const double sinv = u * l, linv = u * s;
            ds = -(viol + s) - gdz;
            const double rc = s * l - smu + corr;
            dl = (-rc - l * ds) * sinv;
            m_p = fmax(m_p, -ds * sinv);
            m_d = fmax(m_d, -dl * linv);
            q1 = fma(ds, l, q1); q2 = fma(s, dl, q2); q3 = fma(ds, dl, q3);
        };
        // y+_k = P_k ds_k + p_k of the Newton system (P_k packed lower triangle): formed by the waves that consume it, off
        // the forward sweep's dependency chain
        // (twisted solve, stages of the first half: the record holds the arrival cost, y+_k = -(Q_k ds_k + q_k))
        auto y_plus = [&](cldouble *rec, int i) {
            double acc = rec[R_PV + i];
#pragma unroll
            for (int j = 0; j < NS; j++) acc = fma(rec[R_P + (i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i)], rec[R_DZ + 4 + j], acc);
            return (TW && k < tw_it) ? -acc : acc;
        };
        // QP: y+ is formed by the bounds wave alone, from its prefetched block of S_xx (registers, see the prefetch behind barrier D)
        // and from T', p, Phi_w, hc in the record -- no global memory access on the path:
        //     y_w = Phi_w dw + hc (du + kbar) + p_w        (the w rows of P are [Phi_w - hc^2 R | -hc Kbar_x] and du = -(hc R dw + Kbar_x dx + kbar))
        //     y_x = S_xx dx - hc Kbar_x' dw + p_x          (lane a of a stage: rows 3a .. 3a+2; S_aa and S_{a,a+1} are its own, the
        //                                                   contribution S_{a,a+1}' dx_a goes to the lane that owns the rows a+1)
        if constexpr (wave == WY && QP && QS) {
            // Q30: the lane of a stage forms all thirteen rows itself -- its three blocks of S_xx (block b: S_bb lower triangle, then S_{b,b+1} row-major, b + 1 cyclic:
            // every unique entry of the symmetric 9 x 9 once) in one trip to the L2, T', p, Phi_w, hc from its record; no hand-over between lanes
            static_assert(WY == 0, "Q30: y+ on the Riccati wave");
            __builtin_amdgcn_s_setprio(FRP_H_PRIO);
            if (hact) {
                typedef double gd2 __attribute__((ext_vector_type(2)));
                ldouble *rec = recs + k * RS;
                const __attribute__((address_space(1))) gd2 *src = (const __attribute__((address_space(1))) gd2 *)((const char *)pws + (unsigned)opq(k) * (PG * 8));
                double sb[PG];
#pragma unroll
                for (int q = 0; q < PG / 2; q++) { const gd2 v = src[q]; sb[2 * q] = v.x; sb[2 * q + 1] = v.y; }
                const double hcq = rec[R_HC];
                double dx[9], dw[4], yx[9], yw[4];
#pragma unroll
                for (int i = 0; i < 9; i++) { dx[i] = rec[R_DZ + 8 + i]; yx[i] = 0.0; }
#pragma unroll
                for (int g = 0; g < 4; g++) dw[g] = rec[R_DZ + 4 + g];
#pragma unroll
                for (int bq = 0; bq < 3; bq++) {
                    const int ia = 3 * bq, ib = bq == 2 ? 0 : ia + 3;
                    const double *S = sb + 16 * bq, *da = dx + ia, *db = dx + ib;
                    yx[ia + 0] += S[0] * da[0] + S[1] * da[1] + S[3] * da[2];
                    yx[ia + 1] += S[1] * da[0] + S[2] * da[1] + S[4] * da[2];
                    yx[ia + 2] += S[3] * da[0] + S[4] * da[1] + S[5] * da[2];
#pragma unroll
                    for (int t = 0; t < 3; t++) yx[ia + t] += S[6 + 3 * t] * db[0] + S[7 + 3 * t] * db[1] + S[8 + 3 * t] * db[2];
#pragma unroll
                    for (int q = 0; q < 3; q++) yx[ib + q] += S[6 + q] * da[0] + S[9 + q] * da[1] + S[12 + q] * da[2];
                }
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const double kd = rec[R_T + 4 + i] * dw[0] + rec[R_T + 20 + i] * dw[1] + rec[R_T + 36 + i] * dw[2] + rec[R_T + 52 + i] * dw[3];
                    yx[i] = __builtin_fma(-hcq, kd, yx[i]) + rec[R_PV + 4 + i];
                }
#pragma unroll
                for (int g = 0; g < 4; g++)
                    yw[g] = __builtin_fma(rec[R_PHIW + g], dw[g], __builtin_fma(hcq, rec[R_DZ + g] + rec[R_T + 16 * g + 13], rec[R_PV + g]));
                // (every read of T' by this lane is behind it: RT_HYP is a Kbar_x slot of its own record)
#pragma unroll
                for (int i = 0; i < 9; i++) rec[R_D + 4 + i] = yx[i];
#pragma unroll
                for (int i = 0; i < 6; i++) rec[RT_HYP + i] = yx[i]; // rows 4..9 once more, for the Hessian lane of stage k - 1 in the next evaluation
#pragma unroll
                for (int g = 0; g < 4; g++) rec[R_D + g] = yw[g];
            }
            __builtin_amdgcn_s_setprio(FRP_R_PRIO);
        } else if constexpr (wave == WY && QP) {
            if constexpr (WY == 0) { __builtin_amdgcn_s_setprio(FRP_H_PRIO); fetch_sx(); } // (one trip to the L2 on the path: 8 x 16 bytes per lane)
            ldouble *rec = recs + k * RS;
            const int aq = opq(half) < 3 ? opq(half) : 2, ia = 3 * aq, ib = aq == 2 ? 0 : ia + 3;
            double ux[3], yw[4];
            {
                const double hcq = rec[R_HC];
                double da[3], db[3], vx[3];
#pragma unroll
                for (int t = 0; t < 3; t++) { da[t] = rec[R_DZ + 8 + ia + t]; db[t] = rec[R_DZ + 8 + ib + t]; }
                // S_aa: sx[0] = (0,0), sx[1] = (1,0), sx[2] = (1,1), sx[3] = (2,0), sx[4] = (2,1), sx[5] = (2,2);  S_ab[t][s] = sx[6 + 3 t + s]
                ux[0] = sx[0] * da[0] + sx[1] * da[1] + sx[3] * da[2];
                ux[1] = sx[1] * da[0] + sx[2] * da[1] + sx[4] * da[2];
                ux[2] = sx[3] * da[0] + sx[4] * da[1] + sx[5] * da[2];
#pragma unroll
                for (int t = 0; t < 3; t++) ux[t] += sx[6 + 3 * t] * db[0] + sx[7 + 3 * t] * db[1] + sx[8 + 3 * t] * db[2];
#pragma unroll
                for (int q = 0; q < 3; q++) vx[q] = sx[6 + q] * da[0] + sx[9 + q] * da[1] + sx[12 + q] * da[2];
                // - hc Kbar_x' dw and p_x
                double dw[4];
#pragma unroll
                for (int g = 0; g < 4; g++) dw[g] = rec[R_DZ + 4 + g];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const double kd = rec[R_T + 4 + ia + t] * dw[0] + rec[R_T + 20 + ia + t] * dw[1] + rec[R_T + 36 + ia + t] * dw[2] + rec[R_T + 52 + ia + t] * dw[3];
                    ux[t] = __builtin_fma(-hcq, kd, ux[t]) + rec[R_PV + 4 + ia + t];
                }
#pragma unroll
                for (int g = 0; g < 4; g++)
                    yw[g] = __builtin_fma(rec[R_PHIW + g], dw[g], __builtin_fma(hcq, rec[R_DZ + g] + rec[R_T + 16 * g + 13], rec[R_PV + g]));
                WSYNC(); // every read of T' by this wave is behind us: its Kbar_x slots serve as scratch now
                if (kact) {
#pragma unroll
                    for (int q = 0; q < 3; q++) rec[RT_YV + ib + q] = vx[q];
                }
            }
            WSYNC();
            if (kact) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const double y = ux[t] + rec[RT_YV + ia + t];
                    rec[R_D + 4 + ia + t] = y;
                    rec[aq < 2 ? RT_HYP + ia + t : R_DUMP] = y; // rows 4..9 once more, for the Hessian lanes of the next evaluation
                }
                if (half == 0) {
#pragma unroll
                    for (int g = 0; g < 4; g++) rec[R_D + g] = yw[g];
                }
            }
            if constexpr (WY == 0 && !W0H) __builtin_amdgcn_s_setprio(FRP_R_PRIO);
        }
        if constexpr (wave == 1 && QP) {
            if (own0) { // the values before the step, for the Hessian lanes (T' slots the y+ lanes do not read)
                ldouble *rec = recs + k * RS;
#pragma unroll
                for (int i = 0; i < 4; i++) rec[RT_HU + i] = ms.z[i];
#pragma unroll
                for (int i = 0; i < 6; i++) { rec[RT_HVE + i] = ms.z[11 + i]; rec[RT_HY + i] = ms.y[4 + i]; }
            }
        } else if constexpr (wave == 0 && QP) {
            // (nothing: y+ and the Hessian's inputs come from the other two waves)
        } else if constexpr (wave == 0) {
            // (the Newton step of the Hessian's inputs stays in the record: the next evaluation reads it there, before the forward
            // sweep that overwrites it)  y+ of the stage for the model wave's commit: it goes to the d slots of
            // the record, which are dead from the last sweep to the next model phase (the model wave has 60 registers of
            // persistent state; this wave has none)
            if (hact) {
                ldouble *rec = recs + k * RS;
                // (split over the three lanes of a stage by rows -- lane-dependent addresses into the packed triangle -- this measured
                // no faster: 5.4 k vs 4.7 k cycles for the phase on this wave, the launch time unchanged)
                // This wave forms the rows its own Hessian lanes read back (p, v: 4..9), the model wave -- idle until the commit --
                // the other seven.
                double yall[6];
#pragma unroll
                for (int i = 0; i < 6; i++) yall[i] = y_plus(rec, 4 + i);
                if (half == 0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) rec[R_D + 4 + i] = yall[i];
                }
            }
            WSYNC();
            if (hact) {
#pragma unroll
                for (int i = 0; i < 6; i++) hyp[i] = (k < N - 1) ? recs[(k + 1) * RS + R_D + 4 + i] : 0.0;
            }
        } else if constexpr (wave == 1) {
            if (own0) { // the values before the step, for the Hessian lanes (T' is dead from here to the next factorisation)
                ldouble *rec = recs + k * RS;
#pragma unroll
                for (int i = 0; i < 4; i++) rec[RT_HU + i] = ms.z[i];
#pragma unroll
                for (int i = 0; i < 6; i++) { rec[RT_HVE + i] = ms.z[11 + i]; rec[RT_HY + i] = ms.y[4 + i]; }
                // y+ rows 0..3 (w) and 10..12 (e) for the commit below; rows 4..9: the Riccati wave
#pragma unroll
                for (int i = 0; i < 4; i++) rec[R_D + i] = y_plus(rec, i);
#pragma unroll
                for (int i = 10; i < NS; i++) rec[R_D + i] = y_plus(rec, i);
            }
        }
        if constexpr (IS_B) { // bound rows of this wave
            if (kact) {
                cldouble *rec = recs + k * RS;
        const int halfp = opq(half);
#pragma unroll
        for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
                    const int ib = r * H, i = ib + half;
                    if (i >= NZ) continue;
                    const double lb = ROW_PICK(lower_bound(i));
                    const double ub = ROW_PICK(upper_bound(i));
                    const double zi = bz[r], dzi = rec[R_DZ + i];
                    double ds, dl;
                    cstep(bsl[r], bll[r], bcl[r], -dzi, lb - zi, ds, dl);
                    cstep(bsu[r], blu[r], bcu[r], dzi, zi - ub, ds, dl);
                }
            }
        }
        if constexpr (IS_F) { // corridor rows
            if (kact) {
                cldouble *rec = recs + k * RS;
                dzf[0] = rec[R_DZ + 8]; dzf[1] = rec[R_DZ + 9]; dzf[2] = rec[R_DZ + 10];
                face_bases();
                for_faces([&](int t, double a0, double a1, double a2, double bb) __attribute__((always_inline)) {
                    double ds, dl;
                    cstep(fs[t], fl_[t], PARK ? prk[FL + t] : fcr[t], a0 * dzf[0] + a1 * dzf[1] + a2 * dzf[2],
                          a0 * fpq[0] + a1 * fpq[1] + a2 * fpq[2] - bb, ds, dl);
                });
            }
        }
        if constexpr (IS_B) {
            publish(xs, wave, lane, 8, wave_max(m_p)); publish(xs, wave, lane, 9, wave_max(m_d));
            publish(xs, wave, lane, 10, wave_sum_mx(q1)); publish(xs, wave, lane, 11, wave_sum_mx(q2)); publish(xs, wave, lane, 12, wave_sum_mx(q3));
        }
        BAR_P(4); // ------------------------------------------------------------- F
        {
            const double mp = rmax(8), md = rmax(9);
            const double ap = uni((mp > a.ftb) ? a.ftb * fast_rcp(mp) : 1.0);
            const double ad = uni((md > a.ftb) ? a.ftb * fast_rcp(md) : 1.0);
            // multiplier safeguard: s_i lam_i >= mu_new / KAPPA_LAM for every pair after the step
            const double fprod = uni((mu * (double)mtot + ap * rsum(10) + ad * (rsum(11) + ap * rsum(12))) * inv_kmtot);
            step_cc = ap;
            auto commit = [&](double &s, double &l, double corr, double gdz, double viol) {
                const double sinv = fast_rcp(s);
                const double ds = -(viol + s) - gdz;
                const double rc = s * l - smu + corr;
                const double dl = (-rc - l * ds) * sinv;
                const double sn = s + ap * ds;
                double ln = l + ad * dl;
                if (ln * sn < fprod) ln = fprod * fast_rcp(sn);
                s = sn; l = ln;
            };
            hap = ap;
            if constexpr (IS_B) {
        const int halfp = opq(half);
#pragma unroll
        for (int r = RB0; r < RB1; r++) {
            FRP_RSB();
                    const int ib = r * H, i = ib + half;
                    if (i >= NZ) continue;
                    const double lb = ROW_PICK(lower_bound(i));
                    const double ub = ROW_PICK(upper_bound(i));
                    cldouble *rec = recs + k * RS; // (the step stays in the record until the next forward sweep)
                    const double zi = bz[r], dzi = rec[R_DZ + i];
                    commit(bsl[r], bll[r], bcl[r], -dzi, lb - zi);
                    commit(bsu[r], blu[r], bcu[r], dzi, zi - ub);
                    bz[r] = zi + ap * dzi;
                    if (ib < 8) bzp[r] += ap * rec[R_DZ + (i < 4 ? i + 4 : (i < 8 ? i - 4 : i))];
                }
            }
            if constexpr (IS_F) {
                {
                    cldouble *rec = recs + k * RS;
                    dzf[0] = rec[R_DZ + 8]; dzf[1] = rec[R_DZ + 9]; dzf[2] = rec[R_DZ + 10];
                }
                face_bases();
                for_faces([&](int t, double a0, double a1, double a2, double bb) __attribute__((always_inline)) {
                    commit(fs[t], fl_[t], PARK ? prk[FL + t] : fcr[t], a0 * dzf[0] + a1 * dzf[1] + a2 * dzf[2], a0 * fpq[0] + a1 * fpq[1] + a2 * fpq[2] - bb);
                });
                if constexpr (!FPZ) { fpos[0] += ap * dzf[0]; fpos[1] += ap * dzf[1]; fpos[2] += ap * dzf[2]; }
            }
            // (the iterate last: the corridor rows of the same wave take their violation from the position BEFORE the step)
            if constexpr (wave == 1) {
                if (kact) { // (the Newton step is valid until the next predictor's forward sweep, y+ until this wave's next model phase)
                    cldouble *rec = recs + k * RS;
#pragma unroll
                    for (int i = 0; i < NZ; i++) ms.z[i] += ap * rec[R_DZ + i];
#pragma unroll
                    for (int i = 0; i < NS; i++) ms.y[i] += ap * (rec[R_D + i] - ms.y[i]); // y <- y + ap (y+ - y)
                }
            }
        }
        if constexpr (TW) { tw_on = tw_on && tw_next; tw_it = tw_on ? tw_m : 0; } // (one way)
        it++;
    }

    // ---------------------------------------------------------------- outputs
#if FRP_EARLY_FACTOR
    if constexpr (wave == 0 && !TW) { // (the Riccati wave skipped the termination tests: what it reports it computes here, from the same partial results)
        nm.eq = uni(red(xs, WEQ, 0)); nm.in = uni(rmax(0)); nm.rc = uni(rmax(1)); nm.gap = uni(rsum(2));
        nm.rs = uni(stationarity_norm<NP>(recs, N));
        mu = uni(nm.gap * (KAPPA_LAM * inv_kmtot));
    }
#endif
    PROF_FLUSH(wave, it);
    __builtin_amdgcn_s_setprio(0);
    // the next problem is claimed only now (its latency hides behind the write-out): a slot that claimed it while it still
    // had a solve ahead of it would keep it from the slots that go idle at the end of the launch
    if constexpr (QW && wave == 0) { // (a long solve gives its CU back first)
        if (iso_mine && lane == 0) { atomicSub(a.cu_slots + sh.cukey, 256); atomicSub(a.counter + 1, 1); sh.ctl->head_first = -1; }
    }
    if (wave == 0 && lane == 0) sh.ctl->next = claim_next(a, sh.cukey); // (two dependent global round trips, under the model wave's write-out)
    if constexpr (wave == 1) {
        // the objective is reported, not iterated on: evaluated once, at the returned iterate
        double l_obj = 0.0;
        if (own0) {
            double *zo = a.z + ((size_t)b * N + k) * NZ;
#pragma unroll
            for (int i = 0; i < NZ; i++) zo[i] = ms.z[i];
            double p10[NPRE];
#pragma unroll
            for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
            l_obj = stage_cost(ms.z, p10, stage_class(k, N), model, nullptr);
#ifdef FRP_DEBUG_OBJ // per-stage cost instead of the yaw in the returned plan (diagnosis of code-generation variants)
            zo[16] = l_obj;
#endif
        }
        l_obj = wave_sum(l_obj);
        if (a.info && lane == 0) a.info[(size_t)b * FRP_INFO_STRIDE + 4] = l_obj;
    }
    if (wave == 0 && lane == 0) {
        a.exitflag[b] = flag;
        a.iters[b] = it;
        if (a.info) {
            double *o = a.info + (size_t)b * FRP_INFO_STRIDE;
            o[0] = nm.eq; o[1] = nm.in; o[2] = nm.rs; o[3] = nm.rc; o[5] = mu; o[6] = step_cc; o[7] = (double)nfallback; // o[4] = objective: wave 1
            o[8] = it > 0 ? xs[X_RED + 3 * 16 + 14] : 0.0; o[9] = it > 0 ? xs[X_RED + 3 * 16 + 15] : 0.0; o[10] = it > 0 ? xs[X_RED + 2 * 16 + 14] : 0.0;
            o[11] = nm.gap;
#ifdef FRP_PROFILE // tools/timeline.py: start and end of this solve on the 100 MHz wall clock instead of the last two fields
            o[6] = (double)pwall0_; o[7] = (double)wall_clock64();
#endif
        }
    }
}

// Persistent workgroups: grid = min(B, resident workgroups); each pulls the next problem index from a device counter
// (zeroed by the launcher) until the batch is exhausted.
#undef inv_kmtot
template <int NP, int FL, bool FREG, int ROLE, bool TW>
__device__ __forceinline__ void role_loop(const KernelArgs &a, const Shared &sh)
{
    // (solve_one claims the next problem when it leaves its iteration)
#ifndef FRP_STAGGER // experiment knob: the k-th workgroup of a CU starts its first solve k * FRP_STAGGER * 8 k cycles late (the four workgroups of a
#define FRP_STAGGER 0 // CU otherwise run the first round of a launch in lockstep -- 46 us per iteration against 41 later: profiles/r05_timeline.txt)
#endif
    if (QW && FRP_STAGGER > 0 && ROLE == 0 && sh.rsimd > 0)
        for (int q = 0; q < sh.rsimd * FRP_STAGGER; q++) __builtin_amdgcn_s_sleep(127);
    if (ROLE == 0 && (threadIdx.x & 63) == 0) {
        const int hf = QW ? sh.ctl->head_first : -1;
        if (hf >= 0) sh.ctl->next = a.order ? a.order[hf] : hf; // a head-start problem: this workgroup was first on its CU and has marked it
        else {
            // (a workgroup that arrived behind the CU's first one gives that one the ~2 us its two atomics need to mark the CU -- losing the race
            // costs nothing but the head start: the long solve then shares its CU with this workgroup's first solve)
            if (QW && a.head_start > 0 && sh.late) __builtin_amdgcn_s_sleep(100); // (~6 k cycles)
            sh.ctl->next = claim_next(a, sh.cukey);
        }
    }
    for (;;) {
        BAR();
        const int b = sh.ctl->next;
        if (b >= a.B) break;
        BAR(); // everybody has read the index before solve_one's exit overwrites it
        solve_one<NP, FL, FREG, ROLE, TW>(a, b, sh);
    }
}

#ifdef FRP_NUM_VGPR // experiment knob: a hard register cap for every kernel of the translation unit (the waves-per-EU attribute is relaxed by the compiler when the LDS size limits the occupancy anyway)
#define FRP_VGPR_ATTR __attribute__((amdgpu_num_vgpr(FRP_NUM_VGPR)))
#else
#define FRP_VGPR_ATTR
#endif
template <int NP, int FL, bool FREG, int WPE, bool TW>
__global__ __launch_bounds__(QW ? 192 : 256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) FRP_VGPR_ATTR void nmpc_ipm_lds_kernel(KernelArgs a)
{
    static_assert(!(QW && QL) || (size_t)(NP * RS + X_TOTAL + 3 * NP + 1) * 8 + sizeof(Ctl) + 5 * sizeof(int) <= 40960, "Q4: four workgroups per CU in 160 KB of LDS");
    static_assert(!QS || (size_t)(NP * RS + X_TOTAL + 1) * 8 + sizeof(Ctl) + 5 * sizeof(int) <= 53248, "Q30: three workgroups per CU in 160 KB of LDS");
#ifdef FRP_DYN_LDS // experiment knob (round 6): LDS sized at launch -- the compiler then cannot tell that the LDS limits the occupancy below WPE and keeps the register cap of WPE waves per SIMD
    extern __shared__ double s_dyn_[];
    double *s_recs = s_dyn_, *s_xs = s_recs + NP * RS, *s_tw = s_xs + X_TOTAL + (QS ? 0 : 3 * NP);
    Ctl &s_ctl = *reinterpret_cast<Ctl *>(s_tw + (TW ? TW_TOTAL : 1));
    int *s_place = reinterpret_cast<int *>(&s_ctl + 1);
#else
    __shared__ double s_recs[NP * RS];
    __shared__ double s_xs[X_TOTAL + (QS ? 0 : 3 * NP)];
    __shared__ double s_tw[TW ? TW_TOTAL : 1];
    __shared__ Ctl s_ctl;
    __shared__ int s_place[5];
#endif
    static_assert(!TW || NP == 20, "the twisted solve is built on the three-lanes-per-stage model phase");
    Shared sh;
    sh.recs = (ldouble *)s_recs; sh.xs = (ldouble *)s_xs; sh.tw = (ldouble *)s_tw; sh.ctl = &s_ctl; sh.cukey = -1; sh.late = 0; sh.rsimd = -1;
    if (TW && threadIdx.x == 0) { s_tw[TW_DUMP] = 0.0; s_tw[TW_ZERO] = 0.0; s_ctl.fail1 = 0; s_ctl.fail2 = 0; }
    if (threadIdx.x == 0) { s_xs[X_C0] = 0.0; s_xs[X_C1] = 1.0; s_ctl.dec_tag = 0xffffffffu; s_ctl.dec_stop = 0; s_ctl.head_first = -1; if constexpr (QW) s_place[3] = 1; }
    // ---- which wave plays which role.  A wavefront stays on the SIMD it was launched on, and one wavefront of every
    // resident workgroup sits on each SIMD of the CU.  The Riccati role keeps its SIMD busy for ~70 % of an iteration, the
    // three helper roles for 13-18 % each, so the iteration rate of a CU is set by the SIMD with the most work on it.  With
    // three workgroups per CU the roles are placed by SIMD: the k-th workgroup to arrive on a CU (a per-CU counter in the
    // workspace, zeroed by the launcher) runs its Riccati role on SIMD k, every workgroup runs its heaviest helper (the
    // model) on SIMD 3, which hosts no Riccati wave; the two light helpers share the SIMDs of the other two workgroups'
    // Riccati waves.  Anything unexpected (SIMDs not distinct, other residency) falls back to role = wave index.  (With two
    // workgroups per CU -- Riccati waves on SIMD 0 / 1, model waves on SIMD 2 / 3 -- the same idea LOSES 2.4 % on configs[3]
    // against the dispatcher's own rotation, so it is applied to the three-per-CU variants only.)
    const int widx = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int role = widx;
#ifndef FRP_NO_PLACE
#define FRP_NO_PLACE 0
#endif
    if constexpr (QW) {
        // Q4: four workgroups of three wavefronts per CU -- twelve waves, three per SIMD.  Every workgroup claims a SIMD for its Riccati
        // wave that no other workgroup's Riccati wave has taken (a bit mask per CU in the workspace, zeroed by the launcher), trying the
        // SIMDs its own waves sit on; its other two waves take the roles 1 and 2 in the cyclic order of their SIMDs behind the claimed
        // one, so that with the dispatcher's rotation {s, s+1, s+2} every SIMD ends up with one Riccati wave, one model + bounds wave
        // and one faces + bounds wave.  Anything unexpected (two waves on one SIMD, no free SIMD) falls back to role = wave index.
        if (a.cu_slots && !FRP_NO_PLACE) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
            const int simd = (int)((hw >> 4) & 3u);
            const unsigned key = ((xcc & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            sh.cukey = (int)__builtin_amdgcn_readfirstlane(key); // (every wave of a workgroup sits on the same CU)
            if ((threadIdx.x & 63) == 0) s_place[widx] = simd;
            __syncthreads();
            const int s0 = s_place[0], s1 = s_place[1], s2 = s_place[2];
            if (threadIdx.x == 0) {
                int rs = -1, first = 0;
                if (s0 != s1 && s0 != s2 && s1 != s2) {
                    const int cand[3] = {s0, s1, s2};
                    for (int q = 0; q < 3 && rs < 0; q++) {
                        const int old = atomicOr(a.cu_slots + key, 1 << cand[q]);
                        if (!(old & (1 << cand[q]))) rs = cand[q];
                        if (q == 0 && (old & 15) == 0) first = 1; // no workgroup had claimed a SIMD of this CU before this one
                    }
                }
                s_place[4] = rs;
                // head start: the first workgroup on a CU takes one of the longest expected solves and marks the CU at once (see claim_next:
                // the workgroups that arrive behind it find the mark and wait for the solve to end)
                int hf = -1;
                if (first && a.head_start > 0 && a.iso_it > 0) {
                    const int h = atomicAdd(a.counter + 2, 1);
                    if (h < a.head_start) { hf = h; atomicAdd(a.cu_slots + key, 256); atomicAdd(a.counter + 1, 1); }
                }
                s_ctl.head_first = hf;
                s_place[3] = first;
            }
            __syncthreads();
            const int rs = s_place[4];
            sh.late = s_place[3] ? 0 : 1;
            sh.rsimd = rs;
            if (rs >= 0) {
                // the other two waves, ordered by (simd - rs) mod 4
                const int da = (simd - rs) & 3;
                int dmin = 4;
                const int d0 = (s0 - rs) & 3, d1 = (s1 - rs) & 3, d2 = (s2 - rs) & 3;
                if (d0 && d0 < dmin) dmin = d0;
                if (d1 && d1 < dmin) dmin = d1;
                if (d2 && d2 < dmin) dmin = d2;
                role = da == 0 ? 0 : (da == dmin ? 1 : 2);
            }
#ifdef FRP_PROFILE
            if (threadIdx.x == 0) atomicAdd((unsigned long long *)&g_prof_seg[24 + (rs >= 0 ? rs : 4)], 1ull);
#endif
        }
    } else if (WPE >= 3 && a.cu_slots && !FRP_NO_PLACE) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID: SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID [3:0]
        const int simd = (int)((hw >> 4) & 3u);
        if ((threadIdx.x & 63) == 0) s_place[widx] = simd;
        if (threadIdx.x == 0) {
            const unsigned key = ((xcc & 7u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            s_place[4] = atomicAdd(a.cu_slots + key, 1);
        }
        __syncthreads();
        const int s0 = s_place[0], s1 = s_place[1], s2 = s_place[2], s3 = s_place[3], arrival = s_place[4];
        const bool distinct = ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15;
        if (distinct && arrival < 3) {
            const int rs = arrival; // SIMD of this workgroup's Riccati wave
            // the two SIMDs that are neither rs nor 3, in increasing order, take the roles 2 and 3
            int lo = -1;
            for (int q = 0; q < 3; q++)
                if (q != rs && lo < 0) lo = q;
#ifndef FRP_PLACE_CYCLIC
#define FRP_PLACE_CYCLIC 1
#endif
            // SIMD 3 hosts no Riccati wave: it takes the heaviest helper (the faces wave); model and bounds go to the other two
            // SIMDs, cyclically, so that every Riccati wave shares its SIMD with one of each
            const int s_model = FRP_PLACE_CYCLIC ? (rs + 1) % 3 : lo;
            role = simd == rs ? 0 : (simd == 3 ? 3 : (simd == s_model ? 1 : 2));
#ifndef FRP_TW_PLACE // (0: the placement of the plain solve)
#define FRP_TW_PLACE 1
#endif
            // twisted variants: the model wave runs the first half's sweeps, so IT is the heaviest helper and takes SIMD 3 (measured against
            // the plain placement, m = 9: 768 problems 468 -> 447 us, 1024: 631 -> 619; 4096 problems, m = 6: 1.021 -> 0.974 ms)
            if (TW && FRP_TW_PLACE) role = simd == rs ? 0 : (simd == 3 ? 1 : (simd == s_model ? 3 : 2));
        }
    }
    // one copy of the solver loop per role: the four waves run different code between the same barriers
    const int wave = __builtin_amdgcn_readfirstlane(role);
    if (wave == 0) {
        role_loop<NP, FL, FREG, 0, TW>(a, sh);
        // (single-problem launches of the drop-in context: this wave made the last claim; the head is zero again for the next call)
        if (a.self_reset && (threadIdx.x & 63) == 0) *a.counter = 0;
    } else if (wave == 1) role_loop<NP, FL, FREG, 1, TW>(a, sh);
    else if (wave == 2 || QW) role_loop<NP, FL, FREG, 2, TW>(a, sh);
    else {
        if constexpr (!QW) role_loop<NP, FL, FREG, 3, TW>(a, sh);
    }
    if (a.done_flag) { // (single-problem launches of the drop-in context: the host is spinning on this word)
        __threadfence_system(); // every wave's outputs (the plan: wave 1; flags and diagnostics: wave 0) before the word
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int NP, int FL, bool FREG, int WPE, bool TW = false>
static hipError_t launch_variant(const KernelArgs &k, int slots, hipStream_t stream)
{
#ifdef FRP_DYN_LDS
    const size_t lds = (size_t)(NP * RS + X_TOTAL + (QS ? 0 : 3 * NP) + (TW ? TW_TOTAL : 1)) * 8 + sizeof(Ctl) + 5 * sizeof(int) + 16;
    static bool once = [&] { return hipFuncSetAttribute(reinterpret_cast<const void *>(&nmpc_ipm_lds_kernel<NP, FL, FREG, WPE, TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; }();
    (void)once;
    hipLaunchKernelGGL((nmpc_ipm_lds_kernel<NP, FL, FREG, WPE, TW>), dim3(slots), dim3(QW ? 192 : 256), lds, stream, k);
#else
    hipLaunchKernelGGL((nmpc_ipm_lds_kernel<NP, FL, FREG, WPE, TW>), dim3(slots), dim3(QW ? 192 : 256), 0, stream, k);
#endif
    return hipGetLastError();
}
// the twisted variants: frp_nmpc_options.twist = m (stages eliminated forward), -1 = 9 N / 20 (3 N / 10 beyond 1024 problems); anything the twisted solve does not cover
// (N > 20, N < 4, m outside 2 .. N - 2) runs the plain solve, like the oracle's (oracle/nmpc_ipm.c, kkt_solve)
static inline int twist_stages(const KernelArgs &k)
{
    // -1: 9 N / 20 while the problems of the launch have a CU (nearly) to themselves; beyond ~1000 problems a shorter first half loses
    // less (4096 problems: +7.8 % at 9, +2.5 % at 6 of 20 -- the option is not meant for that regime)
    const int m = k.twist < 0 ? (k.B <= 1024 ? 9 * k.N / 20 : 3 * k.N / 10) : k.twist;
    return (k.N >= 4 && k.N <= 20 && m >= 2 && m <= k.N - 2) ? m : 0;
}

} // namespace lr / lrq

// Two translation units (build.py): the variants that keep the corridor rows in registers (FREG) are compiled with
// -amdgpu-use-amdgpu-trackers=1 (-4 % on the (20, 2) variant of the headline workload), the variants that re-read them from
// the parameters without it (the same flag costs them 2-10 %).  frp_ipm_lds_mem.hip includes this file with FRP_LDS_MEM_TU
// and contributes launch_ipm_lds_mem only; a build that defines neither macro (probes, experiments) gets everything here.
#if defined(FRP_LDS_Q4_TU)
// frp_ipm_lds_q4.hip: the four-problems-per-CU variants (three-wave workgroups, P in global memory); contributes launch_ipm_lds_q4 only
hipError_t launch_ipm_lds_q4(const KernelArgs &k, int slots, hipStream_t stream)
{
#ifdef FRP_Q4_MORE_ROWS // experiment (round 6): the four-per-CU form for up to 15 rows in registers / up to 30 rows re-read (the tick's 30-row stages)
    if (k.MF > 15) return FRP_LR::launch_variant<20, 10, false, 3>(k, slots, stream);
    if (k.MF > 6) return FRP_LR::launch_variant<20, 5, true, 3>(k, slots, stream);
#endif
    return FRP_LR::launch_variant<20, 2, true, 3>(k, slots, stream);
}
#elif defined(FRP_LDS_Q30_TU)
// frp_ipm_lds_q30.hip: N <= 30, <= 16 corridor rows, three problems per CU (215-double records, P in global memory); contributes launch_ipm_lds_q30 only
#ifndef FRP_Q30_FREG
#define FRP_Q30_FREG 1
#endif
hipError_t launch_ipm_lds_q30(const KernelArgs &k, int slots, hipStream_t stream)
{
    return FRP_LR::launch_variant<30, 8, (FRP_Q30_FREG != 0), 3>(k, slots, stream);
}
#elif defined(FRP_LDS_S2_TU)
// frp_ipm_lds_s2.hip: the small-launch variants (N <= 20, rows in registers, at most two problems per CU): the kernels of the main translation unit compiled for
// TWO wavefronts per SIMD -- in a translation unit of their own so that the sweeps, functions behind calls, get the 256-register budget too (a function takes the
// tightest budget among its callers); contributes launch_ipm_lds_s2 only
hipError_t launch_ipm_lds_s2(const KernelArgs &k, int slots, hipStream_t stream)
{
    // (more than 15 rows: ten per lane IN REGISTERS -- at 256 registers the corridor wave holds the constants of its rows, 140 registers of state, where the
    // three-per-CU build re-reads them from the parameters in every phase; 55 spilled.  Built for ONE wavefront per SIMD the kernel takes 316 registers, i.e. AGPR copies: slower, 0.156 ms)
    if (k.twist) return k.MF <= 6 ? FRP_LR::launch_variant<20, 2, true, 2, true>(k, slots, stream) : (k.MF <= 15 ? FRP_LR::launch_variant<20, 5, true, 2, true>(k, slots, stream) : FRP_LR::launch_variant<20, 10, true, 2, true>(k, slots, stream)); // (k.twist: resolved by launch_ipm_lds)
    return k.MF <= 6 ? FRP_LR::launch_variant<20, 2, true, 2>(k, slots, stream) : (k.MF <= 15 ? FRP_LR::launch_variant<20, 5, true, 2>(k, slots, stream) : FRP_LR::launch_variant<20, 10, true, 2>(k, slots, stream));
}
#elif defined(FRP_LDS_MEM_TU)
hipError_t launch_ipm_lds_mem(const KernelArgs &k, int slots, hipStream_t stream)
{
    if (k.N <= 20 && k.twist) return FRP_LR::launch_variant<20, 10, false, 3, true>(k, slots, stream); // (k.twist: resolved by launch_ipm_lds)
    if (k.N <= 20) return FRP_LR::launch_variant<20, 10, false, 3>(k, slots, stream);
    if (k.N <= 32) return FRP_LR::launch_variant<32, 15, false, 2>(k, slots, stream);
    return FRP_LR::launch_variant<64, 30, false, 1>(k, slots, stream);
}
#else
#ifdef FRP_PROFILE
void debug_read_prof_lds(long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(FRP_LR::g_prof_lds), sizeof(long long) * 64);
    (void)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(FRP_LR::g_prof_seg), sizeof(long long) * 32);
    long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(FRP_LR::g_prof_lds), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(FRP_LR::g_prof_seg), z, sizeof(long long) * 32);
}
#endif

// The Q4 variants (frp_ipm_lds_q4.hip) take the launches they cover: plain solve, N <= 20, at most six corridor rows per stage.
// FRP_Q4=0 in the environment keeps the three-per-CU variants (A/B runs).
#ifdef FRP_LDS_SPLIT_TU
hipError_t launch_ipm_lds_q4(const KernelArgs &k, int slots, hipStream_t stream); // frp_ipm_lds_q4.hip
hipError_t launch_ipm_lds_q30(const KernelArgs &k, int slots, hipStream_t stream); // frp_ipm_lds_q30.hip
hipError_t launch_ipm_lds_s2(const KernelArgs &k, int slots, hipStream_t stream);  // frp_ipm_lds_s2.hip
static bool q4_enabled()
{
    static const bool on = [] { const char *e = getenv("FRP_Q4"); return !(e && e[0] == '0'); }();
    return on;
}
static bool q30_enabled() // FRP_Q30=0 keeps the two-per-CU variants (A/B runs)
{
    static const bool on = [] { const char *e = getenv("FRP_Q30"); return !(e && e[0] == '0'); }();
    return on;
}
#else
static hipError_t launch_ipm_lds_q4(const KernelArgs &, int, hipStream_t) { return hipErrorInvalidValue; }
static hipError_t launch_ipm_lds_q30(const KernelArgs &, int, hipStream_t) { return hipErrorInvalidValue; }
static bool q4_enabled() { return false; } // (a single-translation-unit build has one record layout)
static bool q30_enabled() { return false; }
#endif
#if (defined(FRP_QP) || defined(FRP_QW)) && !defined(FRP_LDS_Q4_TU)
static bool q4_covers(const KernelArgs &) { return false; } // (bisection builds: the launch stays on this translation unit)
int lds_q4_max_rows() { return 6; }
#else
// ... and only launches with more problems than the three-per-CU variants hold at once: a problem that has a CU (nearly) to itself
// iterates faster on four wavefronts (69 k cycles per iteration against 80 k: profiles/r05_wave_phases.txt)
static int device_cus()
{
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}
static std::atomic<int> g_q4_min_b{[] { const char *e = getenv("FRP_Q4_MIN_B"); return e ? atoi(e) : -1; }()}; // (frp_nmpc_set_q4_min_batch, a process-wide tuning hook; -1: three workgroups per CU)
int lds_q4_max_rows() { static const int m = [] { const char *e = getenv("FRP_Q4_MAXF"); return e ? atoi(e) : 6; }(); return m; } // (experiment knob: needs a -DFRP_Q4_MORE_ROWS build of the Q4 unit)
static bool q4_covers(const KernelArgs &k)
{
    const int B = k.variant_B > 0 ? k.variant_B : k.B; // (the chunks of a host batch: the whole batch's variant)
    return q4_enabled() && k.pws && k.N <= 20 && k.MF <= lds_q4_max_rows() && FRP_LR::twist_stages(k) == 0 && B > (g_q4_min_b.load(std::memory_order_relaxed) >= 0 ? g_q4_min_b.load(std::memory_order_relaxed) : 3 * device_cus());
}
#endif

// The Q30 variant (frp_ipm_lds_q30.hip): 20 < N <= 30, at most 16 corridor rows per stage, plain solve, and a launch of at least ~2.3 rounds of its resident
// workgroups (7 x CUs problems): the 168-register build is 6 % slower per iteration than the 243-register one, and measured on configs[3] the third slot per CU
// pays from B = 2048 on (B = 512 / 1024 / 2048 / 4096 / 16384: 1.72 / 2.18 / 2.56 / 3.48 / 11.09 ms against 1.63 / 2.06 / 2.58 / 4.05 / 13.90 at two per CU)
#if (defined(FRP_QP) || defined(FRP_QW)) && !defined(FRP_LDS_Q4_TU)
static bool q30_covers(const KernelArgs &) { return false; }
#else
static bool q30_covers(const KernelArgs &k)
{
    const int B = k.variant_B > 0 ? k.variant_B : k.B;
    static const int env_b = [] { const char *e = getenv("FRP_Q30_MIN_B"); return e ? atoi(e) : -1; }();
    const int set_b = g_q4_min_b.load(std::memory_order_relaxed); // (frp_nmpc_set_q4_min_batch moves the threshold of BOTH high-residency variant sets: the tests send every covered launch here)
    const int min_b = set_b >= 0 ? set_b : env_b;
    return q30_enabled() && k.pws && k.N > 20 && k.N <= 30 && k.MF <= 16 && FRP_LR::twist_stages(k) == 0 && B > (min_b >= 0 ? min_b : 7 * device_cus());
}
#endif
// Small launches (N <= 20, rows in registers, at most two problems per CU -- the drop-in call is the case of ONE): the same kernels compiled for two wavefronts per
// SIMD, i.e. 256 registers instead of 168 -- no spills, nothing to fetch back on the chain: 0.1260 -> 0.1226 ms per drop-in call, 0.1142 -> 0.1095 with the twisted
// solve (FRP_SMALL2=0 switches it off).  The choice goes by the size of the whole batch (variant_B), like that of the high-residency variants.
static bool small2_covers(const KernelArgs &k)
{
#ifndef FRP_LDS_SPLIT_TU
    return false; // (a build without the other translation units has no such variants)
#endif
    static const int env = [] { const char *e = getenv("FRP_SMALL2"); return e ? atoi(e) : 1; }();
    const int B = k.variant_B > 0 ? k.variant_B : k.B;
    static const int max_b = [] { const char *e = getenv("FRP_SMALL2_MAXB"); return e ? atoi(e) : 0; }(); // (experiments: the two-per-CU builds for larger batches too)
    return env != 0 && k.N <= 20 && k.MF <= 30 && B <= (max_b > 0 ? max_b : 2 * device_cus()) && !q4_covers(k);
}
// workgroups resident per CU: LDS-bound (4 x 40 KB on the Q4 variants; 3 x 51 KB, 3 x 52 KB on the Q30 variant, 2 x 79 KB, 1 x 157 KB), two by registers on the small-launch variants
int lds_workgroups_per_cu(const KernelArgs &k) { return k.N <= 20 ? (q4_covers(k) ? 4 : (small2_covers(k) ? 2 : 3)) : (k.N <= 32 ? (q30_covers(k) ? 3 : 2) : 1); }
bool lds_q30_enabled() { return q30_enabled(); }
size_t lds_q30_pws_doubles_per_slot() { return (size_t)30 * FRP_LR::PG; }
// doubles of packed-P workspace a resident workgroup of the Q4 variants needs (KernelArgs::pws)
size_t lds_q4_pws_doubles_per_slot() { return (size_t)20 * FRP_LR::PG; }
bool lds_q4_enabled() { return q4_enabled(); }
#if (defined(FRP_QP) || defined(FRP_QW)) && !defined(FRP_LDS_Q4_TU)
int lds_q4_set_min_batch(int) { return -1; }
#else
int lds_q4_set_min_batch(int min_b) { return g_q4_min_b.exchange(min_b < 0 ? -1 : min_b); }
#endif

bool lds_kernel_supports(int N, int MF) { return N >= 1 && N <= 64 && MF >= 0 && MF <= FRP_MAX_FACES; }

#ifdef FRP_LDS_SPLIT_TU
hipError_t launch_ipm_lds_mem(const KernelArgs &k, int slots, hipStream_t stream); // frp_ipm_lds_mem.hip
#else
static hipError_t launch_ipm_lds_mem(const KernelArgs &k, int slots, hipStream_t stream)
{
    if (k.N <= 20 && k.twist) return FRP_LR::launch_variant<20, 10, false, 3, true>(k, slots, stream);
    if (k.N <= 20) return FRP_LR::launch_variant<20, 10, false, 3>(k, slots, stream);
    if (k.N <= 32) return FRP_LR::launch_variant<32, 15, false, 2>(k, slots, stream);
    return FRP_LR::launch_variant<64, 30, false, 1>(k, slots, stream);
}
#endif

#ifndef FRP_WPE20 // experiment knob: register budget of the N <= 20 variants (3 waves per SIMD = 168 VGPRs; 4 = 128)
#define FRP_WPE20 3
#endif
// counter / order already set up by launch_ipm
hipError_t launch_ipm_lds(const KernelArgs &k0, int slots, hipStream_t stream)
{
    KernelArgs k = k0;
    k.twist = FRP_LR::twist_stages(k0);
    const int MF = k.MF;
    if (q4_covers(k0)) return launch_ipm_lds_q4(k, slots, stream);
    if (q30_covers(k0)) return launch_ipm_lds_q30(k, slots, stream);
#if (defined(FRP_QP) || defined(FRP_QW)) && !defined(FRP_LDS_Q4_TU) // bisection builds (parts of Q4 on the main translation unit): one variant
    return (k.N <= 20 && MF <= 6 && !k.twist && k.pws) ? FRP_LR::launch_variant<20, 2, true, FRP_WPE20>(k, slots, stream) : hipErrorInvalidValue;
#else
#ifdef FRP_LDS_SPLIT_TU
    if (small2_covers(k0)) return launch_ipm_lds_s2(k, slots, stream);
#endif
    if (k.N <= 20 && k.twist) {
        if (MF <= 6) return FRP_LR::launch_variant<20, 2, true, FRP_WPE20, true>(k, slots, stream);
        if (MF <= 15) return FRP_LR::launch_variant<20, 5, true, FRP_WPE20, true>(k, slots, stream);
    } else if (k.N <= 20) {
        if (MF <= 6) return FRP_LR::launch_variant<20, 2, true, FRP_WPE20>(k, slots, stream);
        if (MF <= 15) return FRP_LR::launch_variant<20, 5, true, FRP_WPE20>(k, slots, stream);
    } else if (k.N <= 32) {
        if (MF <= 6) return FRP_LR::launch_variant<32, 3, true, 2>(k, slots, stream);
#ifndef FRP_N32_WPE // experiment knobs (round 6): register budget / rows in registers of the N <= 32, <= 16-row variant -- the proxy for three problems per CU
#define FRP_N32_WPE 2
#endif
#ifndef FRP_N32_FREG
#define FRP_N32_FREG 1
#endif
        if (MF <= 16) return FRP_LR::launch_variant<32, 8, (FRP_N32_FREG != 0), FRP_N32_WPE>(k, slots, stream);
    } else if (MF <= 8) return FRP_LR::launch_variant<64, 8, true, 1>(k, slots, stream);
    return launch_ipm_lds_mem(k, slots, stream);
#endif
}
#endif // FRP_LDS_MEM_TU

} // namespace frp

#if defined(FRP_PROFILE) && !defined(FRP_LDS_MEM_TU) && !defined(FRP_LDS_Q4_TU) && !defined(FRP_LDS_Q30_TU) && !defined(FRP_LDS_S2_TU)
extern "C" void frp_debug_read_prof_lds(long long *out) { frp::debug_read_prof_lds(out); }
#endif
#if defined(FRP_PROFILE) && defined(FRP_LDS_Q4_TU)
// (the Q4 variants too; slots 24..28 of the segment counters: workgroups whose Riccati wave claimed SIMD 0..3 / fell back to role = wave index)
extern "C" void frp_debug_read_prof_lds_q4(long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(frp::FRP_LR::g_prof_lds), sizeof(long long) * 64);
    (void)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(frp::FRP_LR::g_prof_seg), sizeof(long long) * 32);
    long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_lds), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_seg), z, sizeof(long long) * 32);
}
#endif
#if defined(FRP_PROFILE) && defined(FRP_LDS_Q30_TU)
extern "C" void frp_debug_read_prof_lds_q30(long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(frp::FRP_LR::g_prof_lds), sizeof(long long) * 64);
    (void)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(frp::FRP_LR::g_prof_seg), sizeof(long long) * 32);
    long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_lds), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_seg), z, sizeof(long long) * 32);
}
#endif
#if defined(FRP_PROFILE) && defined(FRP_LDS_MEM_TU) && !defined(FRP_LDS_Q4_TU)
// (the re-reading variants are a translation unit of their own, with their own copy of the profile counters)
extern "C" void frp_debug_read_prof_lds_mem(long long *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(frp::FRP_LR::g_prof_lds), sizeof(long long) * 64);
    (void)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(frp::FRP_LR::g_prof_seg), sizeof(long long) * 32);
    long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_lds), z, sizeof z);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(frp::FRP_LR::g_prof_seg), z, sizeof(long long) * 32);
}
#endif
