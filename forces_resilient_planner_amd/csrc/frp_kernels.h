// frp_kernels.h -- internal interface between the C-ABI layer (frp_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace frp {

// Per-stage HBM record (doubles): everything the serial Riccati sweeps stream, laid out so that one
// wavefront moves it with 64-lane coalesced loads/stores.
//   E part (written by the evaluation / step phases, 192 doubles = 3 x 64):
constexpr int REC_LIN = 0;      // compact linearisation (51): Apv Ape Avv Ave BpT BvT Bvw
constexpr int REC_D = 51;       // d = prev(z_k) - s_{k+1}, s-order [w; x]  (13)
constexpr int REC_PHID = 64;    // diag of Phi = cost Hessian + bound barriers (17)
constexpr int REC_PHIPOS = 81;  // corridor barrier block on pos (3 x 3)
constexpr int REC_PHI = 90;     // predictor rhs gradient phi_aff (17)
constexpr int REC_HC = 107;     // (u_i, w_i) cost coupling -2 w_rate of this stage
constexpr int REC_PHIB = 108;   // corrector rhs: phi_cc = PHIB + (sigma mu) PHIC  (17 + 17)
constexpr int REC_PHIC = 125;
constexpr int REC_ZERO = 142;   // a slot of the record that always holds 0.0 (target of masked gathers)
constexpr int REC_HD = 143;     // exact Hessian of y'c(z) over (rates, T, v, e): the 45 structurally non-zero
                                // entries of the upper triangle, see hd_pack()
constexpr int REC_HD_SIZE = 45;
constexpr int REC_E_SIZE = 192; // 188..191 pad
//   F part (written by the factorisation sweep):
constexpr int REC_T = 192;      // T' = [R | Kbar_x | kbar | hc] as tile register 0 (64): lane (g,c) <-> T'[g][c]
constexpr int REC_PD = 256;     // P_{k+1} d (16, s-order rows)                                   } written as one contiguous
constexpr int REC_P = 272;      // P_k, packed lower triangle: (row, col <= row) -> row (row + 1) / 2 + col  (91, padded 96) } 112-double run
constexpr int REC_PV = 368;     // p_k of the corrector solve (16, s-order rows): [P | p] is read as one 112-double run
constexpr int REC_STRIDE = 384;

// Packed index of entry (i, j) of the 10 x 10 dynamics Hessian over (rates 0..2, T 3, v 4..6, e 7..9), or -1 where
// it is structurally zero: the acceleration is affine in (T, v) jointly, so the (T,T), (T,v) and (v,v) blocks vanish.
__host__ __device__ constexpr bool hd_zero(int i, int j) { return i >= 3 && i <= 6 && j >= 3 && j <= 6; }
__host__ __device__ constexpr int hd_pack(int i, int j)
{
    if (i > j) { const int t = i; i = j; j = t; }
    if (hd_zero(i, j)) return -1;
    int n = 0;
    for (int a = 0; a < 10; a++)
        for (int b = a; b < 10; b++) {
            if (a == i && b == j) return n;
            if (!hd_zero(a, b)) n++;
        }
    return -1;
}
static_assert(hd_pack(9, 9) == REC_HD_SIZE - 1, "45 structurally non-zero entries");
struct HdTable {
    int v[100];
};
constexpr HdTable make_hd_table()
{
    HdTable t{};
    for (int a = 0; a < 10; a++)
        for (int b = 0; b < 10; b++) t.v[a * 10 + b] = hd_pack(a, b);
    return t;
}
static constexpr HdTable HD_TABLE = make_hd_table(); // folded to immediates once the callers are unrolled
__host__ __device__ inline int hd_index(int i, int j) { return HD_TABLE.v[i * 10 + j]; }
constexpr int DZ_ROWS = 20;     // dz rows: du(4) + ds(13) + 3 pad rows (tile rows 13..15)
constexpr int Y_ROWS = 16;      // y rows: 13 + 3 pad rows

constexpr double S_MIN = 1e-2;          // smallest initial slack (infeasible start shift)
constexpr double MU_FLOOR_FRAC = 0.3;   // centring target floor = 0.3 * tol_comp
constexpr double THETA_DOWN = 0.25;     // dynamics-Hessian weight after an indefinite pivot block ...
constexpr double THETA_UP = 0.1;        // ... and its recovery per successful iteration
constexpr double KAPPA_LAM = 2.0;       // multiplier safeguard: s_i lam_i >= mu / KAPPA_LAM after every step
constexpr double DIVERGE_MU = 10.0;     // mu > 10 max(1, mu0): a (locally) infeasible instance, exit -7 (see oracle/nmpc_ipm.c)
constexpr double DIVERGE_RS = 1e12;

struct KernelArgs {
    int B, N, M, MF, model, maxit, hessian;
    double tol_stat, tol_eq, tol_ineq, tol_comp, mu0, ftb;
    const double *xinit, *x0, *params;
    const int *nfaces;
    double *z;
    int *exitflag, *iters;
    double *info;
    double *ws;
    int *counter;     // work-queue head (set by the launcher)
    int *cu_slots;    // per-CU arrival counters of the resident workgroups, zeroed by the launcher (frp_ipm_lds.hip: role placement)
    const int *order; // launch order of the problems, or null = index order (set by the launcher)
    const int *models; // per-problem FRP_MODEL_*, or null = `model` for the whole batch
};

constexpr int CU_SLOT_ENTRIES = 2048; // (XCC, SE, SH, CU) of HW_ID
size_t ws_bytes(int B, int N, int MF);
hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream);
// frp_ipm_lds.hip: the LDS-resident kernel (queue counter / order already set up by launch_ipm)
int lds_workgroups_per_cu(int N);
bool lds_kernel_supports(int N, int MF);
hipError_t launch_ipm_lds(const KernelArgs &k, int slots, hipStream_t stream);
hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream);

} // namespace frp
