// frp_kernels.h -- internal interface between the C-ABI layer (frp_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace frp {

// per-stage HBM record (doubles): everything the serial Riccati chain streams, laid out so that one
// wavefront reads it with 64-lane coalesced loads.
constexpr int REC_LIN = 0;      // compact linearisation (51): Apv Ape Avv Ave BpT BvT Bvw
constexpr int REC_D = 51;       // d = prev(z_k) - s_{k+1}  (13)
constexpr int REC_KB = 64;      // Kb = Quu^-1 Qus (4 x 13)
constexpr int REC_R = 116;      // R = Quu^-1 (4 x 4)
constexpr int REC_PD = 132;     // P_{k+1} d (13)
constexpr int REC_KV = 145;     // kb = R q_u (4)
constexpr int REC_PV = 149;     // p_k (13)
constexpr int REC_PHID = 162;   // diag of Phi = H + barrier (17)
constexpr int REC_PHIPOS = 179; // corridor barrier block on pos (3 x 3)
constexpr int REC_PHI = 188;    // rhs gradient phi (17)
constexpr int REC_HC = 205;     // (u_i, w_i) cost coupling -2 w_rate of this stage
constexpr int REC_STRIDE = 208;

constexpr double S_MIN = 1e-2;          // smallest initial slack (infeasible start shift)
constexpr double MU_FLOOR_FRAC = 0.1;   // centring target floor = 0.1 * tol_comp
constexpr double DIVERGE_MU = 1e6;
constexpr double DIVERGE_RS = 1e12;

struct KernelArgs {
    int B, N, M, MF, model, maxit;
    double tol_stat, tol_eq, tol_ineq, tol_comp, mu0, ftb;
    const double *xinit, *x0, *params;
    const int *nfaces;
    double *z;
    int *exitflag, *iters;
    double *info;
    double *ws;
};

size_t ws_bytes(int B, int N, int MF);
hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream);
hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream);

} // namespace frp
