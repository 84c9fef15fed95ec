// frp_kernels.h -- internal interface between the C-ABI layer (frp_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace frp {

constexpr int REC_HD_SIZE = 45; // exact Hessian of y'c(z) over (rates, T, v, e): structurally non-zero entries of the upper triangle, see hd_pack()

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
constexpr int FRP_MAX_FACES = 30; // live corridor rows per stage the solver kernels hold (the reference's num_const: rows beyond 30 are dropped by its adapter, forces_normal.cpp:114)

constexpr double S_MIN = 1e-2;          // smallest initial slack (infeasible start shift)
constexpr double MU_FLOOR_FRAC = 0.3;   // centring target floor = 0.3 * tol_comp
constexpr double THETA_DOWN = 0.25;     // dynamics-Hessian weight after an indefinite pivot block ...
constexpr double THETA_UP = 0.1;        // ... and its recovery per successful iteration
constexpr double KAPPA_LAM = 2.0;       // multiplier safeguard: s_i lam_i >= mu / KAPPA_LAM after every step
constexpr double DIVERGE_RS = 1e12;

struct KernelArgs {
    int B, N, M, MF, model, maxit, hessian, twist;
    double tol_stat, tol_eq, tol_ineq, tol_comp, mu0, ftb, diverge_mu;
    const double *xinit, *x0, *params;
    const int *nfaces;
    double *z;
    int *exitflag, *iters;
    double *info;
    double *ws;       // queue workspace (ws_bytes): counter, per-CU counters, keys, order, packed-P blocks of the Q4 variants
    double *pws;      // packed P_k of the resident workgroups (Q4 variants: 20 x 92 doubles each; set by the launcher, null = not available)
    int *counter;     // work-queue head (set by the launcher); counter[1]: CUs a long solve has to itself right now
    int head_start;   // (set by the launcher; Q4 variants) the first `head_start` problems of the launch order -- the longest expected solves -- are taken by the
                      // first workgroup to arrive on a CU, which keeps that CU to itself from the start (counter[2] deals them out; 0 = none)
    int iso_it, iso_cap; // (set by the launcher) a solve that reaches iteration iso_it claims its CU (0 = never); at most iso_cap CUs at a time
    int *cu_slots;    // per-CU arrival counters of the resident workgroups, zeroed by the launcher (frp_ipm_lds.hip: role placement)
    const int *order; // launch order of the problems, or null = index order (set by the launcher)
    const int *models; // per-problem FRP_MODEL_*, or null = `model` for the whole batch
    const int *order_hint; // per-problem expected work (last tick's iteration count), or null = order by the cost of the initial guess
    int self_reset;        // B == 1 only (the drop-in context): the queue head is zero on entry and the kernel leaves it zero --
                           // no reset launch in front of the solve; no role placement (one workgroup: nothing to place)
    int variant_B;         // > 0: the kernel variant (three / four problems per CU) is chosen as for a launch of this many problems -- the chunks
                           // of frp_nmpc_solve_batch_host all run the whole batch's variant (they sum in another order); 0 = by B
    int slot_reserve;      // resident workgroups the launch leaves free (0 = none): the pipelined host path keeps room for the NEXT batch's
                           // gather kernel beside the persistent solver workgroups (frp_nmpc_solve_batch_host_begin)
    int *done_flag;        // B == 1 only, or null: a word in host-coherent memory that receives done_seq once every output of the solve is
    int done_seq;          // visible to the host -- the drop-in call spins on it instead of paying a stream synchronisation (~10 us)
};

constexpr int CU_SLOT_ENTRIES = 2048; // (XCC, SE, SH, CU) of HW_ID
size_t ws_bytes(int B, int N, int MF);
hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream);
// measurement hook: event pairs around the dominant kernel of the launches between begin and end (frp_nmpc_kernel_timing_*)
hipError_t kernel_timing_begin(int max_launches, int stride);
hipError_t kernel_timing_end(float *avg_ms, int *launches);
// frp_ipm_lds.hip: the LDS-resident kernel (queue counter / order already set up by launch_ipm)
int lds_workgroups_per_cu(const KernelArgs &k);
size_t lds_q4_pws_doubles_per_slot();
bool lds_q4_enabled();
int lds_q4_max_rows();                 // corridor rows per stage the four-per-CU variants take (6; experiment knob FRP_Q4_MAXF)
bool lds_q30_enabled();                // (frp_ipm_lds_q30.hip: 20 < N <= 30 at three problems per CU)
size_t lds_q30_pws_doubles_per_slot();
int lds_q4_set_min_batch(int min_b); // (frp_nmpc_set_q4_min_batch)
bool lds_kernel_supports(int N, int MF);
hipError_t launch_ipm_lds(const KernelArgs &k, int slots, hipStream_t stream);
hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream);

} // namespace frp
