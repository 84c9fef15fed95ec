// frp_pack.hip -- SURVEY 8f row f-1: the reference adapter's parameter packing and result bookkeeping, batched on the
// device, so that a receding-horizon loop over B planners never leaves HBM.
//
//   pack_kernel     FORCESNormal::solveNormal up to the solver call (plan_manage/src/forces_normal.cpp:55-136;
//                   forces_final.cpp:55-135 is identical) + setParasNormal (:36-52):
//                     xinit = mpc_output row 1 state, x0 = rows 1..N (shifted warm start),
//                     all_parameters row i = [ref_pos(3) f_ext(3) w_wp w_in w_rate yaw_ref | A (M x 3) | b - ||E a||_2 (M)],
//                     rows beyond the polytope's face count (and beyond num_const) zero-padded.
//   update_kernel   FORCESNormal::updateNormal (:142-168) + NMPCSolver::updateFORCESResults
//                   (plan_manage/src/nmpc_solver.cpp:524-543): rows 0..N-1 <- solver output, yaw wrapped into
//                   [-pi, pi], row N <- row N-1.
//
// Pure HBM-bound byte/FP64 work: one wavefront per (problem, stage) row, lanes over the row's columns, so every
// store is a coalesced 512-byte run.  Algorithmic bytes per problem (N = 20, M = 30, F stored faces per stage):
//   pack   read  8 (21*17 + 3 + 20*3 + 20 + 20*9 + 20*F*4) + 4*20    write 8 (9 + 340 + 2600) + 4*20
//   update read  8*340                                                 write 8*357
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/frp_nmpc.h"

namespace frp {

constexpr int PK_NZ = 17, PK_NPRE = 10;

__global__ __launch_bounds__(64) void pack_kernel(frp_nmpc_pack p)
{
    const int row = blockIdx.x;            // (problem, stage)
    const int b = row / p.N, i = row % p.N;
    const int lane = threadIdx.x;
    const int np = PK_NPRE + 4 * p.M;
    const double *mo = p.mpc_output + (size_t)b * (p.N + 1) * PK_NZ;
    // x0 row i = plan row i + 1 (forces_normal.cpp:74-97); xinit = state part of plan row 1 (:62-72)
    if (lane < PK_NZ) p.x0[((size_t)b * p.N + i) * PK_NZ + lane] = mo[(i + 1) * PK_NZ + lane];
    if (i == 0 && lane < 9) p.xinit[(size_t)b * 9 + lane] = mo[PK_NZ + 8 + lane];
    // this stage's polytope (poly_constraints[poly_indices(i)], forces_normal.cpp:112) and tube matrix E_i
    const int pi = p.poly_index ? p.poly_index[(size_t)b * p.N + i] : i;
    const size_t pbase = (size_t)b * p.NPOLY + pi;
    const double *A = p.poly_A + pbase * p.F * 3, *bb = p.poly_b + pbase * p.F;
    int nf = p.poly_nfaces[pbase];
    nf = nf < p.M ? nf : p.M;              // faces beyond num_const are dropped (:114)
    nf = nf < p.F ? nf : p.F;
    const double *E = p.ellipsoid + ((size_t)b * p.N + i) * 9;
    double *out = p.params + ((size_t)b * p.N + i) * np;
    const bool last = i == p.N - 1;
    for (int c = lane; c < np; c += 64) {
        double v = 0.0;
        if (c < 3) v = p.ref_pos[((size_t)b * p.N + i) * 3 + c];                         // :99-102
        else if (c < 6) v = p.external_acc[(p.external_acc_per_stage ? (size_t)b * p.N + i : (size_t)b) * 3 + c - 3]; // :103-106
        else if (c == 6) v = last ? p.w_terminal_wp : p.w_stage_wp;                      // :36-52
        else if (c == 7) v = last ? p.w_terminal_input : p.w_stage_input;
        else if (c == 8) v = p.w_input_rate;
        else if (c == 9) v = p.ref_yaw[(size_t)b * p.N + i];                             // :107-108
        else if (c < PK_NPRE + 3 * p.M) {                                                // A row-major (:116-123)
            const int j = (c - PK_NPRE) / 3;
            v = j < nf ? A[c - PK_NPRE] : 0.0;
        } else {                                                                         // b_j - ||E a_j||_2 (:124-125)
            const int j = c - PK_NPRE - 3 * p.M;
            if (j < nf) {
                const double a0 = A[3 * j], a1 = A[3 * j + 1], a2 = A[3 * j + 2];
                double n2 = 0.0;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double t = E[3 * r] * a0 + E[3 * r + 1] * a1 + E[3 * r + 2] * a2;
                    n2 += t * t;
                }
                v = bb[j] - sqrt(n2);
            }
        }
        out[c] = v;
    }
    if (lane == 0) p.nfaces[(size_t)b * p.N + i] = nf;
}

__global__ __launch_bounds__(64) void update_kernel(int B, int N, const double *__restrict__ z, const int *__restrict__ exitflag,
                                                    double *__restrict__ mpc_output)
{
    const int row = blockIdx.x;            // (problem, plan row 0..N)
    const int b = row / (N + 1), i = row % (N + 1);
    const int lane = threadIdx.x;
    if (lane >= PK_NZ) return;
    if (exitflag && exitflag[b] != FRP_EXIT_OPTIMAL) return; // the reference keeps the old plan when the solver fails (nmpc_solver.cpp:392-420)
    const int src = i < N ? i : N - 1;     // row N duplicates row N-1 (nmpc_solver.cpp:543)
    double v = z[((size_t)b * N + src) * PK_NZ + lane];
    if (lane == 16) {                      // yaw wrap (nmpc_solver.cpp:527-541)
        const double PI = 3.14159265358979323846;
        if (v < -PI) v += 2 * PI;
        else if (v > PI) v -= 2 * PI;
    }
    mpc_output[((size_t)b * (N + 1) + i) * PK_NZ + lane] = v;
}

} // namespace frp

extern "C" int frp_nmpc_pack_batch(const frp_nmpc_pack *p, void *stream)
{
    if (!p || p->B <= 0 || p->N < 2 || p->M < 0 || p->F <= 0 || p->NPOLY <= 0 || !p->mpc_output || !p->external_acc || !p->ref_pos ||
        !p->ref_yaw || !p->ellipsoid || !p->poly_A || !p->poly_b || !p->poly_nfaces || !p->xinit || !p->x0 || !p->params || !p->nfaces)
        return FRP_ERR_ARG;
    if (!p->poly_index && p->NPOLY != p->N) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::pack_kernel, dim3((unsigned)((size_t)p->B * p->N)), dim3(64), 0, static_cast<hipStream_t>(stream), *p);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

extern "C" int frp_nmpc_update_batch(int B, int N, const double *z, const int *exitflag, double *mpc_output, void *stream)
{
    if (B <= 0 || N < 2 || !z || !mpc_output) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::update_kernel, dim3((unsigned)((size_t)B * (N + 1))), dim3(64), 0, static_cast<hipStream_t>(stream), B, N, z,
                       exitflag, mpc_output);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}
