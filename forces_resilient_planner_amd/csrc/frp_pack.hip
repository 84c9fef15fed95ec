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
// Pure HBM-bound byte/FP64 work: one 256-thread workgroup per problem, threads over the problem's flattened output
// elements, so every store instruction of a wave is a coalesced 512-byte run.  Algorithmic bytes per problem (N = 20, M = 30, F stored faces per stage):
//   pack   read  8 (21*17 + 3 + 20*3 + 20 + 20*9 + 20*F*4) + 4*20    write 8 (9 + 340 + 2600) + 4*20
//   update read  8*340                                                 write 8*357
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/frp_nmpc.h"

namespace frp {

constexpr int PK_NZ = 17, PK_NPRE = 10;

// One workgroup per problem: its 2600 + 340 + 9 output doubles are written as contiguous 2 KB runs.
__global__ __launch_bounds__(256) void pack_kernel(frp_nmpc_pack p)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int np = PK_NPRE + 4 * p.M;
    const double *mo = p.mpc_output + (size_t)b * (p.N + 1) * PK_NZ;
    // x0 rows 0..N-1 = plan rows 1..N (forces_normal.cpp:74-97): one contiguous copy; xinit = state of plan row 1 (:62-72)
    for (int e = tid; e < p.N * PK_NZ; e += 256) p.x0[(size_t)b * p.N * PK_NZ + e] = mo[PK_NZ + e];
    if (tid < 9) p.xinit[(size_t)b * 9 + tid] = mo[PK_NZ + 8 + tid];
    // per stage: its polytope (poly_constraints[poly_indices(i)], forces_normal.cpp:112) and live face count
    // s_lim: rows of the stage this call has to write -- all M, or (frp_nmpc_pack.padded_rows_are_zero) the live rows and those the PREVIOUS call on these buffers left
    // live: everything beyond is zero already.  A Monte-Carlo fleet with six-face corridors writes 34 of the 130 slots of a stage instead of all of them.
    __shared__ int s_pi[64], s_nf[64], s_lim[64];
    if (tid < p.N) {
        int pi = p.poly_index ? p.poly_index[(size_t)b * p.N + tid] : tid;
        pi = pi < 0 ? 0 : (pi < p.NPOLY ? pi : p.NPOLY - 1); // a corrupt index must not read outside the polytope arrays
        int nf = p.poly_nfaces[(size_t)b * p.NPOLY + pi];
        nf = nf < p.M ? nf : p.M;              // faces beyond num_const are dropped (:114)
        nf = nf < p.F ? nf : p.F;
        nf = nf > 0 ? nf : 0;
        int lim = p.M;
        if (p.padded_rows_are_zero) {
            int old = p.nfaces[(size_t)b * p.N + tid]; // (what the previous call stored; clamped: a caller that set the flag on fresh buffers must not make this kernel skip a row it owes)
            old = old < 0 ? p.M : (old > p.M ? p.M : old);
            lim = nf > old ? nf : old;
        }
        s_pi[tid] = pi; s_nf[tid] = nf; s_lim[tid] = lim;
        p.nfaces[(size_t)b * p.N + tid] = nf;
    }
    __syncthreads();
    int lmax = 0;
    for (int i = 0; i < p.N; i++) lmax = s_lim[i] > lmax ? s_lim[i] : lmax;
    const bool fin = p.mode && p.mode[b] == FRP_MODEL_FINAL; // this planner runs the final solver: setParasFinal's weights
    const double w_wp = fin ? p.wf_stage_wp : p.w_stage_wp, w_in = fin ? p.wf_stage_input : p.w_stage_input, w_rate = fin ? p.wf_input_rate : p.w_input_rate;
    const double wt_wp = fin ? p.wf_terminal_wp : p.w_terminal_wp, wt_in = fin ? p.wf_terminal_input : p.w_terminal_input;
    // Three loops with uniform control flow instead of one if-chain over the 130 slots of a stage: a wavefront of 64 consecutive
    // slots straddled the header, the A block and the b block, so it executed all three bodies -- the norm of the b rows included --
    // for every element (0.064 ms per 4096 planners, 0.23 of the HBM roofline; round 4).
    double *out = p.params + (size_t)b * p.N * np;
    // (1) the ten leading slots of every stage (forces_normal.cpp:36-52, 99-108)
    for (int e = tid; e < p.N * PK_NPRE; e += 256) {
        const int i = e / PK_NPRE, c = e - i * PK_NPRE;
        const bool last = i == p.N - 1;
        double v;
        if (c < 3) v = p.ref_pos[((size_t)b * p.N + i) * 3 + c];
        else if (c < 6) v = p.external_acc[(p.external_acc_per_stage ? (size_t)b * p.N + i : (size_t)b) * 3 + c - 3];
        else if (c == 6) v = last ? wt_wp : w_wp;
        else if (c == 7) v = last ? wt_in : w_in;
        else if (c == 8) v = w_rate;
        else v = p.ref_yaw[(size_t)b * p.N + i];
        out[(size_t)i * np + c] = v;
    }
    // (2) A row-major, rows beyond the live count zero (:116-123)
    const int m3 = 3 * p.M, l3 = 3 * lmax; // (the loops run over the first lmax rows of every stage: lmax = M unless the flag above is set)
    if (l3 > 0) {
        const float inv_l3 = 1.0f / (float)l3;
        for (int e = tid; e < p.N * l3; e += 256) {
            int i = (int)(((float)e + 0.5f) * inv_l3); // stage (exact: e < 2^16)
            i = i * l3 > e ? i - 1 : ((i + 1) * l3 <= e ? i + 1 : i);
            const int c = e - i * l3;
            if (c / 3 >= s_lim[i]) continue;
            const size_t pbase = (size_t)b * p.NPOLY + s_pi[i];
            out[(size_t)i * np + PK_NPRE + c] = (c / 3 < s_nf[i]) ? p.poly_A[pbase * p.F * 3 + c] : 0.0;
        }
        // (3) b_j - ||E a_j||_2 (:124-125)
        const float inv_m = 1.0f / (float)lmax;
        for (int e = tid; e < p.N * lmax; e += 256) {
            int i = (int)(((float)e + 0.5f) * inv_m);
            i = i * lmax > e ? i - 1 : ((i + 1) * lmax <= e ? i + 1 : i);
            const int j = e - i * lmax;
            if (j >= s_lim[i]) continue;
            double v = 0.0;
            if (j < s_nf[i]) {
                const size_t pbase = (size_t)b * p.NPOLY + s_pi[i];
                const double *A = p.poly_A + pbase * p.F * 3, *E = p.ellipsoid + ((size_t)b * p.N + i) * 9;
                const double a0 = A[3 * j], a1 = A[3 * j + 1], a2 = A[3 * j + 2];
                double n2 = 0.0;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double t = E[3 * r] * a0 + E[3 * r + 1] * a1 + E[3 * r + 2] * a2;
                    n2 += t * t;
                }
                v = p.poly_b[pbase * p.F + j] - sqrt(n2);
            }
            out[(size_t)i * np + PK_NPRE + m3 + j] = v;
        }
    }
}

__global__ __launch_bounds__(256) void update_kernel(int B, int N, const double *__restrict__ z, const int *__restrict__ exitflag,
                                                     double *__restrict__ mpc_output)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (exitflag && exitflag[b] != FRP_EXIT_OPTIMAL) return; // the reference keeps the old plan when the solver fails (nmpc_solver.cpp:397-424)
    const double PI = 3.14159265358979323846;
    for (int e = tid; e < (N + 1) * PK_NZ; e += 256) {
        const int i = e / PK_NZ, c = e - i * PK_NZ;
        const int src = i < N ? i : N - 1;    // row N duplicates row N-1 (nmpc_solver.cpp:543)
        double v = z[((size_t)b * N + src) * PK_NZ + c];
        if (c == 16) {                        // yaw wrap (nmpc_solver.cpp:527-541)
            if (v < -PI) v += 2 * PI;
            else if (v > PI) v -= 2 * PI;
        }
        mpc_output[(size_t)b * (N + 1) * PK_NZ + e] = v;
    }
}

// NMPCSolver::initMPCOutput (nmpc_solver.cpp:265-286) as solveNMPC applies it (:363-364): a planner whose last solve did
// not return 1 restarts from the constant plan [0 0 0 T | 0 0 0 T | state].  One workgroup per planner, coalesced rows.
__global__ __launch_bounds__(256) void coldstart_kernel(int B, int N, const double *state, const int *exitflag, double thrust, double *mpc_output)
{
    const int b = blockIdx.x;
    if (exitflag && exitflag[b] == 1) return;
    double *mo = mpc_output + (size_t)b * (N + 1) * PK_NZ;
    const double *st = state ? state + (size_t)b * 9 : nullptr;
    __shared__ double row[PK_NZ];
    if (threadIdx.x < PK_NZ) {
        const int j = threadIdx.x;
        row[j] = j >= 8 ? (st ? st[j - 8] : mo[PK_NZ + j]) : ((j == 3 || j == 7) ? thrust : 0.0); // state == NULL: the plan's own stage-1 state
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (N + 1) * PK_NZ; e += 256) mo[e] = row[e % PK_NZ];
}

} // namespace frp

extern "C" int frp_nmpc_pack_batch(const frp_nmpc_pack *p, void *stream)
{
    if (!p || p->B <= 0 || p->N < 2 || p->N > 64 || p->M < 0 || p->F <= 0 || p->NPOLY <= 0 || !p->mpc_output || !p->external_acc || !p->ref_pos ||
        !p->ref_yaw || !p->ellipsoid || !p->poly_A || !p->poly_b || !p->poly_nfaces || !p->xinit || !p->x0 || !p->params || !p->nfaces)
        return FRP_ERR_ARG;
    if (!p->poly_index && p->NPOLY != p->N) return FRP_ERR_ARG;
    if ((long long)p->N * (10 + 4 * p->M) >= 65536) return FRP_ERR_ARG; // N * np must stay below 2^16 (float index split in pack_kernel)
    hipLaunchKernelGGL(frp::pack_kernel, dim3((unsigned)p->B), dim3(256), 0, static_cast<hipStream_t>(stream), *p);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

extern "C" int frp_nmpc_coldstart_batch(int B, int N, const double *state, const int *exitflag, double thrust, double *mpc_output, void *stream)
{
    if (B <= 0 || N < 1 || !mpc_output) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::coldstart_kernel, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), B, N, state, exitflag, thrust, mpc_output);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

extern "C" int frp_nmpc_update_batch(int B, int N, const double *z, const int *exitflag, double *mpc_output, void *stream)
{
    if (B <= 0 || N < 2 || !z || !mpc_output) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::update_kernel, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), B, N, z,
                       exitflag, mpc_output);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}
