// frp_reference.hip -- SURVEY 8f row f-4 (first half): the stage references of a tick, batched on the device.
// NMPCSolver::getCurTraj (plan_manage/src/nmpc_solver.cpp:109-142) samples the kinodynamic path at the stage times
// and NMPCSolver::calculate_yaw (:834-862) turns the direction to a look-ahead point into a low-pass-filtered yaw
// reference; setFORCESParams (:486, :493-495) seeds the filter with the plan's stage-1 yaw and collects
// ref_total_pos_ / ref_total_yaw_ -- the ref_pos / ref_yaw inputs of frp_nmpc_corridor_batch (f-3) and
// frp_nmpc_pack_batch (f-1).  The A* that produces the path (kinodynamic_astar.cpp) stays on the host.
//
// One wavefront per planner: lane i samples stage i (position, look-ahead direction, atan2) in parallel; the yaw
// filter is a recurrence over the stages (last_yaw_), run by lane 0 over values parked in LDS.  HBM-bound and tiny:
// 24 K bytes of path (shared paths stay in L2) + 17*8 bytes of plan in, 32 N + 4 bytes out per planner.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/frp_nmpc.h"

namespace frp {

__global__ __launch_bounds__(64) void reference_kernel(frp_nmpc_reference p)
{
    // The stage time decides, by truncation, WHICH path sample is used: no fused multiply-add here, so that a time
    // on a sample boundary lands on the same side as in the reference's scalar x86 code.
#pragma clang fp contract(off)
    __shared__ double s_yaw[64];
    __shared__ int s_far[64];
    const int b = blockIdx.x, i = threadIdx.x;
    const double *path = p.kino_path + (p.path_per_planner ? (size_t)b * p.K * 3 : 0);
    int size = p.kino_size ? p.kino_size[p.path_per_planner ? b : 0] : p.K;
    size = size < p.K ? size : p.K;
    size = size > 1 ? size : 1; // an empty path degenerates to its first sample rather than to an out-of-bounds read
    const double *plan1 = p.mpc_output + ((size_t)b * (p.N + 1) + 1) * 17; // mpc_output_.at(1)
    if (i < p.N) {
        // getCurTraj (:111-132)
        const double index_time = i * p.Ts + p.time_offset[b];
        // (a time offset of -Ts or less before the path start, or a non-finite one: the reference's unsigned index wraps and
        // reads far outside the path; here such a planner gets the end of the path instead of faulting the launch.
        // Offsets in (-Ts, 0) truncate to sample 0 as in the reference.)
        const double qf = index_time / p.Ts;
        const unsigned int ki = (qf > -1.0 && qf < 2147483647.0) ? (unsigned int)(int)qf : 0x7fffffffu;
        const double *last = path + 3 * (size_t)(size - 1);
        double r[3], f[3];
        if (ki + 1 < (unsigned int)size) {
            const double w = fmod(index_time, p.Ts) / p.Ts;
            const double *a = path + 3 * (size_t)ki;
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] = a[k] + w * (a[3 + k] - a[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] = last[k];
        }
        const double *fw = ki + 5 < (unsigned int)size ? path + 3 * (size_t)(ki + 5) : last;
#pragma unroll
        for (int k = 0; k < 3; ++k) { f[k] = fw[k]; p.ref_pos[((size_t)b * p.N + i) * 3 + k] = r[k]; }
        // calculate_yaw (:836-840): the direction part
        const double d0 = f[0] - r[0], d1 = f[1] - r[1], d2 = f[2] - r[2];
        s_far[i] = sqrt(d0 * d0 + d1 * d1 + d2 * d2) > 0.1;
        s_yaw[i] = atan2(d1, d0);
        if (i == 0 && p.replan) { // "Hard to follow the reference" (:136-140)
            const double e0 = r[0] - plan1[8], e1 = r[1] - plan1[9], e2 = r[2] - plan1[10];
            p.replan[b] = sqrt(e0 * e0 + e1 * e1 + e2 * e2) > 1.0;
        }
    }
    __syncthreads();
    if (i == 0) { // the filter recurrence (:840-861); last_yaw_ starts from the plan's stage-1 yaw (:486)
        double last_yaw = plan1[16];
        for (int k = 0; k < p.N; ++k) {
            const double yaw_temp = s_far[k] ? s_yaw[k] : last_yaw;
            double yaw = yaw_temp;
            if (fabs(yaw_temp - last_yaw) > p.pi) yaw = yaw_temp > 0 ? yaw_temp - 2 * p.pi : yaw_temp + 2 * p.pi;
            yaw = 0.2 * last_yaw + 0.8 * yaw;
            last_yaw = yaw;
            s_yaw[k] = yaw;
        }
    }
    __syncthreads();
    if (i < p.N) p.ref_yaw[(size_t)b * p.N + i] = s_yaw[i];
}

// switch_to_final (nmpc_solver.cpp:436-447); one thread per planner
__global__ void mode_kernel(int B, int N, const double *mpc_output, const double *time_offset, const int *kino_size, int size_per_planner,
                            const double *end_pt, int end_per_planner, double Ts, double radius, int *mode)
{
#pragma clang fp contract(off)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int max_index = (int)((N * Ts + time_offset[b]) / Ts);
    const double *last = mpc_output + ((size_t)b * (N + 1) + (N - 1)) * 17 + 8; // ref_end = mpc_output_.at(N - 1) position
    const double *e = end_pt + (end_per_planner ? 3 * (size_t)b : 0);
    const double d0 = last[0] - e[0], d1 = last[1] - e[1], d2 = last[2] - e[2];
    if (max_index >= kino_size[size_per_planner ? b : 0] || sqrt(d0 * d0 + d1 * d1 + d2 * d2) < radius) mode[b] = FRP_MODEL_FINAL;
}

} // namespace frp

extern "C" int frp_nmpc_mode_batch(int B, int N, const double *mpc_output, const double *time_offset, const int *kino_size,
                                   int size_per_planner, const double *end_pt, int end_per_planner, double Ts, double radius,
                                   int *mode, void *stream)
{
    if (B <= 0 || N < 1 || !mpc_output || !time_offset || !kino_size || !end_pt || !mode || !(Ts > 0.0)) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::mode_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), B, N, mpc_output,
                       time_offset, kino_size, size_per_planner, end_pt, end_per_planner, Ts, radius, mode);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

extern "C" int frp_nmpc_reference_batch(const frp_nmpc_reference *p, void *stream)
{
    if (!p || p->B <= 0 || p->N < 1 || p->N > 64 || p->K < 1 || !p->kino_path || !p->time_offset || !p->mpc_output || !p->ref_pos || !p->ref_yaw)
        return FRP_ERR_ARG;
    if (!(p->Ts > 0.0) || !(p->pi > 3.0)) return FRP_ERR_ARG;
    hipLaunchKernelGGL(frp::reference_kernel, dim3((unsigned)p->B), dim3(64), 0, static_cast<hipStream_t>(stream), *p);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}
