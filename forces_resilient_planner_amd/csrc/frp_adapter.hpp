// frp_adapter.hpp -- C++ host-side mirror of the reference's solver adapters, batched.
//
// The reference wraps its solver in FORCESNormal / FORCESFinal
//   (plan_manage/include/plan_manage/nmpc_utils.h:61-106, src/forces_normal.cpp, src/forces_final.cpp):
//     setParasNormal(w_stage_wp, w_stage_input, w_input_rate, w_terminal_wp, w_terminal_input)   :36-52
//     solveNormal(mpc_output, external_acc, ref_total_pos, ref_total_yaw, ellipsoid_matrices,
//                 poly_constraints, poly_indices)                                                :55-140
//     updateNormal(mpc_output)                                                                   :142-168
// This header restates those three methods for B independent planners at once, without Eigen / ROS /
// DecompROS types (plain row-major arrays), and hands the packed batch to the C-ABI in
// include/frp_nmpc.h.  Same names, same argument meaning, same packing arithmetic
// (b_j - ||E a_j||_2 robust tightening, zero padding of unused rows, faces beyond num_const dropped).
// The Python twin used by the tests is forces_resilient_planner_amd/adapter.py.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "../../include/frp_nmpc.h"

namespace frp {

struct HorizonValues { // plan_manage/include/plan_manage/nmpc_utils.h:43-58 (struct 'Values')
    int planning_horizon = FRP_N_REF;
    int num_var = FRP_NZ;
    int num_const = FRP_NH_REF;
    int num_pre_params = FRP_NPRE;
    int num_iter() const { return num_pre_params + 4 * num_const; }
};

// One polytope {x : A x <= b} with nf live rows (DecompROS LinearConstraint3D, decomp_geometry/polyhedron.h:98-147)
struct PolytopeView {
    const double *A; // nf x 3 row-major
    const double *b; // nf
    int nf;
};

class BatchedForcesAdapter {
public:
    BatchedForcesAdapter(int batch, int model, HorizonValues v = HorizonValues())
        : B_(batch), model_(model), value_(v),
          xinit_((size_t)batch * FRP_NX), x0_((size_t)batch * v.planning_horizon * FRP_NZ),
          params_((size_t)batch * v.planning_horizon * v.num_iter(), 0.0),
          nfaces_((size_t)batch * v.planning_horizon, 0), output_((size_t)batch * v.planning_horizon * FRP_NZ),
          exitflag_(batch, 0), iters_(batch, 0), info_((size_t)batch * FRP_INFO_STRIDE, 0.0)
    {
        // (this header against the library that is actually loaded: a stale library would write another info stride into info_)
        if (FRP_NMPC_ABI_CHECK() != FRP_OK) throw std::runtime_error("frp_nmpc: header / library ABI mismatch (see stderr)");
    }

    // forces_normal.cpp:36-52 / forces_final.cpp:36-51 -- same weights for every problem of the batch
    void setParas(double w_stage_wp, double w_stage_input, double w_input_rate, double w_terminal_wp, double w_terminal_input)
    {
        const int N = value_.planning_horizon, np = value_.num_iter();
        for (int b = 0; b < B_; b++) {
            double *p = &params_[(size_t)b * N * np];
            for (int i = 0; i < N; i++) {
                p[i * np + 6] = w_stage_wp;
                p[i * np + 7] = w_stage_input;
                p[i * np + 8] = w_input_rate;
            }
            p[(N - 1) * np + 6] = w_terminal_wp;
            p[(N - 1) * np + 7] = w_terminal_input;
        }
    }

    // forces_normal.cpp:55-136 for problem b (everything before the solver call).
    //   mpc_output: (N+1) x 17 plan deque, row-major;  external_acc[3];  ref_total_pos: N x 3;
    //   ref_total_yaw: N;  ellipsoid_matrices: N x 3 x 3;  polys[i] = poly_constraints[poly_indices(i)]
    void pack(int b, const double *mpc_output, const double *external_acc, const double *ref_total_pos,
              const double *ref_total_yaw, const double *ellipsoid_matrices, const PolytopeView *polys)
    {
        const int N = value_.planning_horizon, np = value_.num_iter(), M = value_.num_const, pre = value_.num_pre_params;
        std::memcpy(&xinit_[(size_t)b * FRP_NX], mpc_output + 1 * FRP_NZ + 8, FRP_NX * sizeof(double)); // :62-72
        std::memcpy(&x0_[(size_t)b * N * FRP_NZ], mpc_output + FRP_NZ, (size_t)N * FRP_NZ * sizeof(double)); // :74-97
        double *p = &params_[(size_t)b * N * np];
        for (int i = 0; i < N; i++) {
            double *pi = p + (size_t)i * np;
            for (int c = 0; c < 3; c++) {
                pi[c] = ref_total_pos[3 * i + c]; // :99-102
                pi[3 + c] = external_acc[c];      // :103-106
            }
            pi[9] = ref_total_yaw[i]; // :107-108
            const PolytopeView &pl = polys[i];
            const double *E = ellipsoid_matrices + 9 * i;
            const int nf = pl.nf < M ? pl.nf : M; // faces beyond num_const are dropped (:114)
            for (int j = 0; j < M; j++) {
                double *Aj = pi + pre + 3 * j;
                double &bj = pi[pre + 3 * M + j];
                if (j < nf) {
                    const double *a = pl.A + 3 * j;
                    Aj[0] = a[0]; Aj[1] = a[1]; Aj[2] = a[2];
                    double n2 = 0.0; // b_j - || E a_j ||_2  (:124-125)
                    for (int r = 0; r < 3; r++) {
                        const double t = E[3 * r] * a[0] + E[3 * r + 1] * a[1] + E[3 * r + 2] * a[2];
                        n2 += t * t;
                    }
                    bj = pl.b[j] - std::sqrt(n2);
                } else { // :127-135
                    Aj[0] = Aj[1] = Aj[2] = 0.0;
                    bj = 0.0;
                }
            }
            nfaces_[(size_t)b * N + i] = nf;
        }
    }

    // forces_normal.cpp:139 for the whole batch, host buffers (H2D + solve + D2H inside the library).
    // Returns FRP_OK or a library error; per-problem exit flags are in exitflag().
    int solve(const frp_nmpc_options *opt = nullptr)
    {
        frp_nmpc_batch bt;
        std::memset(&bt, 0, sizeof bt);
        bt.B = B_; bt.N = value_.planning_horizon; bt.M = value_.num_const; bt.model = model_;
        int mf = 0;
        for (int v : nfaces_) mf = v > mf ? v : mf;
        bt.MF = mf;
        bt.xinit = xinit_.data(); bt.x0 = x0_.data(); bt.params = params_.data(); bt.nfaces = nfaces_.data();
        bt.z = output_.data(); bt.exitflag = exitflag_.data(); bt.iters = iters_.data(); bt.info = info_.data();
        return frp_nmpc_solve_batch_host(&bt, opt);
    }

    // forces_normal.cpp:142-168 for problem b: output.x01..xN -> mpc_output rows 0..N-1
    void update(int b, double *mpc_output) const
    {
        const int N = value_.planning_horizon;
        std::memcpy(mpc_output, &output_[(size_t)b * N * FRP_NZ], (size_t)N * FRP_NZ * sizeof(double));
    }

    // NMPCSolver::updateFORCESResults (nmpc_solver.cpp:524-543): yaw wrap of rows 0..N-1, duplicate the last row
    static void updateFORCESResults(double *mpc_output, int N)
    {
        const double PI = 3.14159265358979323846;
        for (int i = 0; i < N; i++) {
            double &yaw = mpc_output[i * FRP_NZ + 16];
            if (yaw < -PI) yaw += 2 * PI;
            else if (yaw > PI) yaw -= 2 * PI;
        }
        std::memcpy(mpc_output + (size_t)N * FRP_NZ, mpc_output + (size_t)(N - 1) * FRP_NZ, FRP_NZ * sizeof(double));
    }

    // NMPCSolver::initMPCOutput (nmpc_solver.cpp:265-286): cold-start plan, hover thrust guess 7.3 N (nmpc_utils.h:191)
    static void initMPCOutput(const double state[9], double *mpc_output, int N, double thrust = 7.3)
    {
        for (int i = 0; i <= N; i++) {
            double *r = mpc_output + (size_t)i * FRP_NZ;
            r[0] = r[1] = r[2] = 0.0; r[3] = thrust;
            r[4] = r[5] = r[6] = 0.0; r[7] = thrust;
            std::memcpy(r + 8, state, 9 * sizeof(double));
        }
    }

    const std::vector<int> &exitflag() const { return exitflag_; }
    const std::vector<int> &iterations() const { return iters_; }
    const std::vector<double> &params() const { return params_; }
    const std::vector<double> &xinit() const { return xinit_; }
    const std::vector<double> &x0() const { return x0_; }
    const std::vector<int> &nfaces() const { return nfaces_; }
    const std::vector<double> &output() const { return output_; }

private:
    int B_, model_;
    HorizonValues value_;
    std::vector<double> xinit_, x0_, params_;
    std::vector<int> nfaces_;
    std::vector<double> output_;
    std::vector<int> exitflag_, iters_;
    std::vector<double> info_;
};

} // namespace frp
