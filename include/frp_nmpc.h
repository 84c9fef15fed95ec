/*
 * frp_nmpc.h -- C-ABI of libfrp_nmpc_amd.so, the MI355X (gfx950) NMPC solver that replaces the
 * solver behind the reference's plan_manage adapters.
 *
 * Two groups of entry points:
 *
 * (1) Drop-in symbols with the reference's exact names, argument order and struct layouts:
 *       FORCESNLPsolver_normal_solve / FORCESNLPsolver_final_solve
 *     replacing the licence-locked ForcesPro binary
 *       plan_manage/solver/normal/FORCESNLPsolver_normal/include/FORCESNLPsolver_normal.h:317-323
 *       plan_manage/solver/final/FORCESNLPsolver_final/include/FORCESNLPsolver_final.h:317-323
 *     called at plan_manage/src/forces_normal.cpp:139 and forces_final.cpp:138.
 *     A caller compiled against the reference's own headers links against this library unchanged
 *     (sizes/offsets are asserted in csrc/frp_capi.hip: params 23600 B, output 2720 B, info 136 B).
 *
 * (2) A batched, horizon-generic API (not in the reference, needed by BASELINE.json configs[1..4]):
 *     B independent problems, device-resident buffers, per-problem exit flags.
 *
 * All pointers in (2) are DEVICE pointers unless the name ends in _host; no torch / C++ types cross
 * this boundary.  All functions return 0 on success or a negative frp error; solver exit flags
 * (per problem) use the reference's codes (FORCESNLPsolver_normal.h:110-139).
 */
#ifndef FRP_NMPC_H
#define FRP_NMPC_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- problem constants (matlab_code/setup.m:36-43) ---- */
#define FRP_NZ 17
#define FRP_NX 9
#define FRP_NEQ 13
#define FRP_NPRE 10
#define FRP_NPAR(M) (FRP_NPRE + 4 * (M))
#define FRP_N_REF 20
#define FRP_NH_REF 30

#define FRP_MODEL_NORMAL 0 /* stage-N cost of mpc_objectiveN_normal.m */
#define FRP_MODEL_FINAL 1  /* stage-N cost of mpc_objectiveN_final.m (adds 20 w_wp |v|^2) */

/* per-problem solver exit flags, same values as the reference (FORCESNLPsolver_normal.h:110-139) */
#define FRP_EXIT_OPTIMAL 1
#define FRP_EXIT_MAXIT 0
#define FRP_EXIT_FACTORIZATION (-5)
#define FRP_EXIT_BADFUNCEVAL (-6)
#define FRP_EXIT_NOPROGRESS (-7)
#define FRP_EXIT_PARAM_VALUE (-11)
/* Values of the reference's list that this solver NEVER returns (FORCESNLPsolver_normal.h:110-139): 2 TIMEOUT (:118 -- there is no
   wall-clock budget: the iteration limit `maxit` bounds a solve, and exhausting it is 0 = MAXIT), -4 (wrong number of inequalities:
   the face counts are validated as -11), -12 PARAM_VALUE_TIMEOUT (:136).  A caller that switches on them needs no new case. */
/* Return values of the two drop-in entry points that are NOT solver outcomes.  The reference's callers accept a plan only on
   exitflag == 1 (nmpc_solver.cpp:398) and treat every other value alike, so these are safe for them; a caller that looks
   closer can tell a machine problem from a bad parameter:
     -100  the reference's own LICENSE_ERROR value, "solver not valid on this machine" (FORCESNLPsolver_normal.h:139): no usable
           HIP device -- this library has no CPU path;
     -101  (not a ForcesPro value) a HIP runtime call failed during the solve: allocation, copy, launch or synchronisation. */
#define FRP_EXIT_NO_DEVICE (-100)
#define FRP_EXIT_DEVICE_FAULT (-101)

/* library-level errors */
#define FRP_OK 0
#define FRP_ERR_NO_DEVICE (-1001)
#define FRP_ERR_HIP (-1002)
#define FRP_ERR_ARG (-1003)
#define FRP_ERR_MODEL_MISMATCH (-1004)

#define FRP_INFO_STRIDE 12 /* doubles per problem in the info array, see frp_nmpc_batch.info */

/* Layout version of everything below (the two FORCES entry points keep the reference's layout for ever).  It changes whenever a
 * struct of this header gains, loses or moves a field or FRP_INFO_STRIDE changes: a caller built against another header would
 * hand over short structs / a short info array.  Call FRP_NMPC_ABI_CHECK() once after loading the library (the C++ adapter and the
 * Python loader do) and refuse to continue unless it returns FRP_OK. */
#define FRP_NMPC_ABI_VERSION 6
int frp_nmpc_abi_version(void);
/* FRP_OK when the caller's header agrees with the library on the version, on the size of frp_nmpc_options and frp_nmpc_batch and
 * on the info stride; FRP_ERR_ARG otherwise (with one line on stderr saying what differs). */
int frp_nmpc_abi_check(int abi_version, size_t options_bytes, size_t batch_bytes, int info_stride);
#define FRP_NMPC_ABI_CHECK() frp_nmpc_abi_check(FRP_NMPC_ABI_VERSION, sizeof(frp_nmpc_options), sizeof(frp_nmpc_batch), FRP_INFO_STRIDE)

typedef struct frp_nmpc_options {
    int maxit;        /* 200  (FORCESNLPsolver_normal.h:86)                */
    double tol_stat;  /* 1e-4 (mpc_generator_normal.m:76)                  */
    double tol_eq;    /* 1e-4 (:77)                                        */
    double tol_ineq;  /* 1e-4 (:78)                                        */
    double tol_comp;  /* 1e-4 (:79)                                        */
    double mu0;       /* initial barrier parameter, 1.0                    */
    double ftb;       /* fraction to boundary, 0.99 (normal.h:89)          */
    int hessian;      /* 1 (default): exact Lagrangian Hessian of the RK2 dynamics with Gauss-Newton
                         fallback when the reduced Hessian is indefinite; 0: Gauss-Newton only   */
    double diverge_mu;/* 1e3: exit -7 (NOPROGRESS) once the average complementarity exceeds
                         diverge_mu * max(1, mu0) -- multipliers growing without bound = a (locally)
                         infeasible instance.  Converging solves of the BASELINE workloads stay below 10;
                         10 is the early-exit setting of the configs[3] benchmark (bench.py)          */
    int twist;        /* 0 (default): one backward Riccati recursion over the horizon, on one wavefront.
                         m > 0: the Newton system of an iteration is solved from both ends of the horizon at once -- the
                         stages 0 .. m-1 forward (arrival-cost recursion, the pinned x_0 as a 1e12 penalty) on a second
                         wavefront while the stages m .. N-1 run backward; the halves meet in a 13 x 13 system at stage m.
                         -1: m = 9 N / 20 (3 N / 10 when the launch has more than 1024 problems).  Horizons 4 <= N <= 20 with 2 <= m <= N - 2, anything else runs the plain
                         solve.  A LATENCY option: it shortens the dependency chain of an iteration (-13 % per launch
                         while there are fewer problems than resident workgroups, <= ~1000), and costs throughput
                         once the GPU is full (+3 % at 4096 problems).  The Newton direction carries the penalty's
                         rounding (~1e-5 relative); residuals and termination tests are the plain solve's, so the
                         same KKT points are reached (oracle: orc_options.twist).  DESIGN 9.1                     */
} frp_nmpc_options;

typedef struct frp_nmpc_batch {
    int B;      /* problems                                                             */
    int N;      /* horizon (stages); 20 in the reference (setup.m:36), <= 64 here       */
    int M;      /* corridor rows in the parameter layout (30 in the reference, setup.m:42) */
    int MF;     /* max LIVE corridor rows of any stage (<= M, <= 30 = the reference's
                   num_const: its adapter drops rows beyond 30, forces_normal.cpp:114);
                   selects the kernel variant; a larger value is FRP_ERR_ARG               */
    int model;  /* FRP_MODEL_NORMAL / FRP_MODEL_FINAL                                    */
    const double *xinit;  /* [B][9]            params.xinit            (normal.h:156)  */
    const double *x0;     /* [B][N][17]        params.x0, initial guess (normal.h:159)  */
    const double *params; /* [B][N][10+4M]     params.all_parameters   (normal.h:162),
                             per stage: ref(3) f_ext(3) w_wp w_in w_rate yaw_ref | A row-major Mx3 | b(M)
                             (matlab_code/setup.m:60-66)                                 */
    const int *nfaces;    /* [B][N] live rows per stage, or NULL: trailing all-zero rows (the padding
                             forces_normal.cpp:127-135 writes) are detected and dropped  */
    double *z;            /* [B][N][17] out    output.x01..xN          (normal.h:173-236) */
    int *exitflag;        /* [B] out                                                     */
    int *iters;           /* [B] out, interior-point iterations (info.it)                */
    double *info;         /* [B][FRP_INFO_STRIDE] out or NULL (FORCESNLPsolver_normal.h:241-301 where a field has a namesake):
                             [0] res_eq, [1] res_ineq, [2] rsnorm, [3] rcompnorm, [4] pobj, [5] mu -- at the returned
                             iterate; [6] step_cc, [8] mu_aff, [9] sigma, [10] step_aff -- of the last iteration taken
                             (0 when none was); [7] iterations redone with the Gauss-Newton Hessian;
                             [11] dgap = sum of slack * multiplier over all inequalities at the returned iterate (the
                             duality gap of the local QP model: dobj = pobj - dgap, rdgap = |dgap / pobj|)            */
    const int *model_per_problem; /* [B] FRP_MODEL_* of each problem, or NULL: `model` for all.  A fleet whose
                             planners switch to the final solver one by one (switch_to_final,
                             nmpc_solver.cpp:381, 446-447) stays one batch.                    */
    const int *order_hint; /* [B] expected work of each problem, or NULL.  Only the ORDER in which the solver's
                             persistent workgroups take problems off the queue depends on it (largest first), never a
                             result.  A receding-horizon caller passes the PREVIOUS tick's `iters` -- the same buffer
                             as `iters` is fine, it is read before the solve writes it; values <= 0 = unknown.
                             NULL: ordered by the objective of the initial guess.                               */
} frp_nmpc_batch;

void frp_nmpc_default_options(frp_nmpc_options *opt);

/* Bytes of device workspace frp_nmpc_solve_batch needs for (B, N, MF). */
size_t frp_nmpc_workspace_bytes(int B, int N, int MF);

/* Solve B problems.  `workspace` = device buffer of at least frp_nmpc_workspace_bytes() bytes,
 * `stream` = hipStream_t (NULL = default stream).  Asynchronous: returns after the launch. */
int frp_nmpc_solve_batch(const frp_nmpc_batch *batch, const frp_nmpc_options *opt,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Same with HOST buffers: allocates/copies/synchronises internally (plumbing + tests). */
int frp_nmpc_solve_batch_host(const frp_nmpc_batch *batch_host, const frp_nmpc_options *opt);

/* Optional: pin a caller buffer in place and map it into the device's address space (hipHostRegister).  When EVERY array of a
 * frp_nmpc_solve_batch_host call lies inside registered ranges the call stages nothing: a gather kernel reads the live part of the
 * inputs straight from the caller's memory and the solver writes plans, flags and diagnostics in place (same plans, bit for bit).
 * The caller keeps the buffer alive and unchanged in size until frp_nmpc_host_unregister(ptr) (same `ptr`); registering costs
 * milliseconds -- do it once for buffers that are reused from tick to tick, not per call.
 * Registrations are counted: the same (ptr) registered twice is unpinned by its second unregistration; a range that lies INSIDE a
 * registered range is accepted as an alias of it (nothing is pinned twice) and is unregistered by its own pointer -- the enclosing
 * range cannot be unregistered before its aliases (FRP_ERR_ARG); a range that partly overlaps a registered one is refused.  An array
 * that is freed without being unregistered leaves its pages pinned AND its address range in the registry: a later allocation at the same
 * address would be taken for the old mapping.  frp_nmpc_host_registered() answers whether [ptr, ptr + bytes) is covered;
 * frp_nmpc_host_unregister_all() drops every registration (it waits for a host batch in flight). */
int frp_nmpc_host_register(void *ptr, size_t bytes);
int frp_nmpc_host_unregister(void *ptr);
int frp_nmpc_host_registered(const void *ptr, size_t bytes);
int frp_nmpc_host_unregister_all(void);

/* Two host batches in flight (registered buffers only; FRP_ERR_ARG if any array of the batch is not registered): _begin enqueues the
 * gather of the batch's inputs and its solve and returns a ticket, _wait blocks until that batch's plans, flags and diagnostics are in
 * the caller's arrays (written in place by the solver).  While batch k solves, the gather kernel of batch k + 1 reads its inputs over
 * the host link (the pipelined solves leave a few resident workgroup slots free for it), so a caller that alternates two sets of
 * registered buffers -- begin(A); begin(B); loop { wait(A); use A; refill A; begin(A); wait(B); ... } -- sees max(gather, solve) per
 * batch instead of their sum.  Same plans as frp_nmpc_solve_batch_host, bit for bit.  At most FRP_NMPC_HOST_INFLIGHT tickets may be
 * outstanding (a further _begin returns FRP_ERR_ARG); the arrays of a batch must stay registered and untouched between its _begin and
 * _wait.  (A caller that keeps its planner state on the device needs none of this: frp_nmpc_solve_batch.) */
#define FRP_NMPC_HOST_INFLIGHT 2
int frp_nmpc_solve_batch_host_begin(const frp_nmpc_batch *batch_host, const frp_nmpc_options *opt, int *ticket);
int frp_nmpc_solve_batch_host_wait(int ticket);

/* Batched model callback = the reference's extfunc (FORCESNLPsolver_normal.h:321,
 * FORCESNLPsolver_normal_casadi2forces.c:42-245) for B*N stage points at once.
 * z [B][N][17], params [B][N][10+4M]; outputs (any may be NULL):
 *   f [B][N], grad_f [B][N][17], c [B][N][13], jac_c [B][N][13*17] column-major ld 13 (stage N-1:
 *   zeros, no dynamics there), h [B][N][M], A of the corridor is its own Jacobian and is not copied. */
int frp_nmpc_stage_eval(int B, int N, int M, int model, const double *z, const double *params,
                        double *f, double *grad_f, double *c, double *jac_c, double *h, void *stream);
int frp_nmpc_stage_eval_host(int B, int N, int M, int model, const double *z, const double *params,
                             double *f, double *grad_f, double *c, double *jac_c, double *h);

/* Average kernel duration (ms) of the last `frp_nmpc_time_solve` call: launches the solve `reps`
 * times on `stream` bracketed by hipEvents on that same stream (bench.py's roofline leg). */
int frp_nmpc_time_solve(const frp_nmpc_batch *batch, const frp_nmpc_options *opt, void *workspace,
                        size_t workspace_bytes, void *stream, int reps, float *avg_ms);

/* Measurement hook (bench.py's roofline leg): between _begin and _end every `stride`-th frp_nmpc_solve_batch call records a
 * hipEvent pair on ITS launch stream immediately around the dominant kernel (the interior-point solve; the launch-order
 * kernels in front of it are outside the pair), for at most `max_launches` recorded calls.  _end waits for the recorded
 * events and returns the number of launches recorded and their average duration in ms -- a sample of the SAME launches a
 * caller timed from the host around the region (an event pair costs the launch ~8 us of queue bubbles on MI355X: stride 4
 * keeps the observed region within 0.2 % of the unobserved one).  Process-global, not re-entrant; FRP_ERR_ARG when misused. */
int frp_nmpc_kernel_timing_begin(int max_launches, int stride);
int frp_nmpc_kernel_timing_end(float *avg_ms, int *launches);

/* Tuning / test hook: launches of the plain solve with N <= 20 and at most 6 corridor rows per stage run on the
 * four-problems-per-CU kernel variants (three-wavefront workgroups) when they hold MORE than `min_batch` problems;
 * below that a problem has a CU nearly to itself and the four-wavefront variants iterate faster.  Default (and `min_batch` < 0):
 * three workgroups per CU of the current device.  0 puts every covered launch on the four-per-CU variants (the parity tests
 * do that at their small batch sizes).  Returns the previous value.  Process-global.
 * Round 6: the same threshold moves the second high-residency variant -- horizons 20 < N <= 30 with at most 16 corridor rows per
 * stage at THREE problems per CU (DESIGN 4 "Round 6"; its own default: more than two workgroups per CU worth of problems). */
int frp_nmpc_set_q4_min_batch(int min_batch);

/* ---- (3) SURVEY 8f row f-1: the adapter's packing / result bookkeeping on the device (all pointers DEVICE) ---- */
typedef struct frp_nmpc_pack {
    int B, N, M;   /* problems, horizon, corridor rows of the parameter layout (num_const, nmpc_utils.h:50)      */
    int NPOLY;     /* polytopes stored per problem (= N when poly_index is NULL)                                 */
    int F;         /* rows stored per polytope (>= any nfaces; rows beyond M are dropped, forces_normal.cpp:114) */
    int external_acc_per_stage;
    /* inputs: the arguments of FORCESNormal::solveNormal (plan_manage/src/forces_normal.cpp:55-60)              */
    const double *mpc_output;   /* [B][N+1][17]  plan deque (row 0 = applied stage)                              */
    const double *external_acc; /* [B][3], or [B][N][3] when external_acc_per_stage != 0 (the per-stage parameter
                                   layout allows it, matlab_code/setup.m:62; the reference passes one vector)       */
    const double *ref_pos;      /* [B][N][3]     ref_total_pos                                                   */
    const double *ref_yaw;      /* [B][N]        ref_total_yaw                                                   */
    const double *ellipsoid;    /* [B][N][3][3]  ellipsoid_matrices E_i (row-major)                              */
    const double *poly_A;       /* [B][NPOLY][F][3]  poly_constraints[.].A_                                      */
    const double *poly_b;       /* [B][NPOLY][F]     poly_constraints[.].b_                                      */
    const int *poly_nfaces;     /* [B][NPOLY]    live rows of each polytope                                      */
    const int *poly_index;      /* [B][N]        poly_indices(i), or NULL: stage i uses polytope i               */
    /* the weights of setParasNormal / setParasFinal (forces_normal.cpp:36-52)                                   */
    double w_stage_wp, w_stage_input, w_input_rate, w_terminal_wp, w_terminal_input;
    /* outputs: the solver inputs of frp_nmpc_batch                                                              */
    double *xinit;  /* [B][9]        */
    double *x0;     /* [B][N][17]    */
    double *params; /* [B][N][10+4M] */
    int *nfaces;    /* [B][N]        */
    /* optional: planners in final mode (switch_to_final) get setParasFinal's weights (forces_final.cpp:36-52)  */
    const int *mode; /* [B] FRP_MODEL_NORMAL / FRP_MODEL_FINAL per planner, or NULL: the weights above for everyone */
    double wf_stage_wp, wf_stage_input, wf_input_rate, wf_terminal_wp, wf_terminal_input;
    /* != 0: the caller guarantees that `params` and `nfaces` are as the PREVIOUS frp_nmpc_pack_batch call with the same
       B, N, M left them (a receding-horizon loop that owns its buffers): the rows of a stage beyond nfaces[stage] are
       zero already, the call writes the live rows and zeroes only those that were live before -- a fleet with six-face
       corridors writes 34 of the 130 slots of a stage.  0 (the first call on fresh buffers, or when in doubt): every slot. */
    int padded_rows_are_zero;
} frp_nmpc_pack;

/* forces_normal.cpp:36-136 for B planners: weights, xinit / shifted x0, per-stage parameters with the robust
 * tightening b_j - ||E_i a_j||_2 and zero padding.  Asynchronous on `stream`. */
int frp_nmpc_pack_batch(const frp_nmpc_pack *p, void *stream);

/* updateNormal (forces_normal.cpp:142-168) + NMPCSolver::updateFORCESResults (nmpc_solver.cpp:524-543) for B
 * planners: plan rows 0..N-1 <- z, yaw wrapped into [-pi, pi], row N <- row N-1.  With exitflag != NULL a planner
 * whose solve did not return 1 keeps its previous plan (nmpc_solver.cpp:397-424). */
int frp_nmpc_update_batch(int B, int N, const double *z, const int *exitflag, double *mpc_output, void *stream);

/* ---- (4) SURVEY 8f row f-2: tube (ego + disturbance ellipsoid) propagation on the device ---- */
typedef struct frp_nmpc_tube {
    int B, N;                 /* planners, horizon (planning_horizon_, <= 64)                                      */
    const double *mpc_output; /* [B][N+1][17]  plan deque; rows 0..N-1 are linearised (nmpc_solver.cpp:498-501)    */
    double mass, drag;        /* nmpc/mass, nmpc/drag_coefficient (nmpc_solver.cpp:72-73)                          */
    double ego_r, ego_h;      /* nmpc/ego_r, nmpc/ego_h: ego_size_ = diag(r^2, r^2, h^2) (:69-70, :90-92)          */
    double noise[3];          /* w_: nmpc/ext_noise_bound per channel (:74, :99)                                   */
    double epsilon, Ts;       /* nmpc_utils.h:188-189                                                              */
    double *ellipsoid;        /* [B][N][3][3] out: ellipsoid_matrices_ E_i (row-major) = frp_nmpc_pack.ellipsoid   */
} frp_nmpc_tube;

/* NMPCSolver::setFORCESParams' tube part (nmpc_solver.cpp:484-521) with updateMatrix (:615-699) and
 * getDistrEllipsoid (:567-611) for B planners.  The feedback gain K is the reference's constant (:28-31).
 * Asynchronous on `stream`. */
int frp_nmpc_tube_batch(const frp_nmpc_tube *p, void *stream);

/* ---- (5) SURVEY 8f row f-3: corridor generation / selection on the device ---- */
#define FRP_CORRIDOR_MAX_F 64           /* rows kept per polytope                                   */
#define FRP_CORRIDOR_MAX_POINTS 65536   /* cloud points per planner (3 x P/8 bytes of LDS masks)    */
typedef struct frp_nmpc_corridor {
    int B, N;               /* planners, horizon                                                                    */
    int F;                  /* rows stored per polytope, 6 <= F <= FRP_CORRIDOR_MAX_F (= frp_nmpc_pack.F)            */
    int P;                  /* points stored per cloud                                                              */
    const double *cloud;    /* [P][3] shared by all planners, or [B][P][3] when cloud_per_planner != 0: vec_obs_    */
    int cloud_per_planner;
    const int *cloud_count; /* live points (<= P) per cloud, [1] or [B]; NULL: P                                    */
    const double *ref_pos;  /* [B][N][3]     ref_pos_ of stage i (nmpc_solver.cpp:117-122)                          */
    const double *ref_yaw;  /* [B][N]        ref_yaw_ of stage i (:861)                                             */
    const double *ellipsoid;/* [B][N][3][3]  E_i (frp_nmpc_tube.ellipsoid)                                          */
    double bbox[3];         /* set_local_bbox(Vec3f(2, 2, 1)) (:323)                                                */
    double seed_len;        /* 0.1: second seed point ahead along the yaw (:318)                                    */
    double inflation;       /* 1.1: "with little inflation" (:302)                                                  */
    double offset_x;        /* 0: dilate()'s offset on the long semi-axis (ellipsoid_decomp.h:67)                   */
    /* outputs = the polytope inputs of frp_nmpc_pack (NPOLY = N)                                                   */
    double *poly_A;         /* [B][N][F][3]  LinearConstraint3D::A_ of polytope k (rows beyond F are not stored)    */
    double *poly_b;         /* [B][N][F]                                                                            */
    int *poly_nfaces;       /* [B][N]        rows of polytope k (may exceed F), 0 for unused polytopes              */
    int *poly_index;        /* [B][N]        poly_indices(i)                                                        */
    int *poly_count;        /* [B] or NULL   polytopes made; negated when one had more than F rows, in which case the
                                             containment test of later stages saw only its first F rows              */
    /* optional uniform grid over a SHARED cloud (frp_nmpc_cloud_grid_build); grid_start == NULL: scan the whole cloud  */
    double grid_origin[3], grid_cell;
    int grid_dims[3];
    const double *grid_points; /* [P][3] the cloud sorted by cell                                                      */
    const int *grid_index;     /* [P]    original cloud index of each sorted point                                     */
    const int *grid_start;     /* [nx*ny*nz + 1] first sorted point of each cell (x fastest, then y, then z)           */
} frp_nmpc_corridor;

#define FRP_CORRIDOR_MAX_CELLS (1 << 22)
/* Bins a cloud into a uniform grid so that a decomposition reads only the cells its local box touches instead of the
 * whole cloud (same results: minima are tie-broken by the original cloud index).  Points outside the grid, NaNs
 * included, are binned into its border cells.  Buffers (device): grid_points [P][3], grid_index [P], grid_start
 * [cells + 1], scratch [cells].  Rebuild when the cloud changes (cloudCallback, nmpc_solver.cpp:989-996), not per tick. */
int frp_nmpc_cloud_grid_build(const double *cloud, int P, const double origin[3], double cell, const int dims[3],
                              double *grid_points, int *grid_index, int *grid_start, int *scratch, void *stream);

/* For every planner, the getSikangConst calls of NMPCSolver::setFORCESParams (nmpc_solver.cpp:288-332, :515):
 * stage i keeps the latest polytope while its tube ellipsoid, inflated, fits; otherwise DecompROS'
 * EllipsoidDecomp3D::dilate is run on the seed segment (decomp_util/line_segment.h:31-35).  Asynchronous on `stream`. */
int frp_nmpc_corridor_batch(const frp_nmpc_corridor *p, void *stream);

/* ---- (6) SURVEY 8f row f-4 (first half): stage references from the kinodynamic path, on the device ---- */
typedef struct frp_nmpc_reference {
    int B, N, K;               /* planners, horizon, samples stored per path                                        */
    const double *kino_path;   /* [K][3] shared, or [B][K][3] when path_per_planner != 0: kino_path_ =
                                  KinodynamicAstar::getKinoTraj(Ts_) (nmpc_solver.cpp:215)                           */
    int path_per_planner;
    const int *kino_size;      /* kino_size_ (<= K) per path, [1] or [B]; NULL: K                                   */
    const double *time_offset; /* [B] (mpc_start_time_ - kino_start_time_).toSec() (:111)                           */
    const double *mpc_output;  /* [B][N+1][17] plan deque: row 1 seeds last_yaw_ (:486) and the replan test (:136)  */
    double Ts;                 /* nmpc_utils.h:189                                                                  */
    double pi;                 /* the file's own constant, 3.1415926 (nmpc_solver.cpp:3)                            */
    double *ref_pos;           /* [B][N][3] out: ref_total_pos_                                                     */
    double *ref_yaw;           /* [B][N]    out: ref_total_yaw_                                                     */
    int *replan;               /* [B] out or NULL: kino_replan_ raised by stage 0 (:136-140)                        */
} frp_nmpc_reference;

/* NMPCSolver::getCurTraj (nmpc_solver.cpp:109-142) + calculate_yaw (:834-862) for all stages of B planners.
 * Asynchronous on `stream`. */
int frp_nmpc_reference_batch(const frp_nmpc_reference *p, void *stream);

/* The mode switch at the end of NMPCSolver::solveNMPC (nmpc_solver.cpp:436-447) for B planners: a planner whose
 * horizon runs past its kinodynamic path ((int)((N Ts + time_offset) / Ts) >= kino_size) or whose plan's last stage
 * is within `radius` (1.0 m) of the goal end_pt switches to the final solver -- mode[b] = FRP_MODEL_FINAL -- and stays
 * there until the caller resets it (a new path, :218).  kino_size: [1] or [B] (size_per_planner), end_pt: [3] or [B][3]
 * (end_per_planner).  mode feeds frp_nmpc_pack.mode and frp_nmpc_batch.model_per_problem.  Asynchronous on `stream`. */
int frp_nmpc_mode_batch(int B, int N, const double *mpc_output, const double *time_offset, const int *kino_size,
                        int size_per_planner, const double *end_pt, int end_per_planner, double Ts, double radius,
                        int *mode, void *stream);

/* NMPCSolver::initMPCOutput (nmpc_solver.cpp:265-286) as applied at the start of solveNMPC (:363-364) for B planners:
 * every planner whose exitflag != 1 (all of them when exitflag is NULL) gets the constant cold-start plan
 * [0 0 0 T | 0 0 0 T | state] in all N+1 rows; T = real_thrust_c_ (nmpc_utils.h:191).  state [B][9] = stateMpc_, or NULL
 * to restart from the plan's own stage-1 state.  Asynchronous on `stream`. */
int frp_nmpc_coldstart_batch(int B, int N, const double *state, const int *exitflag, double thrust, double *mpc_output,
                             void *stream);

/* ---- (7) SURVEY 8f row f-4 (second half): the kinodynamic A* front end on the device ---- */
/* KinodynamicAstar::search return values (path_searching/include/path_searching/kinodynamic_astar.h:169) */
#define FRP_ASTAR_REACH_HORIZON 1
#define FRP_ASTAR_REACH_END 2
#define FRP_ASTAR_NO_PATH 3
#define FRP_ASTAR_REACH_END_BUT_SHOT_FAILS 4
#define FRP_ASTAR_MAX_PATH 256 /* path nodes reported per planner (a node per max_tau seconds of path) */
typedef struct frp_nmpc_astar {
    int B;                     /* planners                                                                              */
    /* the occupancy map all planners search (OccMap, occ_grid/src/occ_map.cpp)                                          */
    const unsigned char *occ;  /* [gx][gy][gz], != 0 <=> occupancy_buffer_ > min_occupancy_log_ (occ_map.cpp:105)        */
    int grid[3];               /* grid_size_ = ceil(map_size_ / resolution_) (occ_map.cpp:789)                           */
    double origin[3];          /* occ_map/origin_*                                                                       */
    double map_size[3];        /* occ_map/map_size_*: states are bounded by (origin, map_size / 2), z by (0.1, map_size_z / 2)
                                  (kinodynamic_astar.cpp:152-154)                                                        */
    double resolution;         /* occ_map/resolution = the A* voxel size (intialGridMap, kinodynamic_astar.cpp:511)       */
    const int *local_box;      /* [B][6] min_id(3), max_id(3) of OccMap::isInLocalMap (voxels outside read as free,
                                  occ_map.cpp:45-57, 101-102), or NULL: the whole map is local                           */
    double ego_r, ego_h;       /* nmpc/ego_r, nmpc/ego_h: the swept cross of checkState (occ_map.cpp:645-684)             */
    /* search parameters (setParam, kinodynamic_astar.cpp:290-305)                                                     */
    double max_tau, init_max_tau, max_vel, max_acc, w_time, horizon, lambda_heu;
    double tie_breaker;        /* 1 + 1 / 10000 (kinodynamic_astar.h:139)                                                */
    int allocate_num;          /* node pool per planner (search/allocate_num; "run out of memory" -> NO_PATH, :255-259)   */
    int check_num;             /* collision samples per primitive (search/check_num)                                     */
    /* the arguments of KinodynamicAstar::search (kinodynamic_astar.cpp:17-19), per planner                              */
    const double *start_pt, *start_vel, *start_acc, *end_pt, *end_vel; /* [B][3] each                                     */
    const double *external_acc; /* [B][3] updateExternalAcc: added to every primitive's input (stateTransit, :838)        */
    const int *active;         /* [B] or NULL: planners with active[b] == 0 are skipped and keep their path (kino_replan_ of
                                  frp_nmpc_reference.replan selects who replans, nmpc_solver.cpp:475-479)                  */
    int init_search;           /* `init`: the first expansion keeps start_acc for init_max_tau / 8 .. init_max_tau; when it
                                  ends in NO_PATH the search is repeated with init = false (nmpc_solver.cpp:190-207)      */
    double Ts;                 /* sampling step of getKinoTraj (nmpc_utils.h:189)                                        */
    int K;                     /* samples stored per path                                                                */
    /* outputs                                                                                                           */
    double *kino_path;         /* [B][K][3] kino_path_ = getKinoTraj(Ts) (nmpc_solver.cpp:209) = frp_nmpc_reference.kino_path
                                  with path_per_planner != 0                                                             */
    int *kino_size;            /* [B] kino_size_; a planner whose search ends in NO_PATH keeps its previous path and size
                                  (getKinoPath returns before assigning them, nmpc_solver.cpp:195-198)                     */
    int *status;               /* [B] FRP_ASTAR_*                                                                        */
    int *stats;                /* [B][4] or NULL: nodes used, expansions, 1 if the search was repeated, path nodes -- negated when
                                  something was lost: more than K samples (the first K are kept), or more path nodes than
                                  FRP_ASTAR_MAX_PATH (then NOTHING is returned: status NO_PATH, the planner keeps its path) */
    double *path_nodes;        /* [B][FRP_ASTAR_MAX_PATH][11] or NULL: state(6), input(3), duration, pool index of every
                                  path node (path_nodes_, :308-320)                                                      */
    const double *retry_pt, *retry_vel; /* [B][3] each or NULL: the start of the REPEATED search.  getKinoPath starts the first
                                  search from the plan interpolated at t_cur with the planned thrust as acceleration, but the
                                  retry from the odometry state (nmpc_solver.cpp:151-153, 190-193); NULL: start_pt / start_vel */
} frp_nmpc_astar;

/* Bytes of device workspace for (B, grid, allocate_num): bit-packed map + per planner node pool, open-set heap, voxel hash. */
size_t frp_nmpc_astar_workspace_bytes(const frp_nmpc_astar *p);

/* NMPCSolver::getKinoPath's search (plan_manage/src/nmpc_solver.cpp:154-215) for B planners: KinodynamicAstar::search with
 * the external acceleration in the primitives, the retry on NO_PATH, getKinoTraj(Ts).  Asynchronous on `stream`. */
int frp_nmpc_astar_batch(const frp_nmpc_astar *p, void *workspace, size_t workspace_bytes, void *stream);

const char *frp_nmpc_version(void);
int frp_nmpc_device_count(void);

/* ---- (1) drop-in ABI: layout-compatible with the reference's generated headers ---- */
typedef struct {
    double xinit[9];             /* normal.h:156 */
    double x0[340];              /* normal.h:159 */
    double all_parameters[2600]; /* normal.h:162 */
    unsigned int num_of_threads; /* normal.h:165 */
} frp_forces_params;

typedef struct {
    double x[20][17]; /* x01 .. x20, normal.h:173-236 (contiguous) */
} frp_forces_output;

typedef struct {
    int it, it2opt;                                                                      /* normal.h:244-247 */
    double res_eq, res_ineq, rsnorm, rcompnorm, pobj, dobj, dgap, rdgap, mu, mu_aff, sigma; /* :249-280 */
    int lsit_aff, lsit_cc;                                                               /* :283-286 */
    double step_aff, step_cc, solvetime, fevalstime;                                     /* :289-298 */
} frp_forces_info;

typedef void (*frp_forces_extfunc)(double *x, double *y, double *lambda, double *params, double *pobj,
                                   double *g, double *c, double *Jeq, double *h, double *Jineq, double *H,
                                   int stage, int iterations, int threadID); /* normal.h:321 */

/* The callback pointer is an identity token here: the device kernels evaluate the shipped quadrotor
 * model themselves.  A non-NULL callback is probed once on the host and compared with the built-in
 * model; a different model makes the call fail with exit flag -11 (no host fallback exists). */
int FORCESNLPsolver_normal_solve(frp_forces_params *params, frp_forces_output *output, frp_forces_info *info,
                                 FILE *fs, frp_forces_extfunc evalextfunctions);
int FORCESNLPsolver_final_solve(frp_forces_params *params, frp_forces_output *output, frp_forces_info *info,
                                FILE *fs, frp_forces_extfunc evalextfunctions);

#ifdef __cplusplus
}
#endif
#endif /* FRP_NMPC_H */
