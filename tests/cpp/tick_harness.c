/* The whole on-device tick driven from plain C through the C-ABI only (HIP runtime for memory; no torch, no C++):
 *   coldstart -> reference -> tube -> corridor -> pack -> solve -> update      (NMPCSolver::solveNMPC, nmpc_solver.cpp:351-482)
 *   tick_harness <in.bin> <out.bin>
 * in.bin : int32 B, N, K, P, T;  doubles: path[K][3], cloud[P][3], plan[B][N+1][17], f_ext[B][3], time_offset[T][B]
 * out.bin: doubles plan[B][N+1][17] after T ticks, then int32 exitflag[B], iters[B], poly_index[B][N] of the last tick */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "frp_nmpc.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define FK(x) do { int rc_ = (x); if (rc_ != FRP_OK) { fprintf(stderr, "frp error %d at %s:%d\n", rc_, __FILE__, __LINE__); return 3; } } while (0)

static void *rd(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 8);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}
static void *up(const void *h, size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) exit(5);
    if (h && bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) exit(5);
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int *hdr = (int *)rd(f, 5 * sizeof(int));
    const int B = hdr[0], N = hdr[1], K = hdr[2], P = hdr[3], T = hdr[4], M = 30, F = 64;
    double *path = (double *)rd(f, sizeof(double) * 3 * K), *cloud = (double *)rd(f, sizeof(double) * 3 * P);
    double *plan = (double *)rd(f, sizeof(double) * (size_t)B * (N + 1) * 17), *fext = (double *)rd(f, sizeof(double) * 3 * B);
    double *toff = (double *)rd(f, sizeof(double) * (size_t)T * B);
    fclose(f);
    const size_t np = 10 + 4 * M;
    double *d_path = (double *)up(path, sizeof(double) * 3 * K), *d_cloud = (double *)up(cloud, sizeof(double) * 3 * P);
    double *d_plan = (double *)up(plan, sizeof(double) * (size_t)B * (N + 1) * 17), *d_f = (double *)up(fext, sizeof(double) * 3 * B);
    double *d_toff = (double *)up(toff, sizeof(double) * (size_t)T * B);
    double *d_ref = (double *)up(NULL, sizeof(double) * (size_t)B * N * 3), *d_yaw = (double *)up(NULL, sizeof(double) * (size_t)B * N);
    double *d_E = (double *)up(NULL, sizeof(double) * (size_t)B * N * 9);
    double *d_A = (double *)up(NULL, sizeof(double) * (size_t)B * N * F * 3), *d_b = (double *)up(NULL, sizeof(double) * (size_t)B * N * F);
    int *d_nf = (int *)up(NULL, sizeof(int) * (size_t)B * N), *d_pi = (int *)up(NULL, sizeof(int) * (size_t)B * N);
    double *d_xinit = (double *)up(NULL, sizeof(double) * 9 * B), *d_x0 = (double *)up(NULL, sizeof(double) * (size_t)B * N * 17);
    double *d_par = (double *)up(NULL, sizeof(double) * (size_t)B * N * np), *d_z = (double *)up(NULL, sizeof(double) * (size_t)B * N * 17);
    int *d_nfs = (int *)up(NULL, sizeof(int) * (size_t)B * N), *d_flag = (int *)up(NULL, sizeof(int) * B), *d_it = (int *)up(NULL, sizeof(int) * B);
    const size_t wsb = frp_nmpc_workspace_bytes(B, N, M);
    void *d_ws = up(NULL, wsb);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int *ones = (int *)malloc(sizeof(int) * B); /* exit flags start at 1: tick 0 cold-starts nobody */
    for (int i = 0; i < B; i++) ones[i] = 1;
    CK(hipMemcpy(d_flag, ones, sizeof(int) * B, hipMemcpyHostToDevice));

    frp_nmpc_options opt;
    frp_nmpc_default_options(&opt);
    frp_nmpc_reference rf = {.B = B, .N = N, .K = K, .kino_path = d_path, .mpc_output = d_plan, .Ts = 0.05, .pi = 3.1415926,
                             .ref_pos = d_ref, .ref_yaw = d_yaw};
    frp_nmpc_tube tb = {.B = B, .N = N, .mpc_output = d_plan, .mass = 0.74, .drag = 0.33, .ego_r = 0.27, .ego_h = 0.0425,
                        .noise = {0.5, 0.5, 0.5}, .epsilon = 0.06, .Ts = 0.05, .ellipsoid = d_E};
    frp_nmpc_corridor cr = {.B = B, .N = N, .F = F, .P = P, .cloud = d_cloud, .ref_pos = d_ref, .ref_yaw = d_yaw, .ellipsoid = d_E,
                            .bbox = {2.0, 2.0, 1.0}, .seed_len = 0.1, .inflation = 1.1, .offset_x = 0.0,
                            .poly_A = d_A, .poly_b = d_b, .poly_nfaces = d_nf, .poly_index = d_pi};
    frp_nmpc_pack pk = {.B = B, .N = N, .M = M, .NPOLY = N, .F = F, .mpc_output = d_plan, .external_acc = d_f, .ref_pos = d_ref,
                        .ref_yaw = d_yaw, .ellipsoid = d_E, .poly_A = d_A, .poly_b = d_b, .poly_nfaces = d_nf, .poly_index = d_pi,
                        .w_stage_wp = 15.0, .w_stage_input = 3.0, .w_input_rate = 80.0, .w_terminal_wp = 15.0, .w_terminal_input = 0.0,
                        .xinit = d_xinit, .x0 = d_x0, .params = d_par, .nfaces = d_nfs};
    frp_nmpc_batch bt = {.B = B, .N = N, .M = M, .MF = M, .model = FRP_MODEL_NORMAL, .xinit = d_xinit, .x0 = d_x0, .params = d_par,
                         .nfaces = d_nfs, .z = d_z, .exitflag = d_flag, .iters = d_it};
    for (int t = 0; t < T; t++) {
        rf.time_offset = d_toff + (size_t)t * B;
        FK(frp_nmpc_coldstart_batch(B, N, NULL, d_flag, 7.3, d_plan, st));
        FK(frp_nmpc_reference_batch(&rf, st));
        FK(frp_nmpc_tube_batch(&tb, st));
        FK(frp_nmpc_corridor_batch(&cr, st));
        FK(frp_nmpc_pack_batch(&pk, st));
        FK(frp_nmpc_solve_batch(&bt, &opt, d_ws, wsb, st));
        FK(frp_nmpc_update_batch(B, N, d_z, d_flag, d_plan, st));
    }
    CK(hipStreamSynchronize(st));
    int *flag = (int *)malloc(sizeof(int) * B), *it = (int *)malloc(sizeof(int) * B), *pi = (int *)malloc(sizeof(int) * (size_t)B * N);
    CK(hipMemcpy(plan, d_plan, sizeof(double) * (size_t)B * (N + 1) * 17, hipMemcpyDeviceToHost));
    CK(hipMemcpy(flag, d_flag, sizeof(int) * B, hipMemcpyDeviceToHost));
    CK(hipMemcpy(it, d_it, sizeof(int) * B, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pi, d_pi, sizeof(int) * (size_t)B * N, hipMemcpyDeviceToHost));
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 1;
    fwrite(plan, sizeof(double), (size_t)B * (N + 1) * 17, o);
    fwrite(flag, sizeof(int), B, o); fwrite(it, sizeof(int), B, o); fwrite(pi, sizeof(int), (size_t)B * N, o);
    fclose(o);
    return 0;
}
