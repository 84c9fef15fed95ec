// Cycle counts of the three Riccati sweeps of the LDS-resident solver, one wavefront alone on its SIMD, on a synthetic
// (diagonally dominant) record set: the stand-alone cost per stage of the serial chain, without barriers, other waves
// or the rest of the solve.  Built against the product source itself.
#include "../../forces_resilient_planner_amd/csrc/frp_ipm_lds.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace frp::lr;
__global__ __launch_bounds__(64) void timing_kernel(int N, int reps, const double *init, long long *out, double *dump)
{
    __shared__ double s_recs[20 * RS];
    __shared__ double s_xs[X_TOTAL + 64];
    ldouble *recs = (ldouble *)s_recs, *xs = (ldouble *)s_xs;
    const int lane = threadIdx.x;
    for (int i = lane; i < 20 * RS; i += 64) recs[i] = init[i];
    for (int i = lane; i < X_TOTAL + 64; i += 64) xs[i] = 0.0;
    if (lane == 0) { xs[X_C0] = 0.0; xs[X_C1] = 1.0; }
    __syncthreads();
    long long t[5] = {0, 0, 0, 0, 0};
    int fails = 0;
    for (int r = 0; r < reps; r++) {
        // the factorisation overwrites the Hessian part of the records with P: restore it (untimed)
        for (int i = lane; i < 20 * RS; i += 64) { const int o = i % RS; if (o >= R_PHID && o < R_PD) recs[i] = init[i]; }
        __syncthreads();
        long long a = clock64();
        fails += sweep_factor<false>(recs, xs, N, 1.0);
        long long b = clock64(); t[0] += b - a; a = b;
        sweep_forward(recs, xs, N);
        b = clock64(); t[1] += b - a; a = b;
        sweep_backvec<false>(recs, xs, N, 0.01);
        b = clock64(); t[2] += b - a; a = b;
        sweep_forward(recs, xs, N);
        b = clock64(); t[3] += b - a;
    }
    if (lane == 0) { for (int i = 0; i < 4; i++) out[i] = t[i]; out[4] = fails; }
    // results of the last repetition, field by field (layout-independent): dz (17), p (13), kbar (4) per stage
    __syncthreads();
    for (int k = 0; k < N; k++) {
        if (lane < 17) dump[k * 34 + lane] = recs[k * RS + R_DZ + lane];
        if (lane < 13) dump[k * 34 + 17 + lane] = recs[k * RS + R_PV + lane];
        if (lane < 4) dump[k * 34 + 30 + lane] = recs[k * RS + R_T + 16 * lane + 13];
    }
}
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 20, reps = 50;
    std::vector<double> h(20 * RS, 0.0);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (int k = 0; k < 20; k++) {
        double *r = h.data() + k * RS;
        for (int i = 0; i < 64; i++) r[R_LIN + i] = 0.05 * rnd();
        if (k == 19) for (int i = 0; i < 64; i++) r[R_LIN + i] = 0.0;
        for (int i = 0; i < 17; i++) { r[R_PHID + i] = 10.0 + rnd(); r[R_PHI + i] = rnd(); r[R_PHIB + i] = rnd(); r[R_PHIC + i] = rnd(); }
        for (int i = 0; i < 3; i++) { r[R_PHIPOS + 4 * i] = 1.0; r[R_CB + i] = rnd(); r[R_CC + i] = rnd(); }
        for (int i = 0; i < 45; i++) r[R_HD + i] = 0.01 * rnd();
        r[R_HC] = -0.5; r[R_ZERO] = 0.0; r[R_ONE] = 1.0; r[R_DT] = 0.05;
    }
    double *d, *dd; long long *o;
    hipMalloc(&d, h.size() * 8); hipMalloc(&o, 64); hipMalloc(&dd, 20 * 34 * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; pass++) hipLaunchKernelGGL(timing_kernel, dim3(1), dim3(64), 0, 0, N, reps, d, o, dd);
    long long r[5];
    hipMemcpy(r, o, 40, hipMemcpyDeviceToHost);
    const char *nm[4] = {"factor", "forward (predictor)", "backvec", "forward (corrector)"};
    for (int i = 0; i < 4; i++) printf("%-20s %8.0f cycles per sweep = %6.0f per stage\n", nm[i], (double)r[i] / reps, (double)r[i] / reps / N);
    printf("pivot failures: %lld of %d sweeps\n", r[4], reps);
    if (argc > 2) { // dump for comparisons between builds
        std::vector<double> hd(20 * 34);
        hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(argv[2], "w");
        for (int k = 0; k < N; k++) for (int i = 0; i < 34; i++) fprintf(f, "%d %d %.17g\n", k, i, hd[k * 34 + i]);
        fclose(f);
    }
    return 0;
}
