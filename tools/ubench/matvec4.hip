// Validates the 4x4x4-MFMA mat-vec building block used by the vector sweeps (layout + DPP quad rotation).
// hipcc --offload-arch=gfx950 -O3 matvec4.hip -o matvec4 && ./matvec4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int M>
__device__ __forceinline__ double quad_rot(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x120 + (16 - 4 * M), 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x120 + (16 - 4 * M), 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double matvec4(const d4 &A, double x, double c)
{
    const double x1 = quad_rot<1>(x), x2 = quad_rot<2>(x), x3 = quad_rot<3>(x);
    double d = mfma4(A[0], x, c);
    d = mfma4(A[1], x1, d);
    d = mfma4(A[2], x2, d);
    return mfma4(A[3], x3, d);
}

__global__ void test(const double *A /*16x16 row-major*/, const double *x, const double *c, double *y, double *cyc)
{
    const int l = threadIdx.x, qk = l >> 4, qI = (l >> 2) & 3, qj = l & 3, idx = 4 * qI + qk;
    d4 a;
    for (int m = 0; m < 4; m++) a[m] = A[(4 * qI + qj) * 16 + 4 * ((qI + m) & 3) + qk];
    double xv = x[idx], cv = c[idx];
    double d = matvec4(a, xv, cv);
    if (qj == 0) y[idx] = d;
    // latency of a chained mat-vec (output feeds the next input)
    long long t0 = clock64();
    double z = xv;
    for (int i = 0; i < 1000; i++) z = matvec4(a, z, cv) * 1e-3;
    long long t1 = clock64();
    if (l == 0) cyc[0] = (double)(t1 - t0) / 1000.0;
    if (z == 12345.0) y[0] = z;
}

int main()
{
    double hA[256], hx[16], hc[16], hy[16], ref[16];
    for (int i = 0; i < 256; i++) hA[i] = sin(1.0 + i * 0.37);
    for (int i = 0; i < 16; i++) { hx[i] = cos(0.3 * i + 0.1); hc[i] = 0.01 * i; }
    for (int i = 0; i < 16; i++) { ref[i] = hc[i]; for (int j = 0; j < 16; j++) ref[i] += hA[i * 16 + j] * hx[j]; }
    double *dA, *dx, *dc, *dy, *dcyc;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dx, sizeof hx); hipMalloc(&dc, sizeof hc); hipMalloc(&dy, sizeof hy); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
    test<<<1, 64>>>(dA, dx, dc, dy, dcyc);
    double cyc;
    hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost); hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; i++) err = fmax(err, fabs(hy[i] - ref[i]));
    printf("matvec4 max |y - ref| = %.3e  (%s)\n", err, err < 1e-13 ? "OK" : "WRONG");
    printf("chained matvec4 (+ one v_mul): %.1f cycles\n", cyc);
    return 0;
}
