// Accuracy of the gfx950 v_rcp_f64 seed and of one / two Newton steps on it (decides how many steps fast_rcp needs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *o0, double *o1, double *o2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    o0[i] = r;
    double e = fma(-v, r, 1.0);
    r = fma(r, e, r);
    o1[i] = r;
    e = fma(-v, r, 1.0);
    r = fma(r, e, r);
    o2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        hx[i] = ldexp(1.0 + u, (int)(s % 81) - 40) * ((i & 1) ? 1.0 : 1.0);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)hx[i];
        e0 = fmax(e0, (double)fabsl((h0[i] - t) / t)); e1 = fmax(e1, (double)fabsl((h1[i] - t) / t)); e2 = fmax(e2, (double)fabsl((h2[i] - t) / t));
    }
    printf("v_rcp_f64 max relative error: seed %.3e, one Newton step %.3e, two steps %.3e (eps = 1.11e-16)\n", e0, e1, e2);
    return 0;
}
