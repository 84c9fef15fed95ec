// Builds the forward-sweep step up piece by piece to see where the cycles go (one wavefront alone on its SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int M>
__device__ __forceinline__ double quad_rot(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x120 + (16 - 4 * M), 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x120 + (16 - 4 * M), 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
typedef __attribute__((address_space(3))) double ldouble;

template <int V>
__global__ void k(const double *in, double *out, double *sinkp)
{
    __shared__ double sh[20 * 309];
    ldouble *recs = (ldouble *)sh;
    for (int i = threadIdx.x; i < 20 * 309; i += 64) sh[i] = in[i & 127];
    __syncthreads();
    const int lane = threadIdx.x;
    double v = in[lane];
    double ts = in[lane + 1], hf = in[lane + 2], mu = in[lane + 3], m0 = in[lane + 4], m1 = in[lane + 5], m2 = in[lane + 6], m3 = in[lane + 7];
    ldouble *p0 = recs + (lane * 3) % 200, *p1 = recs + (lane * 5) % 200, *p2 = recs + (lane * 7) % 200, *p3 = recs + (lane * 11) % 200;
    ldouble *p4 = recs + (lane * 13) % 200, *p5 = recs + (lane * 17) % 200, *p6 = recs + 250 + (lane & 1);
    ldouble *w0 = recs + 260 + (lane >> 4), *w1 = recs + 270 + (lane & 15);
    long long t0 = clock64();
    for (int it = 0; it < 100; it++) {
        ldouble *q0 = p0, *q1 = p1, *q2 = p2, *q3 = p3, *q4 = p4, *q5 = p5, *q6 = p6, *x0 = w0, *x1 = w1;
        for (int kk = 0; kk < 20; kk++) {
            const double d = mfma4(ts, hf * v, 0.0);
            double acc = 0.0;
            if (V >= 1) {
                const double v1 = quad_rot<1>(v), v2 = quad_rot<2>(v), v3 = quad_rot<3>(v);
                acc = mfma4(m0, v, 0.0);
                acc = mfma4(m1, v1, acc);
                acc = mfma4(m2, v2, acc);
                acc = mfma4(m3, v3, acc);
            }
            const double r = -d - quad_rot<1>(d);
            const double du = r + quad_rot<2>(r);
            if (V >= 2) { x0[0] = du; x1[0] = v; }
            v = mfma4(mu, du, acc);
            if (V >= 3) { // operands of the next stage (single set: the loads must land before the next step)
                m0 = q0[309]; m1 = q1[309]; m2 = q2[309]; m3 = q3[309]; ts = q4[309]; mu = q5[309]; hf = q6[309];
            }
            if (V >= 2) { q0 += 309; q1 += 309; q2 += 309; q3 += 309; q4 += 309; q5 += 309; q6 += 309; x0 += 309; x1 += 309; }
        }
    }
    long long t1 = clock64();
    if (lane == 0) out[V] = (double)(t1 - t0) / 2000.0;
    if (v == 1234.5) sinkp[0] = v;
}
int main()
{
    double h[256];
    for (int i = 0; i < 256; i++) h[i] = 1e-3 * (1 + i % 7);
    double *d, *o, *s;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 32 * 8); hipMalloc(&s, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) { k<0><<<1, 64>>>(d, o, s); k<1><<<1, 64>>>(d, o, s); k<2><<<1, 64>>>(d, o, s); k<3><<<1, 64>>>(d, o, s); }
    double r[8];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const char *nm[] = {"du chain + closing MFMA", "+ 3 rotations and the 4-MFMA x part", "+ 2 LDS stores and pointer steps", "+ 7 operand gathers for the next stage (single set)"};
    for (int i = 0; i < 4; i++) printf("%-55s %7.1f cycles per stage\n", nm[i], r[i]);
    return 0;
}
