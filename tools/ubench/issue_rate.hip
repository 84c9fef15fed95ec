// Issue cost per instruction class for ONE wavefront alone on its SIMD (independent instructions, 8 chains), and how much
// VALU work hides under a 4x4x4 FP64 MFMA.  hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int rot(int v) { return __builtin_amdgcn_mov_dpp(v, 0x124, 0xF, 0xF, true); }
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define TIME(...)                                                                   \
    {                                                                               \
        long long t0 = clock64();                                                   \
        for (int i = 0; i < 1000; i++) { __VA_ARGS__; }                             \
        long long t1 = clock64();                                                   \
        if (threadIdx.x == 0) out[n] = (double)(t1 - t0) / 1000.0;                  \
        n++;                                                                        \
    }
__global__ void k(const double *in, double *out, double *sinkp)
{
    __shared__ double sh[1024];
    double a[8], s = 0.0;
    int q[8];
    const double c = in[64 + threadIdx.x];
    for (int j = 0; j < 8; j++) { a[j] = in[threadIdx.x + j]; q[j] = (int)threadIdx.x * (j + 3); sh[threadIdx.x + 64 * j] = a[j]; }
    const bool f = threadIdx.x & 1;
    int n = 0;
#define F64(j) a[j] = __builtin_fma(a[j], c, c);
    TIME(REP8(F64))                                            // 0: 8 independent v_fma_f64
#define ADD32(j) q[j] = q[j] + (int)threadIdx.x;
    TIME(REP8(ADD32))                                          // 1: 8 independent v_add_u32
#define DPP(j) q[j] = rot(q[j]);
    TIME(REP8(DPP))                                            // 2: 8 independent v_mov_b32_dpp
#define CND(j) q[j] = f ? q[j] : q[(j + 1) & 7];
    TIME(REP8(CND))                                            // 3: 8 v_cndmask_b32 (loosely dependent)
#define LDS(j) a[j] += sh[(q[j] & 511) + j];
    TIME(REP8(LDS))                                            // 4: 8 x (v_and + ds_read_b64 + v_add_f64)
#define MF(j) a[j] = mfma4(c, c, a[j]);
    TIME(REP8(MF))                                             // 5: 8 independent mfma4
#define MFV(j) a[j] = mfma4(c, c, a[j]); q[j] = q[j] + (int)threadIdx.x; q[j] = rot(q[j]); q[j] ^= 5;
    TIME(REP8(MFV))                                            // 6: 8 x (mfma4 + 3 32-bit VALU)
#define MFV6(j) a[j] = mfma4(c, c, a[j]); q[j] = q[j] + (int)threadIdx.x; q[j] = rot(q[j]); q[j] ^= 5; q[j] += 77; q[j] = rot(q[j]); q[j] ^= 9;
    TIME(REP8(MFV6))                                           // 7: 8 x (mfma4 + 6 32-bit VALU)
#define MFD(j) a[j] = mfma4(c, c, a[j]); a[(j + 4) & 7] = __builtin_fma(a[(j + 4) & 7], c, c);
    TIME(REP8(MFD))                                            // 8: 8 x (mfma4 + 1 v_fma_f64 on another chain)
    for (int j = 0; j < 8; j++) s += a[j] + q[j];
    if (s == 1234.5) sinkp[0] = s;
}
int main()
{
    double h[192];
    for (int i = 0; i < 192; i++) h[i] = 1e-3 * (1 + i % 7);
    double *d, *o, *s;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 32 * 8); hipMalloc(&s, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, s); k<<<1, 64>>>(d, o, s);
    double r[32];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    const char *nm[] = {"8 v_fma_f64", "8 v_add_u32", "8 v_mov_b32_dpp", "8 v_cndmask_b32", "8 (v_and + ds_read_b64 + v_add_f64)", "8 mfma_f64_4x4x4",
                        "8 (mfma4 + 3 32-bit VALU)", "8 (mfma4 + 6 32-bit VALU)", "8 (mfma4 + v_fma_f64)"};
    for (int i = 0; i < 9; i++) printf("%-40s %7.1f cycles per group = %5.1f per unit\n", nm[i], r[i], r[i] / 8);
    return 0;
}
