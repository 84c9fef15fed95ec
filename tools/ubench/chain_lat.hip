// Latency of the dependent instruction chains the Riccati vector sweeps are made of (one wavefront alone on its SIMD).
// hipcc --offload-arch=gfx950 -O3 chain_lat.hip -o chain_lat && ./chain_lat
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
typedef double d4 __attribute__((ext_vector_type(4)));

#define TIME(name, ...)                                                                                    \
    {                                                                                                      \
        double z = x0;                                                                                     \
        long long t0 = clock64();                                                                          \
        _Pragma("unroll 4") for (int i = 0; i < 2000; i++) { __VA_ARGS__; }                                        \
        long long t1 = clock64();                                                                          \
        if (threadIdx.x == 0) out[n] = (double)(t1 - t0) / 2000.0;                                         \
        n++;                                                                                               \
        sink += z;                                                                                         \
    }

__global__ void k(const double *in, double *out, double *sinkp)
{
    const double x0 = in[threadIdx.x], a = in[64 + threadIdx.x], c = in[128 + threadIdx.x];
    const bool f0 = (threadIdx.x & 12) == 0, f13 = threadIdx.x == 29;
    double sink = 0.0;
    int n = 0;
    TIME("v_mul_f64 chain", z = z * a)
    TIME("v_fma_f64 chain", z = __builtin_fma(z, a, c))
    TIME("v_add_f64 chain", z = z + c)
    TIME("mfma4 chain through C", z = mfma4(a, c, z))
    TIME("mfma4 chain through B", z = mfma4(a, z, c))
    TIME("mfma4 chain through A", z = mfma4(z, a, c))
    TIME("mfma4(B) + v_add", z = mfma4(a, z, c) + c)
    TIME("quad_rot<1> + add", z = z + quad_rot<1>(z))
    TIME("quad reduce (2 rot + 2 add)", { const double r = z + quad_rot<1>(z); z = r + quad_rot<2>(r); })
    TIME("select (2 cndmask) + mul", { const double h = z * a; z = f0 ? h : (f13 ? 1.0 : z); })
    TIME("mfma4(B) + quad reduce", { const double d = mfma4(a, z, 0.0); const double r = -d - quad_rot<1>(d); z = r + quad_rot<2>(r); })
    TIME("forward stage chain: mul, select, mfma, reduce, mfma", {
        const double h = z * a; const double v1 = f0 ? h : (f13 ? 1.0 : z);
        const double d = mfma4(a, v1, 0.0); const double r = -d - quad_rot<1>(d); const double du = r + quad_rot<2>(r);
        z = mfma4(c, du, a); })
    TIME("old forward chain: 2 x (select, 3 rot, 2x2 mfma, add)", {
        const double h = z * a; const double v1 = f0 ? h : (f13 ? 1.0 : z);
        double x1 = quad_rot<1>(v1); double x2 = quad_rot<2>(v1); double x3 = quad_rot<3>(v1);
        double d0 = mfma4(a, v1, 0.0); double d1 = mfma4(c, x1, 0.0); d0 = mfma4(c, x2, d0); d1 = mfma4(a, x3, d1);
        const double du = -(d0 + d1); const double v2 = f0 ? du : (f13 ? 1.0 : z);
        x1 = quad_rot<1>(v2); x2 = quad_rot<2>(v2); x3 = quad_rot<3>(v2);
        d0 = mfma4(a, v2, 0.0); d1 = mfma4(c, x1, 0.0); d0 = mfma4(c, x2, d0); d1 = mfma4(a, x3, d1);
        z = d0 + d1; })
    TIME("ds_write + ds_read round trip", {
        __shared__ double sh[64]; sh[threadIdx.x] = z; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); z = sh[threadIdx.x ^ 1] ; })
    TIME("v_readlane + v_mov (uniform round trip)", z = __shfl(z, 5) + c)
    if (sink == 1234.5) sinkp[0] = sink;
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
    const char *names[] = {"v_mul_f64 chain", "v_fma_f64 chain", "v_add_f64 chain", "mfma4 chain through C", "mfma4 chain through B", "mfma4 chain through A",
                           "mfma4(B) + v_add", "quad_rot<1> + add", "quad reduce (2 rot + 2 add)", "select (2 cndmask) + mul", "mfma4(B) + quad reduce",
                           "forward stage chain: mul, select, mfma, reduce, mfma", "old forward chain: 2 x (select, 3 rot, 2x2 mfma, add)",
                           "ds_write + ds_read round trip", "v_readlane + v_add (uniform round trip)"};
    for (int i = 0; i < 15; i++) printf("%-62s %7.1f cycles\n", names[i], r[i]);
    return 0;
}
