// Micro-benchmark + layout probe for v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950.
// hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_probe(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/, double *D /*16x16*/)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int DEP>
__global__ void mfma_rate(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (DEP) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * iters);
    out[1 + blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <int DEP>
__global__ void fma_rate(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-9, b = 1e-9;
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (DEP) {
            c0 = fma(a, c0, b); c0 = fma(a, c0, b); c0 = fma(a, c0, b); c0 = fma(a, c0, b);
            c0 = fma(a, c0, b); c0 = fma(a, c0, b); c0 = fma(a, c0, b); c0 = fma(a, c0, b);
        } else {
            c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b);
            c4 = fma(a, c4, b); c5 = fma(a, c5, b); c6 = fma(a, c6, b); c7 = fma(a, c7, b);
        }
    }
    long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (8.0 * iters);
    out[1 + blockIdx.x * 64 + l] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

// chip-level FP64 FMA throughput: many waves
__global__ void fma_chip(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-9, b = 1e-9;
    double c[16];
    for (int j = 0; j < 16; j++) c[j] = j;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = fma(a, c[j], b);
    double s = 0;
    for (int j = 0; j < 16; j++) s += c[j];
    out[blockIdx.x * blockDim.x + l] = s;
}
__global__ void mfma_chip(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    out[blockIdx.x * blockDim.x + l] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

int main()
{
    double hA[64], hB[64], hD[256];
    for (int i = 0; i < 64; i++) { hA[i] = 1 + (i * 7) % 13; hB[i] = 2 + (i * 5) % 11; }
    double *dA, *dB, *dD, *dout;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dout, 8 * (1 + 4096 * 256));
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    layout_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double ref = 0;
            for (int k = 0; k < 4; k++) ref += hA[i * 4 + k] * hB[k * 16 + j];
            double e = hD[i * 16 + j] - ref; if (e < 0) e = -e; if (e > maxerr) maxerr = e;
        }
    printf("layout probe: max |D - A*B| = %g  (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15])\n", maxerr);
    double r;
    const int it = 20000;
    mfma_rate<1><<<1, 64>>>(dout, it); hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost); printf("mfma_f64_16x16x4 dependent chain : %.1f cycles/instr (1 wave)\n", r);
    mfma_rate<0><<<1, 64>>>(dout, it); hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost); printf("mfma_f64_16x16x4 4 independent   : %.1f cycles/instr (1 wave)\n", r);
    fma_rate<1><<<1, 64>>>(dout, it); hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost); printf("v_fma_f64 dependent chain        : %.1f cycles/instr (1 wave)\n", r);
    fma_rate<0><<<1, 64>>>(dout, it); hipMemcpy(&r, dout, 8, hipMemcpyDeviceToHost); printf("v_fma_f64 8 independent          : %.1f cycles/instr (1 wave)\n", r);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); fma_chip<<<4096, 256>>>(dout, 4000); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("chip v_fma_f64 : %.1f TFLOP/s\n", 4096.0 * 256 * 4000 * 16 * 2 / (ms * 1e-3) / 1e12);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); mfma_chip<<<4096, 256>>>(dout, 4000); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("chip mfma_f64_16x16x4 : %.1f TFLOP/s\n", 4096.0 * 4 * 4000 * 4 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12);
    return 0;
}
