// Layout probe + latency of v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 independent 4x4x4 blocks per instruction).
// hipcc --offload-arch=gfx950 -O3 mfma_f64_4x4.hip -o mfma_f64_4x4 && ./mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// one-hot probe: a = e_p, b = e_q  ->  which output lanes see the product
__global__ void probe(unsigned char *out)
{
    const int l = threadIdx.x;
    for (int p = 0; p < 64; p++)
        for (int q = 0; q < 64; q++) {
            const double a = (l == p) ? 1.0 : 0.0, b = (l == q) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(p * 64 + q) * 64 + l] = d != 0.0;
        }
}

template <int DEP>
__global__ void rate(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1e-3;
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (DEP) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (4.0 * iters);
    out[1 + l] = c0 + c1 + c2 + c3;
}

// dependent chain through a VALU op (the use pattern of a matvec chain): mfma -> v_add -> mfma ...
__global__ void rate_mixed(double *out, int iters)
{
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1e-3, c0 = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        b = c0 + 1e-3;
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        b = c0 + 1e-3;
    }
    long long t1 = clock64();
    if (l == 0) out[0] = (double)(t1 - t0) / (2.0 * iters);
    out[1 + l] = c0;
}

int main()
{
    unsigned char *d; hipMalloc(&d, 64 * 64 * 64);
    probe<<<1, 64>>>(d);
    std::vector<unsigned char> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, h.size(), hipMemcpyDeviceToHost);
    // for every output lane: the (p, q) pairs that contribute
    for (int l = 0; l < 64; l++) {
        printf("D lane %2d <-", l);
        for (int p = 0; p < 64; p++)
            for (int q = 0; q < 64; q++)
                if (h[(p * 64 + q) * 64 + l]) printf(" (a%d,b%d)", p, q);
        printf("\n");
    }
    double *o; hipMalloc(&o, 65 * sizeof(double));
    double r;
    rate<1><<<1, 64>>>(o, 10000); hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost); printf("mfma_f64_4x4x4 dependent chain: %.1f cycles/instr\n", r);
    rate<0><<<1, 64>>>(o, 10000); hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost); printf("mfma_f64_4x4x4 4 independent : %.1f cycles/instr\n", r);
    rate_mixed<<<1, 64>>>(o, 10000); hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost); printf("mfma_f64_4x4x4 -> v_add -> mfma chain: %.1f cycles per (mfma+add)\n", r);
    return 0;
}
