// tools/ubench/zc_read.hip -- how fast a kernel reads pinned HOST memory in place (the registered-buffer path of frp_nmpc_solve_batch_host):
// bytes per lane and load (8 / 16), workgroups, and hipHostMalloc vs malloc + hipHostRegister memory.   hipcc --offload-arch=gfx950 -O3 zc_read.hip -o zc_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
template <typename T>
__global__ void rd(const T *__restrict__ src, size_t n, double *out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        if constexpr (sizeof(T) == 8) acc += v; else acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}
template <typename T>
__global__ void cp(const T *__restrict__ src, T *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the gather's pattern: of every 130-double parameter row the ten leading slots + 18 A entries (doubles 0..27) and 6 b entries (100..105)
// MODE 0: output element -> input element, as frp_capi.hip's gather_inputs_kernel does; 1: lane = (row, slot) with 40 slots per row: the 64-byte
// lines that cover the two segments, read whole
template <int MODE>
__global__ void gather(const double *__restrict__ src, double *__restrict__ dst, size_t rows)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0) {
        for (size_t i = t0; i < rows * 34; i += stride) {
            const size_t r = i / 34, e = i - r * 34;
            dst[i] = src[r * 130 + (e < 28 ? e : e - 28 + 100)];
        }
    } else {
        for (size_t i = t0; i < rows * 48; i += stride) { // 48 slots per row: lines covering [0, 28) and [100, 106) at any alignment (<= 5 + 2 lines = 56 doubles; 48 covers the common cases: a sketch)
            const size_t r = i / 48, e = i - r * 48;
            const size_t base = (r * 130) & ~(size_t)7; // first line of the row
            const size_t a = e < 40 ? base + e : ((r * 130 + 100) & ~(size_t)7) + (e - 40);
            const double v = src[a];
            const long long off = (long long)a - (long long)(r * 130);
            if (off >= 0 && off < 28) dst[r * 34 + off] = v;
            else if (off >= 100 && off < 106) dst[r * 34 + 28 + (off - 100)] = v;
        }
    }
}
int main()
{
    const size_t bytes = 28u << 20;
    double *h1 = nullptr, *d_out = nullptr, *d_dst = nullptr;
    hipHostMalloc(&h1, bytes, hipHostMallocMapped);
    double *h2 = (double *)aligned_alloc(4096, bytes);
    memset(h1, 1, bytes); memset(h2, 1, bytes);
    hipHostRegister(h2, bytes, hipHostRegisterMapped);
    void *m1 = nullptr, *m2 = nullptr;
    hipHostGetDevicePointer(&m1, h1, 0); hipHostGetDevicePointer(&m2, h2, 0);
    hipMalloc(&d_out, 64); hipMalloc(&d_dst, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[2] = {"hipHostMalloc", "hipHostRegister"};
    void *ptrs[2] = {m1, m2};
    for (int w = 0; w < 2; w++)
        for (int blocks : {64, 256, 1024, 2048, 8192})
            for (int width : {8, 16}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (width == 8) hipLaunchKernelGGL(cp<double>, dim3(blocks), dim3(256), 0, 0, (const double *)ptrs[w], d_dst, bytes / 8);
                    else hipLaunchKernelGGL(cp<double2>, dim3(blocks), dim3(256), 0, 0, (const double2 *)ptrs[w], (double2 *)d_dst, bytes / 16);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                printf("%-16s blocks %5d  %2d B per lane: %.3f ms = %.1f GB/s\n", names[w], blocks, width, best, bytes / best * 1e-6);
            }
    { // the parameter rows of 4096 problems x 20 stages in the 30-row layout: 85 MB on the host, 22 MB of it live
        const size_t rows = 4096 * 20, pb = rows * 130 * 8;
        double *hp = (double *)aligned_alloc(4096, pb), *dg = nullptr;
        memset(hp, 1, pb);
        hipHostRegister(hp, pb, hipHostRegisterMapped);
        void *mp = nullptr; hipHostGetDevicePointer(&mp, hp, 0);
        hipMalloc(&dg, rows * 34 * 8);
        for (int mode = 0; mode < 2; mode++)
            for (int blocks : {256, 2048}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, (const double *)mp, dg, rows);
                    else hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, (const double *)mp, dg, rows);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                printf("gather mode %d blocks %4d: %.3f ms = %.1f GB/s of live bytes\n", mode, blocks, best, rows * 34 * 8 / best * 1e-6);
            }
    }
    { // the two parts of the gather at once: 11 MB contiguous (the plans) on one stream, the row segments on another
        const size_t rows = 4096 * 20, pb = rows * 130 * 8, zb = rows * 17 * 8;
        double *hp = (double *)aligned_alloc(4096, pb), *hz = (double *)aligned_alloc(4096, zb), *dg = nullptr, *dz = nullptr;
        memset(hp, 1, pb); memset(hz, 1, zb);
        hipHostRegister(hp, pb, hipHostRegisterMapped); hipHostRegister(hz, zb, hipHostRegisterMapped);
        void *mp = nullptr, *mz = nullptr; hipHostGetDevicePointer(&mp, hp, 0); hipHostGetDevicePointer(&mz, hz, 0);
        hipMalloc(&dg, rows * 34 * 8); hipMalloc(&dz, zb);
        hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
        hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
        for (int conc = 0; conc < 2; conc++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipDeviceSynchronize();
                hipEventRecord(e0, s1);
                if (conc) {
                    hipEventRecord(ea, s1); hipStreamWaitEvent(s2, ea, 0);
                    hipLaunchKernelGGL(cp<double>, dim3(256), dim3(256), 0, s2, (const double *)mz, dz, zb / 8);
                    hipLaunchKernelGGL(gather<0>, dim3(1024), dim3(256), 0, s1, (const double *)mp, dg, rows);
                    hipEventRecord(eb, s2); hipStreamWaitEvent(s1, eb, 0);
                } else {
                    hipLaunchKernelGGL(cp<double>, dim3(256), dim3(256), 0, s1, (const double *)mz, dz, zb / 8);
                    hipLaunchKernelGGL(gather<0>, dim3(1024), dim3(256), 0, s1, (const double *)mp, dg, rows);
                }
                hipEventRecord(e1, s1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("plans (11 MB contiguous) + parameter rows (22 MB live), %s: %.3f ms\n", conc ? "on two streams" : "one after the other", best);
        }
    }
    // the copy engine for comparison
    for (int w = 0; w < 2; w++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); hipMemcpyAsync(d_dst, w ? (void *)h2 : (void *)h1, bytes, hipMemcpyHostToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-16s hipMemcpyAsync: %.3f ms = %.1f GB/s\n", names[w], best, bytes / best * 1e-6);
    }
    return 0;
}
