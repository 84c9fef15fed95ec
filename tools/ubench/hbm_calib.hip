// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS solver's access width (8 B per lane, coalesced
// dwordx2 loads / stores), as MI355X_MICROARCH.md asks ("calibrate on a known byte count in your own access pattern").
// read_kernel streams a buffer far larger than the 256 MiB Infinity Cache once; write_kernel writes one.
// hipcc --offload-arch=gfx950 -O3 hbm_calib.hip -o hbm_calib && rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./hbm_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_kernel(const double *__restrict__ x, size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (; i < n; i += stride) acc += x[i];
    if (acc == 12345.678) out[0] = acc;
}
__global__ void write_kernel(double *__restrict__ x, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = (double)i;
}

int main()
{
    const size_t n = (size_t)1 << 28; // 2 GiB of doubles
    double *x, *o;
    if (hipMalloc(&x, n * sizeof(double)) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
    hipMemset(x, 0, n * sizeof(double));
    for (int rep = 0; rep < 3; rep++) {
        write_kernel<<<4096, 256>>>(x, n);
        read_kernel<<<4096, 256>>>(x, n, o);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %zu\n", n * sizeof(double));
    return 0;
}
