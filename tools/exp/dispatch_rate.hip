// How fast does the chip start workgroups?  Empty / short kernels over grids of one-wave and four-wave workgroups (experiment for the
// one-wave-per-environment step kernel: 8192 workgroups per launch).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(int *out) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1; }
// ~`spin` dependent FP64 instructions per wave: a workgroup that lives for a while
__global__ void k_work(double *out, int spin)
{
    double x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = __fma_rn(x, 1.0000001, 0.5);
    if (x == 12345.0) out[0] = x;
}
int main()
{
    int *d; double *dd;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&dd, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grids[] = {1024, 8192, 65536, 524288};
    for (int bs : {64, 256}) for (int g : grids) {
        hipLaunchKernelGGL(k_empty, dim3(g), dim3(bs), 0, 0, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_empty, dim3(g), dim3(bs), 0, 0, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %6d workgroups of %3d threads: %.2f us per launch = %.1f workgroups / us\n", g, bs, ms * 1000 / 20, g / (ms * 1000 / 20));
    }
    for (int spin : {2000, 20000}) for (int g : {1024, 6144, 8192, 16384}) {
        hipLaunchKernelGGL(k_work, dim3(g), dim3(64), 0, 0, dd, spin);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_work, dim3(g), dim3(64), 0, 0, dd, spin);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%5d dependent FMAs per wave, %6d one-wave workgroups: %.2f us per launch\n", spin, g, ms * 1000 / 10);
    }
    return 0;
}
