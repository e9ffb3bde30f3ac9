// v_mad_u32_u16 with op_sel: LDS byte address of table entry lo16(w) / hi16(w) in ONE instruction each (experiment for ColsIdx::decode)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ unsigned mad_lo16(unsigned w, unsigned base)
{
    unsigned d;
    asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[0,0,0,0]" : "=v"(d) : "v"(w), "v"(base));
    return d;
}
__device__ __forceinline__ unsigned mad_hi16(unsigned w, unsigned base)
{
    unsigned d;
    asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(base));
    return d;
}
__global__ void k(const unsigned *w, double *out, int n, int U)
{
    extern __shared__ double tab[];
    for (int i = threadIdx.x; i < 2 * U; i += blockDim.x) tab[i] = 1000.0 * (i / U) + (i % U);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned base = (unsigned)(unsigned long long)(lds_cdouble *)(tab + U);      // table 1
    const unsigned x = w[i];
    out[2 * i] = *(lds_cdouble *)(unsigned long long)mad_lo16(x, base);
    out[2 * i + 1] = *(lds_cdouble *)(unsigned long long)mad_hi16(x, base);
}
int main()
{
    const int n = 4096, U = 1001;
    unsigned hw[n]; double ho[2 * n];
    for (int i = 0; i < n; i++) hw[i] = ((unsigned)((i * 7919u) % U) << 16) | (unsigned)((i * 104729u) % U);
    unsigned *dw; double *dout;
    (void)hipMalloc(&dw, sizeof hw); (void)hipMalloc(&dout, sizeof ho);
    (void)hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 2 * U * sizeof(double), 0, dw, dout, n, U);
    (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        if (ho[2 * i] != 1000.0 + (hw[i] & 0xffff)) bad++;
        if (ho[2 * i + 1] != 1000.0 + (hw[i] >> 16)) bad++;
    }
    printf("v_mad_u32_u16 decode: %d mismatches of %d\n", bad, 2 * n);
    return bad != 0;
}
