// measures the relative error of v_rsq_f64 and of 1 / 2 Newton refinements (experiment for sweeps.h: inv_dist_pow)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const double *x, double *y0, double *y1, double *y2, double *y3, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = x[i], y = __builtin_amdgcn_rsq(s);
    y0[i] = y;
    {   // one third-order (Halley-type) step: y (1 + e/2 + 3 e^2 / 8), e = 1 - s y^2
        const double t = s * y, e = __fma_rn(-t, y, 1.0), c = __fma_rn(0.375, e, 0.5), ce = c * e;
        y3[i] = __fma_rn(y, ce, y);
    }
    for (int it = 0; it < 2; it++) {
        double t = s * y, e = __fma_rn(-t, y, 1.0);
        y = __fma_rn(0.5 * y, e, y);
        if (it == 0) y1[i] = y; else y2[i] = y;
    }
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n), h(n);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = exp(-20.0 + 40.0 * ((st >> 11) * (1.0 / 9007199254740992.0))); }
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; i++) {
        long double r = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, fabs((double)((a[i] - r) / r))); e1 = fmax(e1, fabs((double)((b[i] - r) / r))); e2 = fmax(e2, fabs((double)((c[i] - r) / r))); e3 = fmax(e3, fabs((double)((h[i] - r) / r)));
    }
    printf("v_rsq_f64 max rel err %.3e ; +1 Newton %.3e ; +2 Newton %.3e ; one third-order step %.3e\n", e0, e1, e2, e3);
    return 0;
}
