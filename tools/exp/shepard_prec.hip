// relative error of the Shepard weight s^(-P/2) computed from v_rsq_f64: (old) one third-order step on y, then y^P by multiplication;
// (new) y0^P times the series of (1-e)^(-P/2) -- experiment for sweeps.h: shepard_w(PowC<P>)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
template <int P>
__global__ void k(const double *x, double *wo, double *wn, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = x[i], y0 = __builtin_amdgcn_rsq(s);
    {
        double y = y0;
        const double t = s * y, e = __fma_rn(-t, y, 1.0), c = __fma_rn(0.375, e, 0.5), ce = c * e;
        y = __fma_rn(y, ce, y);
        double w = y;
        for (int j = 1; j < P; j++) w = w * y;
        wo[i] = w;
    }
    {
        constexpr double a = 0.5 * P, a2 = 0.5 * a * (a + 1.0);
        const double y2 = y0 * y0;
        const double e = __fma_rn(-s, y2, 1.0);
        double yp;
        if (P == 3) yp = y2 * y0;
        else if (P == 5) { const double y4 = y2 * y2; yp = y4 * y0; }
        else { yp = y0; for (int j = 1; j < P; j++) yp = yp * y0; }
        const double c = __fma_rn(a2, e, a);
        const double ce = c * e;
        wn[i] = __fma_rn(yp, ce, yp);
    }
}
template <int P>
void run(const std::vector<double> &x, double *dx, double *d0, double *d1)
{
    const int n = (int)x.size();
    std::vector<double> a(n), b(n);
    hipLaunchKernelGGL(k<P>, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    (void)hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    double eo = 0, en = 0, so = 0, sn = 0;
    for (int i = 0; i < n; i++) {
        const long double r = powl((long double)x[i], -0.5L * P);
        const double ro = fabs((double)((a[i] - r) / r)), rn = fabs((double)((b[i] - r) / r));
        eo = fmax(eo, ro); en = fmax(en, rn); so += ro; sn += rn;
    }
    printf("P=%d: old max rel err %.3e (mean %.3e) ; new max %.3e (mean %.3e)\n", P, eo, so / n, en, sn / n);
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = exp(-30.0 + 34.0 * ((st >> 11) * (1.0 / 9007199254740992.0))); }
    double *dx, *d0, *d1;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    run<3>(x, dx, d0, d1); run<5>(x, dx, d0, d1); run<4>(x, dx, d0, d1); run<8>(x, dx, d0, d1);
    return 0;
}
