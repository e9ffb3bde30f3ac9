// Issue cost of the FP64 instructions the Shepard sweeps are made of (experiment for sweeps.h: shepard_w), relative to v_fma_f64.
// Each kernel runs ITER iterations of 8 INDEPENDENT chains of one instruction kind; W waves per SIMD (grid = CUs * W workgroups of 256).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/valu_cost tools/exp/valu_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define ITER 4096

#define CHAIN8(OP)                                                                                      \
    for (int i = 0; i < ITER; i++) {                                                                    \
        OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)                                        \
    }

#define KERNEL(name, OP)                                                                                \
    __global__ void name(double *out, double seed)                                                      \
    {                                                                                                   \
        double x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        const double k = seed * 0.5;                                                                    \
        (void)k;                                                                                        \
        CHAIN8(OP)                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;             \
    }

#define OP_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define OP_RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define OP_SQRT(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define OP_CMP(x) asm volatile("v_cmp_eq_f64 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc");
// f32 detour: cvt f64->f32, v_rsq_f32, cvt f32->f64
#define OP_RSQ32(x) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x)); asm volatile("v_rsq_f32 %0, %0" : "+v"(f)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(f)); }
#define OP_CVT(x) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(f)); }
// packed f32 and plain f32 for scale
#define OP_FMA32(x) { float f = (float)0; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); }
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %0" : "+v"(*(int *)&x));
#define OP_SEL(x) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(*(int *)&x) : : "vcc");
#define OP_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));

KERNEL(k_fma, OP_FMA)
KERNEL(k_mul, OP_MUL)
KERNEL(k_add, OP_ADD)
KERNEL(k_rsq, OP_RSQ)
KERNEL(k_rcp, OP_RCP)
KERNEL(k_sqrt, OP_SQRT)
KERNEL(k_cmp, OP_CMP)
KERNEL(k_rsq32, OP_RSQ32)
KERNEL(k_cvt, OP_CVT)
KERNEL(k_mov, OP_MOV)
KERNEL(k_sel, OP_SEL)

typedef void (*kern_t)(double *, double);

int main()
{
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    struct { const char *name; kern_t k; int per; } ks[] = {
        {"v_fma_f64", k_fma, 1}, {"v_mul_f64", k_mul, 1}, {"v_add_f64", k_add, 1}, {"v_rsq_f64", k_rsq, 1}, {"v_rcp_f64", k_rcp, 1},
        {"v_sqrt_f64", k_sqrt, 1}, {"v_cmp_eq_f64", k_cmp, 1}, {"cvt+v_rsq_f32+cvt", k_rsq32, 1}, {"cvt f64->f32->f64", k_cvt, 1},
        {"v_mov_b32", k_mov, 1}, {"v_cndmask_b32", k_sel, 1},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz\n", pr.name, cus, pr.clockRate);
    for (int W = 1; W <= 4; W *= 2) {
        double base = 0;
        for (auto &kk : ks) {
            hipLaunchKernelGGL(kk.k, dim3(cus * W), dim3(256), 0, 0, out, 1.5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(kk.k, dim3(cus * W), dim3(256), 0, 0, out, 1.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns_per = ms * 1e6 / 5 / ((double)ITER * 8 * W);      // ns per wave-instruction slot on one SIMD
            if (base == 0) base = ns_per;
            printf("W=%d %-20s %.3f ns per instruction and SIMD  (%.2f x v_fma_f64; %.1f cycles at %.2f GHz)\n", W, kk.name, ns_per, ns_per / base,
                   ns_per * pr.clockRate * 1e-6, pr.clockRate * 1e-6);
        }
    }
    return 0;
}
