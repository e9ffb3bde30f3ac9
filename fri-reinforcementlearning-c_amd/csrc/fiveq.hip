// fiveq.hip -- batched Q-value kernels: FIVE_vag_concl, FIVE_vag_concl_weight, frirl_get_best_action.
//
// One workgroup per environment (grid.x = E); each kernel is one or two streaming sweeps over
// that environment's SoA slab (sweeps.h).  HBM traffic per rule: vag_concl 8*(nant+1) B,
// vag_concl_weight 8*(2*nant+1) B read + 8 B written, get_best_action 8*(nant+1) B for all A actions.
#include "sweeps.h"

namespace frirl {

template <int NANT, int BLOCK, bool IDX>
__global__ __launch_bounds__(BLOCK) void vag_concl_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                           const double *__restrict__ rb, const uint16_t *__restrict__ uidx,
                                                           const int32_t *__restrict__ nrules, int maxR,
                                                           int p, const double *__restrict__ x, double *__restrict__ conc,
                                                           uint32_t *__restrict__ hit)
{
    extern __shared__ double tab_s[];
    const int e = blockIdx.x;
    const int R = nrules[e];
    __shared__ double q_s[NANT];
    __shared__ BlockRed<BLOCK> red;
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];
    const double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const auto cols = ColsSel<IDX>::make(base, uidx + (IDX ? (size_t)e * NANT * maxR : 0), tab_s, maxR, U);
    const QResult res = sweep_q<NANT, BLOCK>(cols, base + (size_t)NANT * maxR, R, q, p, red);
    if (threadIdx.x == 0) {
        hit[e] = res.hit;
        // exact hit -> its consequent (FIVEVagConcl.c:94-99); else vagc / ws (:302,347)
        conc[e] = (res.hit != FRIRL_HIP_NO_HIT) ? base[(size_t)NANT * maxR + res.hit] : res.vagc / res.ws;
    }
}

template <int NANT, int BLOCK, bool IDX>
__global__ __launch_bounds__(BLOCK) void vag_concl_weight_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                  const double *__restrict__ rb, const uint16_t *__restrict__ uidx,
                                                                  const int32_t *__restrict__ nrules,
                                                                  int maxR, int p, const double *__restrict__ x,
                                                                  double *__restrict__ weights, uint32_t *__restrict__ hit)
{
    extern __shared__ double tab_s[];
    const int e = blockIdx.x;
    const int R = nrules[e];
    __shared__ double q_s[NANT];
    __shared__ BlockRed<BLOCK> red;
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];
    const double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const auto cols = ColsSel<IDX>::make(base, uidx + (IDX ? (size_t)e * NANT * maxR : 0), tab_s, maxR, U);
    const QResult res = sweep_q<NANT, BLOCK>(cols, base + (size_t)NANT * maxR, R, q, p, red);
    if (threadIdx.x == 0) hit[e] = res.hit;
    // exact hit: the reference returns the index and leaves weights[] untouched (FIVEVagConclWeight.c:67-69)
    if (res.hit == FRIRL_HIP_NO_HIT) sweep_weights<NANT, BLOCK>(cols, R, q, p, res.ws, weights + (size_t)e * maxR);
}

template <int NANT, int AMAX, int BLOCK, bool IDX>
__global__ __launch_bounds__(BLOCK) void get_best_action_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                 const double *__restrict__ rb, const uint16_t *__restrict__ uidx,
                                                                 const int32_t *__restrict__ nrules,
                                                                 int maxR, int p, const double *__restrict__ states,
                                                                 const double *__restrict__ action_ve, int A,
                                                                 double *__restrict__ actconc, int32_t *__restrict__ best)
{
    constexpr int NS = NANT - 1;
    const int e = blockIdx.x;
    const int R = nrules[e];
    extern __shared__ double tab_s[];
    __shared__ double q_s[NS];
    __shared__ GbaScratch<AMAX, BLOCK> gs;
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NS) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, states[(size_t)e * NS + threadIdx.x]);
    if ((int)threadIdx.x < A) gs.ave[threadIdx.x] = action_ve[threadIdx.x];
    __syncthreads();
    double q[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) q[k] = q_s[k];
    const double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const double *qcol = base + (size_t)NANT * maxR;
    const auto cols = ColsSel<IDX>::make(base, uidx + (IDX ? (size_t)e * NANT * maxR : 0), tab_s, maxR, U);
    int b;
    if constexpr (AMAX == 24) {                 // 9..24 actions, 256 threads: every action in registers (sweep_gba_many)
        __shared__ BlockRed<BLOCK> red;
        double dummy[NANT] = {};
        b = sweep_gba_many<NANT, AMAX, BLOCK, false>(cols, qcol, R, q, dummy, p, A, gs, red, nullptr);
    } else if constexpr (AMAX > 8) {
        __shared__ BlockRed<BLOCK> red;
        double dummy[NANT] = {};
        b = sweep_gba_wide<NANT, 8, AMAX, BLOCK, false>(cols, qcol, R, q, dummy, p, A, gs, red, nullptr);
    } else {
        b = sweep_gba<NANT, AMAX, BLOCK>(cols, qcol, R, q, p, A, gs);
    }
    if ((int)threadIdx.x < A) actconc[(size_t)e * A + threadIdx.x] = gs.actconc[threadIdx.x];
    if (threadIdx.x == 0) best[e] = b;
}

template <template <int, int> class K>
struct Dummy {};

}  // namespace frirl

using namespace frirl_host;

#define FRIRL_NANT_CASES(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)
#define FRIRL_GBA_NANT_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)

static int eff_p(const frirl_hip_tables *t, int p) { return p > 0 ? p : t->nant; }   // FIVEInit.c:89-93

extern "C" int five_hip_vag_concl(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *x, double *conc,
                                  uint32_t *hit, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!x || !conc || !hit) { set_error("five_hip_vag_concl: NULL x/conc/hit"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);
    const bool big = b->E < 256;    // few rule bases: 1024-thread workgroups shorten each sweep
    const bool idx = !big && frirl::use_uidx(t, b);
    const size_t tab = idx ? sizeof(double) * t->nant * (size_t)t->U : 0;
    switch (t->nant) {
#define M(N)                                                                                                                   \
    case N:                                                                                                                    \
        if (big) hipLaunchKernelGGL((frirl::vag_concl_kernel<N, 1024, false>), dim3(b->E), dim3(1024), 0, s, t->u, t->ve, t->U, b->rb, b->uidx, \
                                    b->nrules, b->maxR, eff_p(t, p), x, conc, hit);                                            \
        else if (idx) hipLaunchKernelGGL((frirl::vag_concl_kernel<N, 256, true>), dim3(b->E), dim3(256), tab, s, t->u, t->ve, t->U, b->rb, b->uidx, \
                                b->nrules, b->maxR, eff_p(t, p), x, conc, hit);                                                \
        else hipLaunchKernelGGL((frirl::vag_concl_kernel<N, 256, false>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->uidx,       \
                                b->nrules, b->maxR, eff_p(t, p), x, conc, hit);                                                \
        break;
        FRIRL_NANT_CASES(M)
#undef M
        default: set_error("five_hip_vag_concl: unsupported nant=%d", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("five_hip_vag_concl");
}

extern "C" int five_hip_vag_concl_weight(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *x,
                                         double *weights, uint32_t *hit, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!x || !weights || !hit) { set_error("five_hip_vag_concl_weight: NULL x/weights/hit"); return FRIRL_HIP_EINVAL; }
    if (reinterpret_cast<uintptr_t>(weights) & 15) { set_error("five_hip_vag_concl_weight: weights must be 16-byte aligned"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);
    const bool big = b->E < 256;
    const bool idx = !big && frirl::use_uidx(t, b);
    const size_t tab = idx ? sizeof(double) * t->nant * (size_t)t->U : 0;
    switch (t->nant) {
#define M(N)                                                                                                                          \
    case N:                                                                                                                           \
        if (big) hipLaunchKernelGGL((frirl::vag_concl_weight_kernel<N, 1024, false>), dim3(b->E), dim3(1024), 0, s, t->u, t->ve, t->U, b->rb, b->uidx, \
                                    b->nrules, b->maxR, eff_p(t, p), x, weights, hit);                                                \
        else if (idx) hipLaunchKernelGGL((frirl::vag_concl_weight_kernel<N, 256, true>), dim3(b->E), dim3(256), tab, s, t->u, t->ve, t->U, b->rb, b->uidx, \
                                b->nrules, b->maxR, eff_p(t, p), x, weights, hit);                                                    \
        else hipLaunchKernelGGL((frirl::vag_concl_weight_kernel<N, 256, false>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->uidx,       \
                                b->nrules, b->maxR, eff_p(t, p), x, weights, hit);                                                    \
        break;
        FRIRL_NANT_CASES(M)
#undef M
        default: set_error("five_hip_vag_concl_weight: unsupported nant=%d", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("five_hip_vag_concl_weight");
}

template <int N>
static void launch_gba(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *states, const double *action_ve,
                       int A, double *actconc, int32_t *best, hipStream_t s)
{
    const bool big = b->E < 256;
    const bool idx = !big && frirl::use_uidx(t, b);
    const size_t tab = idx ? sizeof(double) * t->nant * (size_t)t->U : 0;
#define L(AMAX)                                                                                                                          \
    do {                                                                                                                                 \
        if (big) hipLaunchKernelGGL((frirl::get_best_action_kernel<N, AMAX, 1024, false>), dim3(b->E), dim3(1024), 0, s, t->u, t->ve, t->U, \
                                    b->rb, b->uidx, b->nrules, b->maxR, p, states, action_ve, A, actconc, best);                         \
        else if (idx) hipLaunchKernelGGL((frirl::get_best_action_kernel<N, AMAX, 256, true>), dim3(b->E), dim3(256), tab, s, t->u, t->ve, t->U, b->rb, \
                                b->uidx, b->nrules, b->maxR, p, states, action_ve, A, actconc, best);                                     \
        else hipLaunchKernelGGL((frirl::get_best_action_kernel<N, AMAX, 256, false>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, \
                                b->uidx, b->nrules, b->maxR, p, states, action_ve, A, actconc, best);                                     \
    } while (0)
    if (A <= 4) L(4);
    else if (A <= 8) L(8);
    else if (A <= 24 && !big && !frirl_host::opts().no_many) {      // 9..24 actions: every action in registers (sweep_gba_many; 256 threads)
        if (idx) hipLaunchKernelGGL((frirl::get_best_action_kernel<N, 24, 256, true>), dim3(b->E), dim3(256), tab, s, t->u, t->ve, t->U, b->rb,
                                    b->uidx, b->nrules, b->maxR, p, states, action_ve, A, actconc, best);
        else hipLaunchKernelGGL((frirl::get_best_action_kernel<N, 24, 256, false>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb,
                                b->uidx, b->nrules, b->maxR, p, states, action_ve, A, actconc, best);
    } else L(32);          // more actions, or few environments: action-parallel waves (sweep_gba_wide)
#undef L
}

extern "C" int frirl_hip_get_best_action(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *states,
                                         const double *action_ve, int A, double *actconc, int32_t *best, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!states || !action_ve || !actconc || !best) { set_error("frirl_hip_get_best_action: NULL argument"); return FRIRL_HIP_EINVAL; }
    if (A < 1 || A > FRIRL_HIP_MAX_ACTIONS) { set_error("frirl_hip_get_best_action: A=%d outside 1..%d", A, FRIRL_HIP_MAX_ACTIONS); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);
    switch (t->nant) {
#define M(N) case N: launch_gba<N>(t, b, eff_p(t, p), states, action_ve, A, actconc, best, s); break;
        FRIRL_GBA_NANT_CASES(M)
#undef M
        default: set_error("frirl_hip_get_best_action: nant=%d outside 2..9", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("frirl_hip_get_best_action");
}
