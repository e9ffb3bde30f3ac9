// bestact.hip -- conclusion from precomputed rule distances, and slab compaction.
#include "sweeps.h"

namespace frirl {

// FIVEVagConcl_FRIRL_BestAct (reference src/five/FIVEVagConcl_FRIRL_BestAct.c:56-299): first
// exact hit (:89-93) else Shepard (:212-217,265).  One workgroup per rule base.
__global__ __launch_bounds__(256) void bestact_kernel(const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR, int nant,
                                                       int p, const double *__restrict__ dists, double *__restrict__ conc)
{
    const int e = blockIdx.x;
    const int R = nrules[e];
    __shared__ BlockRed<256> red;
    const double *qcol = rb + ((size_t)e * (nant + 1) + nant) * maxR;
    const double *d = dists + (size_t)e * maxR;
    unsigned best = FRIRL_HIP_NO_HIT;
    double sv = 0.0, sw = 0.0;
    for (int r = threadIdx.x; r < R; r += 256) {
        const double dr = fabs(d[r]);
        if (dr == 0.0) best = min(best, (unsigned)r);
        else {
            const double wi = 1.0 / pow_int(dr, p);
            const double t = wi * qcol[r];
            sv = sv + t;
            sw = sw + wi;
        }
    }
    const unsigned hit = blk_min<256>(best, red);
    const double tv = blk_sum<256>(sv, red), tw = blk_sum<256>(sw, red);
    if (threadIdx.x == 0) conc[e] = (hit != FRIRL_HIP_NO_HIT) ? qcol[hit] : tv / tw;
}

// five_remove_rule (reference src/five/five_remove_rule.c:29-85): shift every column left by one
// from rule r on; one workgroup per column, chunked so that reads complete before the overlapping writes.
__global__ __launch_bounds__(256) void remove_rule_kernel(double *__restrict__ rb, int maxR, int cols, int R, int r)
{
    double *col = rb + (size_t)blockIdx.x * maxR;
    for (int base = r; base < R - 1; base += 256) {
        const int i = base + threadIdx.x;
        double v = 0.0;
        const bool in = i < R - 1;
        if (in) v = col[i + 1];
        __syncthreads();
        if (in) col[i] = v;
        __syncthreads();
    }
    if (threadIdx.x == 0) col[R - 1] = 0.0;
}

}  // namespace frirl

extern "C" int five_hip_bestact(const frirl_hip_rulebases *b, int nant, int p, const double *ruledists, double *conc, void *stream)
{
    using namespace frirl_host;
    if (!b || !b->rb || !b->nrules || !ruledists || !conc || nant < 1 || nant > FRIRL_HIP_MAX_NANT) { set_error("five_hip_bestact: bad arguments"); return FRIRL_HIP_EINVAL; }
    int rc = check_device();
    if (rc) return rc;
    hipLaunchKernelGGL(frirl::bestact_kernel, dim3(b->E), dim3(256), 0, as_stream(stream), b->rb, b->nrules, b->maxR, nant, p > 0 ? p : nant,
                       ruledists, conc);
    return check_launch("five_hip_bestact");
}
