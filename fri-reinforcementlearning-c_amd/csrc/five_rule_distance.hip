// five_rule_distance.hip -- batched rule-distance scan + first exact hit.
//
// Replaces reference src/five/five_rule_distance.c:63-295 (AVX2 kernels K1 :82-100 and K2
// :160-236): the reference writes per-dimension squared differences to a 64 B/rule scratch and
// re-reads them; here the two passes are fused in registers, so the only HBM traffic is the
// compulsory one: 8*nant B read per rule (+ 8 B written when the distances are materialised).
//
// Mapping (HBM-bound stream, no reuse => no MFMA, no LDS tiling of the rule data):
//   grid.x = environment, grid.y = rule chunk; 256 threads; every lane loads 16 B (two rules)
//   from each of the nant SoA columns => 1 KiB per wave-instruction, fully coalesced; UNROLL
//   independent column sets are issued before the first use so that >= 12 x 16 B loads per lane
//   are in flight.  The observation's VE values are looked up once per workgroup and staged in
//   LDS.  First exact hit: per-lane minimum index -> wave butterfly -> LDS -> one integer
//   atomicMin per workgroup (deterministic; only taken when a hit exists).
#include "device_common.h"

namespace frirl {

template <int NANT, bool WRITE, int UNROLL>
__global__ __launch_bounds__(FRIRL_BLOCK) void rule_distance_kernel(
    const double *__restrict__ u, const double *__restrict__ ve, int U, const double *__restrict__ rb,
    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
    uint32_t *__restrict__ hit, int rules_per_block)
{
    const int e = blockIdx.x;
    const int R = nrules[e];
    const int r0 = blockIdx.y * rules_per_block;
    if (r0 >= R) return;   // uniform for the workgroup
    int r_end = r0 + rules_per_block;
    if (r_end > R) r_end = R;

    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[FRIRL_WAVES_PER_BLOCK];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];

    const double *__restrict__ base = rb + (size_t)e * (NANT + 1) * maxR;
    double *__restrict__ out = WRITE ? dists + (size_t)e * maxR : nullptr;
    unsigned best = FRIRL_HIP_NO_HIT;
    constexpr int STEP = FRIRL_BLOCK * 2;

    for (int r = r0 + 2 * (int)threadIdx.x; r < r_end; r += STEP * UNROLL) {
        double2 v[UNROLL][NANT];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
#pragma unroll
                for (int k = 0; k < NANT; k++) v[j][k] = *reinterpret_cast<const double2 *>(base + (size_t)k * maxR + rr);
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
                // dimension-ordered sum of squares, separate mul / add (five_rule_distance.c:88-90,171-208)
                double d0 = q[0] - v[j][0].x, d1 = q[0] - v[j][0].y;
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q[k] - v[j][k].x;
                    d1 = q[k] - v[j][k].y;
                    const double s0 = d0 * d0, s1 = d1 * d1;
                    a0 = a0 + s0;
                    a1 = a1 + s1;
                }
                double2 d;
                d.x = __dsqrt_rn(a0);   // IEEE sqrt (vsqrtpd, five_rule_distance.c:211)
                d.y = __dsqrt_rn(a1);
                if (WRITE) *reinterpret_cast<double2 *>(out + rr) = d;   // rows are even-sized: rr+1 < maxR
                // first exact hit among valid rules (five_rule_distance.c:215-217,241-262)
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = block_min_u32(best, red_s);
    if (threadIdx.x == 0 && best != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], best);
}

template <int NANT>
static int launch_nant(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists,
                       uint32_t *hit, hipStream_t s, dim3 grid, int rules_per_block)
{
    constexpr int UNROLL = (NANT <= 4) ? 4 : (NANT <= 8 ? 2 : 1);
    if (ruledists)
        hipLaunchKernelGGL((rule_distance_kernel<NANT, true, UNROLL>), grid, dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb,
                           b->nrules, b->maxR, x, ruledists, hit, rules_per_block);
    else
        hipLaunchKernelGGL((rule_distance_kernel<NANT, false, UNROLL>), grid, dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb,
                           b->nrules, b->maxR, x, ruledists, hit, rules_per_block);
    return frirl_host::check_launch("five_hip_rule_distance");
}

}  // namespace frirl

extern "C" int five_hip_rule_distance(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x,
                                      double *ruledists, uint32_t *hit, void *stream)
{
    using namespace frirl_host;
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!x || !hit) { set_error("five_hip_rule_distance: NULL x/hit"); return FRIRL_HIP_EINVAL; }
    if (ruledists && (reinterpret_cast<uintptr_t>(ruledists) & 15)) { set_error("five_hip_rule_distance: ruledists must be 16-byte aligned"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);

    // Enough workgroups to fill 256 CUs x 8 (>= 4096) without making chunks shorter than one
    // fully unrolled sweep (2048 rules).
    const int min_chunk = 2048;
    int chunks = (4096 + b->E - 1) / b->E;
    const int max_chunks = (b->maxR + min_chunk - 1) / min_chunk;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    int rules_per_block = (b->maxR + chunks - 1) / chunks;
    rules_per_block = ((rules_per_block + 2 * FRIRL_BLOCK - 1) / (2 * FRIRL_BLOCK)) * (2 * FRIRL_BLOCK);
    chunks = (b->maxR + rules_per_block - 1) / rules_per_block;
    dim3 grid((unsigned)b->E, (unsigned)chunks);

    if (hipMemsetAsync(hit, 0xFF, sizeof(uint32_t) * (size_t)b->E, s) != hipSuccess) return check_launch("five_hip_rule_distance(memset)");

    switch (t->nant) {
#define FRIRL_CASE(N) case N: return frirl::launch_nant<N>(t, b, x, ruledists, hit, s, grid, rules_per_block);
        FRIRL_CASE(1) FRIRL_CASE(2) FRIRL_CASE(3) FRIRL_CASE(4) FRIRL_CASE(5) FRIRL_CASE(6) FRIRL_CASE(7) FRIRL_CASE(8)
        FRIRL_CASE(9) FRIRL_CASE(10) FRIRL_CASE(11) FRIRL_CASE(12) FRIRL_CASE(13) FRIRL_CASE(14) FRIRL_CASE(15) FRIRL_CASE(16)
#undef FRIRL_CASE
    }
    set_error("five_hip_rule_distance: unsupported nant=%d", t->nant);
    return FRIRL_HIP_EINVAL;
}
