// five_rule_distance.hip -- batched rule-distance scan + first exact hit.
//
// Replaces reference src/five/five_rule_distance.c:63-295 (AVX2 kernels K1 :82-100 and K2
// :160-236): the reference writes per-dimension squared differences to a 64 B/rule scratch and
// re-reads them; here the two passes are fused in registers, so the only HBM traffic is the
// compulsory one: 8*nant B read per rule (+ 8 B written when the distances are materialised).
//
// Mapping (HBM-bound stream, no reuse => no MFMA, no LDS tiling of the rule data, no XCD-aware remap: nothing but
// the small tables is shared between workgroups):
//   grid.x = environment, grid.y = rule chunk; 256 threads; every lane loads 16 B (two rules)
//   from each of the nant SoA columns => 1 KiB per wave-instruction, fully coalesced; UNROLL (8 for
//   nant <= 5) independent column sets are issued before the first use; loads and stores carry the
//   non-temporal hint (pure stream).  The observation's VE values are looked up once per workgroup
//   and staged in LDS.  First exact hit: per-lane minimum index -> wave butterfly -> LDS -> one
//   integer atomicMin per workgroup (deterministic; only taken when a hit exists).
// Two layouts, same arithmetic on the same doubles (bit-identical results):
//   rule_distance_kernel      streams the f64 VE columns (the reference's layout), 0.76 of HBM peak at cfg2;
//   rule_distance_idx_kernel  streams the 16-bit universe-index mirror and gathers the VE values from an LDS copy
//                             of the tables: 2*nant B read per rule, ~2x the evaluations per second.
// Measurements and the A/B log: profiles/r01_rule_distance_tuning.md.  The FRIRL_HIP_RD_* / FRIRL_HIP_NO_UIDX
// environment variables are experiment hooks (tools/ab_rd.py); unset, the shipped configuration runs.
#include <stdlib.h>

#include "device_common.h"

namespace frirl {

template <int NANT, bool WRITE, int UNROLL, int NT>
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
                for (int k = 0; k < NANT; k++) {
                    const double2 *src = reinterpret_cast<const double2 *>(base + (size_t)k * maxR + rr);
                    if (NT == 1 || NT == 2) { v[j][k].x = __builtin_nontemporal_load(&src->x); v[j][k].y = __builtin_nontemporal_load(&src->y); }
                    else v[j][k] = *src;
                }
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
                if (WRITE) {                                              // rows are even-sized: rr+1 < maxR
                    if (NT == 1 || NT == 3) { __builtin_nontemporal_store(d.x, out + rr); __builtin_nontemporal_store(d.y, out + rr + 1); }
                    else *reinterpret_cast<double2 *>(out + rr) = d;
                }
                // first exact hit among valid rules (five_rule_distance.c:215-217,241-262)
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = block_min_u32(best, red_s);
    if (threadIdx.x == 0 && best != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], best);
}


// Compressed-antecedent form: the rule base's 2-byte universe indices are streamed (4 B per lane and column:
// two rules) and the VE values come from an LDS copy of the vague-environment tables.  Same arithmetic on the
// same operands as above => bit-identical distances; HBM traffic 2*nant B read (+8 B written) per rule.
template <int NANT, bool WRITE, int UNROLL, int BLOCK = FRIRL_BLOCK>
__global__ __launch_bounds__(BLOCK) void rule_distance_idx_kernel(
    const double *__restrict__ u, const double *__restrict__ ve, int U, const uint16_t *__restrict__ uidx,
    const int32_t *__restrict__ nrules, int maxR, const double *__restrict__ x, double *__restrict__ dists,
    uint32_t *__restrict__ hit, int rules_per_block)
{
    extern __shared__ double tab_s[];            // [NANT][U] vague environments
    const int e = blockIdx.x;
    const int R = nrules[e];
    const int r0 = blockIdx.y * rules_per_block;
    if (r0 >= R) return;
    int r_end = r0 + rules_per_block;
    if (r_end > R) r_end = R;

    __shared__ double q_s[NANT];
    __shared__ unsigned red_s[BLOCK / FRIRL_WAVE];
    for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, x[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];

    const uint16_t *__restrict__ base = uidx + (size_t)e * NANT * maxR;
    double *__restrict__ out = WRITE ? dists + (size_t)e * maxR : nullptr;
    unsigned best = FRIRL_HIP_NO_HIT;
    constexpr int STEP = BLOCK * 2;

    for (int r = r0 + 2 * (int)threadIdx.x; r < r_end; r += STEP * UNROLL) {
        uint32_t w[UNROLL][NANT];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
#pragma unroll
                for (int k = 0; k < NANT; k++) w[j][k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(base + (size_t)k * maxR + rr));
            }
        }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) {
            const int rr = r + j * STEP;
            if (rr < r_end) {
                double d0 = q[0] - tab_s[w[j][0] & 0xFFFFu], d1 = q[0] - tab_s[w[j][0] >> 16];
                double a0 = d0 * d0, a1 = d1 * d1;
#pragma unroll
                for (int k = 1; k < NANT; k++) {
                    d0 = q[k] - tab_s[k * U + (w[j][k] & 0xFFFFu)];
                    d1 = q[k] - tab_s[k * U + (w[j][k] >> 16)];
                    const double s0 = d0 * d0, s1 = d1 * d1;
                    a0 = a0 + s0;
                    a1 = a1 + s1;
                }
                double2 d;
                d.x = __dsqrt_rn(a0);
                d.y = __dsqrt_rn(a1);
                if (WRITE) { __builtin_nontemporal_store(d.x, out + rr); __builtin_nontemporal_store(d.y, out + rr + 1); }
                if (d.y == 0.0 && rr + 1 < R) best = min(best, (unsigned)(rr + 1));
                if (d.x == 0.0) best = min(best, (unsigned)rr);
            }
        }
    }
    best = wave_min_u32(best);
    if ((threadIdx.x & (FRIRL_WAVE - 1)) == 0) red_s[threadIdx.x / FRIRL_WAVE] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red_s[0];
        for (int w = 1; w < BLOCK / FRIRL_WAVE; w++) m = red_s[w] < m ? red_s[w] : m;
        if (m != FRIRL_HIP_NO_HIT) atomicMin(&hit[e], m);
    }
}

struct RdTune { int unroll, chunk, nt; };
static RdTune rd_tune()
{
    const frirl_host::Options &o = frirl_host::opts();     // read once from the environment / frirl_hip_set_option, not per launch
    RdTune t;
    t.unroll = o.rd_unroll;          // 0 = shipped default
    t.chunk = o.rd_chunk;            // 0 = shipped default
    t.nt = o.rd_nt;                  // -1 = shipped default (non-temporal loads and stores)
    return t;
}

template <int NANT, int UNROLL, int NT>
static void launch_variant(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists, uint32_t *hit,
                           hipStream_t s, dim3 grid, int rules_per_block)
{
    if (ruledists)
        hipLaunchKernelGGL((rule_distance_kernel<NANT, true, UNROLL, NT>), grid, dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules,
                           b->maxR, x, ruledists, hit, rules_per_block);
    else
        hipLaunchKernelGGL((rule_distance_kernel<NANT, false, UNROLL, NT>), grid, dim3(FRIRL_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules,
                           b->maxR, x, ruledists, hit, rules_per_block);
}

// Shipped configuration (A/B-measured on MI355X, tools/ab_rd.py, profiles/r01_rule_distance_tuning.md):
// non-temporal loads AND stores (pure stream, nothing is re-read: +5..8 %), 8 independent column sets
// per lane for nant <= 5 (all loads issued before the first use), 4 for nant <= 8, 2 above.
template <int NANT>
struct RdConfig {
    static constexpr int UNROLL = (NANT <= 5) ? 8 : (NANT <= 8 ? 4 : 2);
};

template <int NANT>
static int launch_nant(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x, double *ruledists,
                       uint32_t *hit, hipStream_t s, dim3 grid, int rules_per_block)
{
    constexpr int UNROLL = RdConfig<NANT>::UNROLL;
    const size_t tab_bytes = sizeof(double) * NANT * (size_t)t->U;
    if (b->uidx && t->U <= 65536 && tab_bytes <= 48 * 1024 && !frirl_host::opts().no_uidx) {
        constexpr int UI = (NANT <= 8) ? 4 : 2;
        const int un = (NANT <= 5 && rd_tune().unroll) ? rd_tune().unroll : UI;      // tuning hook (experiments only)
#define VI(U_)                                                                                                                               \
    do {                                                                                                                                     \
        if (ruledists)                                                                                                                       \
            hipLaunchKernelGGL((rule_distance_idx_kernel<NANT, true, U_>), grid, dim3(FRIRL_BLOCK), tab_bytes, s, t->u, t->ve, t->U, b->uidx, \
                               b->nrules, b->maxR, x, ruledists, hit, rules_per_block);                                                      \
        else                                                                                                                                 \
            hipLaunchKernelGGL((rule_distance_idx_kernel<NANT, false, U_>), grid, dim3(FRIRL_BLOCK), tab_bytes, s, t->u, t->ve, t->U, b->uidx, \
                               b->nrules, b->maxR, x, ruledists, hit, rules_per_block);                                                      \
    } while (0)
        if (NANT <= 5 && un == 8) VI(8);
        else if (NANT <= 5 && un == 2) VI(2);
        else if (NANT <= 5 && un == 1) VI(1);
        else VI(UI);
#undef VI
        return frirl_host::check_launch("five_hip_rule_distance(uidx)");
    }
    // Large tables (cfg5: 16 x 1001 doubles = 125 KiB): still one LDS copy per workgroup -- gfx950 has 160 KiB of LDS per CU
    // -- with 1024 threads per workgroup (16 waves per CU on one table) and long rule chunks that amortise the table fill.
    if (b->uidx && t->U <= 65536 && tab_bytes <= 150 * 1024 && !frirl_host::opts().no_uidx) {
        constexpr int BIG = 1024;
        int rpb = 32768;                                       // rules per workgroup: 16 sweeps of 2048
        if (rpb > b->maxR) rpb = ((b->maxR + 2 * BIG - 1) / (2 * BIG)) * (2 * BIG);
        const dim3 g2(b->E, (b->maxR + rpb - 1) / rpb);
        hipError_t e1;
        if (ruledists) {
            auto k = rule_distance_idx_kernel<NANT, true, 2, BIG>;
            e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
            if (e1 == hipSuccess) hipLaunchKernelGGL(k, g2, dim3(BIG), tab_bytes, s, t->u, t->ve, t->U, b->uidx, b->nrules, b->maxR, x, ruledists, hit, rpb);
        } else {
            auto k = rule_distance_idx_kernel<NANT, false, 2, BIG>;
            e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
            if (e1 == hipSuccess) hipLaunchKernelGGL(k, g2, dim3(BIG), tab_bytes, s, t->u, t->ve, t->U, b->uidx, b->nrules, b->maxR, x, ruledists, hit, rpb);
        }
        if (e1 != hipSuccess) { frirl_host::set_error("five_hip_rule_distance: cannot reserve %zu B of LDS: %s", tab_bytes, hipGetErrorString(e1)); return FRIRL_HIP_ELAUNCH; }
        return frirl_host::check_launch("five_hip_rule_distance(uidx, large tables)");
    }
    const RdTune tn = rd_tune();
    if (NANT <= 5 && (tn.unroll || tn.nt >= 0)) {      // tuning hooks (experiments only)
        const int un = tn.unroll ? tn.unroll : UNROLL;
        const int nt = tn.nt >= 0 ? tn.nt : 1;
#define V(U_, N_) launch_variant<NANT, U_, N_>(t, b, x, ruledists, hit, s, grid, rules_per_block)
        if (nt == 1) { if (un == 1) V(1, 1); else if (un == 2) V(2, 1); else if (un == 8) V(8, 1); else V(4, 1); }
        else if (nt == 2) { if (un == 1) V(1, 2); else if (un == 2) V(2, 2); else if (un == 8) V(8, 2); else V(4, 2); }
        else if (nt == 3) { if (un == 1) V(1, 3); else if (un == 2) V(2, 3); else if (un == 8) V(8, 3); else V(4, 3); }
        else { if (un == 1) V(1, 0); else if (un == 2) V(2, 0); else if (un == 8) V(8, 0); else V(4, 0); }
#undef V
    } else {
        launch_variant<NANT, UNROLL, 1>(t, b, x, ruledists, hit, s, grid, rules_per_block);
    }
    return frirl_host::check_launch("five_hip_rule_distance");
}

}  // namespace frirl

extern "C" int five_hip_rule_distance_uses_uidx(int32_t nant, int32_t U)
{
    return nant >= 1 && nant <= FRIRL_HIP_MAX_NANT && U <= 65536 && sizeof(double) * nant * (size_t)U <= 150 * 1024 && !frirl_host::opts().no_uidx;
}

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

    // Rule chunk per workgroup (A/B-measured): 1024 rules for rule bases up to 16 K rules, 8192 above; always
    // a multiple of one 512-rule sweep.  grid = (environment, chunk) => >= 8 workgroups per environment at
    // the BASELINE shapes, tens of thousands of short workgroups in total.
    const int forced = frirl::rd_tune().chunk;
    int rules_per_block = forced > 0 ? forced : (b->maxR <= 16384 + 512 ? 1024 : 8192);
    if (b->uidx && forced <= 0) rules_per_block = 2048;      // compressed form: the LDS table fill is amortised over a longer chunk
    rules_per_block = ((rules_per_block + 2 * FRIRL_BLOCK - 1) / (2 * FRIRL_BLOCK)) * (2 * FRIRL_BLOCK);
    int chunks = (b->maxR + rules_per_block - 1) / rules_per_block;
    if (chunks > 65535) { rules_per_block = ((b->maxR / 65535 + 2 * FRIRL_BLOCK) / (2 * FRIRL_BLOCK)) * (2 * FRIRL_BLOCK); chunks = (b->maxR + rules_per_block - 1) / rules_per_block; }
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
