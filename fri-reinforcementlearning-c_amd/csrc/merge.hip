// merge.hip -- multi-agent rule-base merge.
//
// Replaces merge_rb of the reference's many-agent run modes (src/frirl/frirl_agent.c:58-117; update_rules variant :45-53,
// check_possible_states :18-43) and the start-state diversification gen_def_states (:121-139).  A receiver rule base takes
// over a list of sender rules ONE AFTER THE OTHER -- every sender rule sees the receiver as the previous ones left it, so the
// list is sequential by definition; what is parallel is the set of receivers: the first half of a frirl_omp_run round
// (:430-440) hands the master's rules to EVERY other agent.  One workgroup owns one receiver (as in sarsa.hip: every decision is
// workgroup-uniform); per sender rule one fused Q sweep (hit, Shepard sums), the weights sweep when it interpolated, the snapped
// point's sweep when the conclusions differ by more than the qdiff boundaries, then the append / blend / weighted overwrite.
// The receiver's weights array persists between sender rules exactly like FIVERB.weights (an exact hit leaves it untouched,
// FIVEVagConclWeight.c:67-69, and the later weighted overwrite then uses the weights of the last interpolated rule).
#include <string.h>

#include "envs.h"
#include "sweeps.h"

namespace frirl {

struct MergeShared {
    double q_ant[FRIRL_HIP_MAX_NANT];    // raw antecedents of the sender rule
    double ve1[FRIRL_HIP_MAX_NANT];      // their VE values
    double rant[FRIRL_HIP_MAX_NANT];     // grid-snapped antecedents
    double ve3[FRIRL_HIP_MAX_NANT];
    unsigned idx3[FRIRL_HIP_MAX_NANT];
};

template <int NANT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void merge_rb_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, double *__restrict__ rb,
                                                          uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules, int maxR, const frirl_hip_agent ag,
                                                          double *__restrict__ rant_store, const double *__restrict__ sndr_rant, long rule_stride, long dim_stride,
                                                          const double *__restrict__ sndr_rconc, int S, const int32_t *__restrict__ S_dev,
                                                          double *__restrict__ weights, const uint8_t *__restrict__ active, int32_t *__restrict__ full)
{
    const int e = blockIdx.x;
    if (active && !active[e]) return;
    __shared__ MergeShared sh;
    __shared__ BlockRed<BLOCK> red;
    __shared__ int32_t R_s, full_s;
    double *base = rb + (size_t)e * (NANT + 1) * maxR;
    double *qcol = base + (size_t)NANT * maxR;
    double *w_e = weights + (size_t)e * maxR;
    uint16_t *uidx_e = uidx ? uidx + (size_t)e * NANT * maxR : nullptr;
    double *rant_e = rant_store ? rant_store + (size_t)e * NANT * maxR : nullptr;
    const ColsF64 cols{base, maxR};
    const int p = ag.p > 0 ? ag.p : NANT;
    if (S_dev) S = *S_dev;
    if (threadIdx.x == 0) { R_s = nrules[e]; full_s = 0; }
    __syncthreads();
    for (int r = 0; r < S; r++) {
        const int R = R_s;
        if (threadIdx.x < NANT) {
            const int k = threadIdx.x;
            const double a = sndr_rant[(size_t)r * rule_stride + (size_t)k * dim_stride];
            sh.q_ant[k] = a;
            sh.ve1[k] = observe_ve(u, ve, U, k, a);
        }
        const double sndr_q = sndr_rconc[r];
        __syncthreads();
        double q1[NANT];
#pragma unroll
        for (int k = 0; k < NANT; k++) q1[k] = sh.ve1[k];
        const QResult rn = sweep_q<NANT, BLOCK>(cols, qcol, R, q1, p, red);                  // FIVE_vag_concl_weight + FIVE_vag_concl (:72,75): same distances
        if (rn.hit == FRIRL_HIP_NO_HIT) sweep_weights<NANT, BLOCK>(cols, R, q1, p, rn.ws, w_e);
        const double rcvr_q = (rn.hit != FRIRL_HIP_NO_HIT) ? qcol[rn.hit] : rn.vagc / rn.ws;
        const double qdiff = -rcvr_q + sndr_q;                                                // :82
        __syncthreads();       // weights of this rule are complete before anyone reads them; everyone has read qcol[hit]
        if (qdiff > ag.qdiff_pos_boundary || qdiff < ag.qdiff_neg_boundary) {                 // :91
            if (threadIdx.x < NANT) {                                                         // check_possible_states (:18-43), CHECK_STATES = 1
                const int k = threadIdx.x;
                const double g = check_possible_states(sh.q_ant[k], ag.grid_values + (size_t)k * FRIRL_HIP_MAX_GRID, ag.grid_len[k]);
                sh.rant[k] = g;
                const double *uni = u + (size_t)k * U;
                const unsigned j = snap_index(uni, U, g, universe_div(uni, U));
                sh.idx3[k] = j;
                sh.ve3[k] = ve[(size_t)k * U + j];
            }
            __syncthreads();
            double q3[NANT];
            bool same = true;
#pragma unroll
            for (int k = 0; k < NANT; k++) { q3[k] = sh.ve3[k]; same = same && (q3[k] == q1[k]); }
            QResult rr = rn;                                                                  // :99 (same VE point => same sweep result)
            if (!same) rr = sweep_q<NANT, BLOCK>(cols, qcol, R, q3, p, red);
            if (rr.hit == FRIRL_HIP_NO_HIT) {                                                 // :100-103 new rule, mean of the two conclusions
                const double rconc = rr.vagc / rr.ws;
                if (R >= maxR) { if (threadIdx.x == 0) full_s = 1; }
                else {
                    if (threadIdx.x < NANT) {
                        base[(size_t)threadIdx.x * maxR + R] = q3[threadIdx.x];
                        if (uidx_e) uidx_e[(size_t)threadIdx.x * maxR + R] = (uint16_t)sh.idx3[threadIdx.x];
                        if (rant_e) rant_e[(size_t)threadIdx.x * maxR + R] = sh.rant[threadIdx.x];
                    }
                    if (threadIdx.x == 0) {
                        const double a = 0.5 * rconc, b2 = 0.5 * sndr_q;
                        qcol[R] = a + b2;
                        R_s = R + 1;
                    }
                }
            } else if (threadIdx.x == 0) {                                                    // :105 existing rule: 10 % towards the sender
                const double rconc = qcol[rr.hit];
                const double a = 0.9 * rconc, b2 = 0.1 * sndr_q;
                qcol[rr.hit] = a + b2;
            }
            __syncthreads();
            continue;
        }
        // the agent file's update_rules (:45-53): every rule with weight > delta is overwritten with q * weight
        const double a = 0.9 * rcvr_q, b2 = 0.1 * sndr_q;
        const double q = a + b2;
        for (int w = threadIdx.x; w < R; w += BLOCK) {
            const double wt = w_e[w];
            if (wt > ag.weight_significant) qcol[w] = q * wt;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        nrules[e] = R_s;
        if (full) full[e] = full_s;
    }
}

// FIVERB.weights as the learning loop left it (frirl_hip.h: frirl_hip_weights_from_spread)
template <int NANT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void weights_from_spread_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, const double *__restrict__ rb,
                                                                     int maxR, int p, const double *__restrict__ spread_ant, int32_t *__restrict__ spread_R,
                                                                     double *__restrict__ weights)
{
    const int e = blockIdx.x;
    const int R = spread_R[e];
    if (R <= 0) return;          // workgroup-uniform
    __shared__ double q_s[NANT];
    __shared__ BlockRed<BLOCK> red;
    if (threadIdx.x < NANT) q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, spread_ant[(size_t)e * NANT + threadIdx.x]);
    __syncthreads();
    double q[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q[k] = q_s[k];
    const double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const ColsF64 cols{base, maxR};
    const QResult rn = sweep_q<NANT, BLOCK>(cols, base + (size_t)NANT * maxR, R, q, p, red);
    if (rn.hit == FRIRL_HIP_NO_HIT) sweep_weights<NANT, BLOCK>(cols, R, q, p, rn.ws, weights + (size_t)e * maxR);
    __syncthreads();
    if (threadIdx.x == 0) spread_R[e] = 0;
}

}  // namespace frirl

using namespace frirl_host;

extern "C" int frirl_hip_weights_from_spread(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const frirl_hip_envs *envs, double *weights, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!envs || !envs->spread_ant || !envs->spread_R || !weights) { set_error("frirl_hip_weights_from_spread: NULL argument (envs->spread_ant / spread_R / weights)"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    const int pp = p > 0 ? p : t->nant;
    switch (t->nant) {
#define M(N)                                                                                                                              \
    case N:                                                                                                                               \
        hipLaunchKernelGGL((frirl::weights_from_spread_kernel<N, 256>), dim3(b->E), dim3(256), 0, as_stream(stream), t->u, t->ve, t->U, b->rb, b->maxR, pp, \
                           envs->spread_ant, envs->spread_R, weights);                                                                    \
        break;
        M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#undef M
        default: set_error("frirl_hip_weights_from_spread: nant=%d outside 2..9", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("frirl_hip_weights_from_spread");
}

extern "C" int frirl_hip_merge_rb(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, double *rant_store,
                                  const frirl_hip_sender *sender, double *weights, const uint8_t *active, int32_t *full, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!agent || !agent->grid_values || !sender || !sender->rant || !sender->rconc || !weights) { set_error("frirl_hip_merge_rb: NULL argument"); return FRIRL_HIP_EINVAL; }
    if (sender->S < 0 || (sender->rule_stride == 0 && sender->dim_stride == 0)) { set_error("frirl_hip_merge_rb: bad sender description"); return FRIRL_HIP_EINVAL; }
    for (int k = 0; k < t->nant; k++)
        if (agent->grid_len[k] < 1 || agent->grid_len[k] > FRIRL_HIP_MAX_GRID) { set_error("frirl_hip_merge_rb: grid_len[%d]=%d outside 1..%d", k, agent->grid_len[k], FRIRL_HIP_MAX_GRID); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    if (sender->S == 0 && !sender->S_dev) return FRIRL_HIP_OK;
    hipStream_t s = as_stream(stream);
    switch (t->nant) {
#define M(N)                                                                                                                                   \
    case N:                                                                                                                                    \
        hipLaunchKernelGGL((frirl::merge_rb_kernel<N, 256>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->uidx, b->nrules, b->maxR, *agent, \
                           rant_store, sender->rant, (long)sender->rule_stride, (long)sender->dim_stride, sender->rconc, sender->S, sender->S_dev, weights, active, full); \
        break;
        M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#undef M
        default: set_error("frirl_hip_merge_rb: nant=%d outside 2..9", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("frirl_hip_merge_rb");
}

// gen_def_states (reference frirl_agent.c:121-139): start state of every agent of a world of `world` from the master's rule
// list (host arrays: the rule base at omp_init time is the small initial one).  Agent 0 and worlds < 3 keep values_def
// (the reference divides by world - 2).
extern "C" int frirl_hip_gen_def_states(const double *master_rant, int32_t R, int32_t nant, int32_t world, const double *values_def, double *start_states)
{
    if (!master_rant || !values_def || !start_states || R < 1 || nant < 2 || world < 1) { set_error("frirl_hip_gen_def_states: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int ns = nant - 1;
    for (int id = 0; id < world; id++) {
        double *out = start_states + (size_t)id * ns;
        if (id == 0 || world < 3) { memcpy(out, values_def, sizeof(double) * ns); continue; }
        const int gap = R / (world - 2);
        size_t rule = (size_t)(id - 1) * gap;
        if (rule >= (size_t)R) rule = (size_t)R - 1;      // the reference reads one rule PAST its list for the last agent (uninitialised memory): last rule here
        for (int i = 0; i < ns; i++) out[i] = master_rant[rule * nant + i];
    }
    return FRIRL_HIP_OK;
}
