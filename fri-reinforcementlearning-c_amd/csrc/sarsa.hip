// sarsa.hip -- SARSA rule update, rule append, environment step and the fused episode step.
//
// One workgroup owns one environment (its private rule base slab, episode state and sticky flags):
// every decision of the reference's frirl_update_sarsa() is workgroup-uniform, so the whole TD step
// -- Q(s',a'), Q(s,a), threshold test, grid snap, lookup of the snapped point, append or write-back --
// runs inside one launch without inter-workgroup communication or atomics.
#include <string.h>

#include "envs.h"
#include "sweeps.h"

namespace frirl {

struct StepShared {
    double q_ant[FRIRL_HIP_MAX_NANT];      // raw antecedent values of (s, a)
    double cur_q_ant[FRIRL_HIP_MAX_NANT];  // raw antecedent values of (s', a')
    double ve1[FRIRL_HIP_MAX_NANT];        // VE values of q_ant
    double ve2[FRIRL_HIP_MAX_NANT];        // VE values of cur_q_ant
    double rant[FRIRL_HIP_MAX_NANT];       // grid-snapped antecedents of a would-be new rule
    double ve3[FRIRL_HIP_MAX_NANT];        // their VE values
    double cur_states[FRIRL_HIP_MAX_NANT];
    unsigned idx3[FRIRL_HIP_MAX_NANT];    // universe indices of the snapped antecedents
    double reward;
    int success;
    int same;
};

// test hook: frirl_hip_agent.debug_flags bit 0 forces update_rules' second sweep (the candidate path must give the same bits)
__device__ __forceinline__ bool frirl_no_spread_candidates(const frirl_hip_agent &ag) { return (ag.debug_flags & 1) != 0; }

// frirl_update_sarsa + update_rules (reference src/frirl/frirl_update_sarsa.c:348-385, :22-143).
// `qp_known`: Q(s',a') already available (fused step: the greedy sweep produced it, identical
// operands and order -- SURVEY 7(i)); otherwise it is computed by a sweep over ve2.
template <int NANT, int BLOCK, bool TRACK = false, class COLS, class POW>
__device__ int update_sarsa_block(const COLS &cols, const double *__restrict__ u, const double *__restrict__ ve, int U, double *__restrict__ base,
                                  int maxR, int32_t *nrules_e, const frirl_hip_agent &ag, StepShared &sh, double reward,
                                  bool qp_known, double qp, int32_t *fus_e, double *rant_e, BlockRed<BLOCK> &red,
                                  const QResult *rn_known, uint16_t *uidx_e, POW p, SpreadCand *slot, double *spread_ant_e = nullptr, int32_t *spread_R_e = nullptr)
{
    const int R = *nrules_e;
    double q1[NANT], q2[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) { q1[k] = sh.ve1[k]; q2[k] = sh.ve2[k]; }
    double *qcol = base + (size_t)NANT * maxR;

    if (!qp_known) {                                                        // :356  Q(s',a')
        const QResult rp = sweep_q<NANT, BLOCK>(cols, qcol, R, q2, p, red);
        qp = (rp.hit != FRIRL_HIP_NO_HIT) ? qcol[rp.hit] : rp.vagc / rp.ws;
    }
    const QResult rn = rn_known ? *rn_known : sweep_q<NANT, BLOCK, TRACK>(cols, qcol, R, q1, p, red, ag.weight_significant, slot);    // :357  Q(s,a)
    const double qnow = (rn.hit != FRIRL_HIP_NO_HIT) ? qcol[rn.hit] : rn.vagc / rn.ws;
    const double qdiff = ag.alpha * (reward + ag.gamma * qp - qnow);        // :358
    int fus = *fus_e;
    __syncthreads();   // every thread has read *fus_e / *nrules_e before thread 0 may rewrite them

    if (qdiff > ag.qdiff_pos_boundary || qdiff < ag.qdiff_neg_boundary) {   // :363
        // snap the antecedents to the allowed grid (check_possible_states, :146-170)
        if (threadIdx.x < NANT) {
            const int k = threadIdx.x;
            const double r = check_possible_states(sh.q_ant[k], ag.grid_values + (size_t)k * FRIRL_HIP_MAX_GRID, ag.grid_len[k]);
            sh.rant[k] = r;
            const double *uni = u + (size_t)k * U;
            const unsigned j = snap_index(uni, U, r, universe_div(uni, U));
            sh.idx3[k] = j;
            sh.ve3[k] = ve[(size_t)k * U + j];
        }
        __syncthreads();
        double q3[NANT];
        bool same = true;
#pragma unroll
        for (int k = 0; k < NANT; k++) { q3[k] = sh.ve3[k]; same = same && (q3[k] == q1[k]); }
        QResult rr = rn;                                                    // :370 (same VE point => same sweep result)
        if (!same) rr = sweep_q<NANT, BLOCK>(cols, qcol, R, q3, p, red);
        if (rr.hit == FRIRL_HIP_NO_HIT) {                                   // :373-377 append and leave
            if (R >= maxR) return FRIRL_HIP_UPD_FULL;
            const double rconc = rr.vagc / rr.ws;
            if (threadIdx.x < NANT) {
                base[(size_t)threadIdx.x * maxR + R] = q3[threadIdx.x];      // five_add_rule.c:80-81
                if (uidx_e) uidx_e[(size_t)threadIdx.x * maxR + R] = (uint16_t)sh.idx3[threadIdx.x];   // five_add_rule.c:76
                if (rant_e) rant_e[(size_t)threadIdx.x * maxR + R] = sh.rant[threadIdx.x];
            }
            if (threadIdx.x == 0) {
                qcol[R] = rconc + qdiff;
                *nrules_e = R + 1;
                *fus_e = 1;
            }
            return FRIRL_HIP_UPD_INSERTED;
        }
        fus = 0;                                                            // :378
    }

    // update_rules (:22-143); FIVE_vag_concl_weight(q_ant) sees the distances of the sweep above
    int rules = R;
    if (fus) rules--;                                                       // :30-33
    int status;
    if (rn.hit != FRIRL_HIP_NO_HIT && (ag.skip_rules == 0 || (ag.skip_rules == 1 && (int)rn.hit < rules))) {
        if (threadIdx.x == 0) qcol[rn.hit] = qnow + qdiff;                  // :55
        status = FRIRL_HIP_UPD_EXACT;
    } else if (ag.skip_rules == 1 && rn.hit != FRIRL_HIP_NO_HIT && (int)rn.hit == rules) {
        status = FRIRL_HIP_UPD_SKIPPED;                                     // :61-63
    } else {
        if (ag.skip_rules == 0) fus = 0;                                    // :70-73
        const int r_skip = fus ? R - 1 : -1;                                // :76,124-126: the just-inserted rule keeps its Q
        if (rn.hit == FRIRL_HIP_NO_HIT) {      // FIVE_vag_concl_weight interpolated (:40): this call defines FIVERB.weights from now on
            if (spread_ant_e && threadIdx.x < NANT) spread_ant_e[threadIdx.x] = sh.q_ant[threadIdx.x];
            if (spread_R_e && threadIdx.x == 0) *spread_R_e = R;
        }
        // K6+K7: from the candidates tracked during the Q(s,a) sweep when possible (no second pass over the slab), else the sweep
        const bool from_cand = TRACK && rn.tracked && rn.hit == FRIRL_HIP_NO_HIT && !frirl_no_spread_candidates(ag) &&
                               spread_from_candidates<BLOCK>(slot[threadIdx.x], qcol, rn.ws, qnow, qdiff, ag.weight_significant, r_skip, red);
        if (!from_cand) sweep_update<NANT, BLOCK>(cols, qcol, R, q1, p, rn.ws, qnow, qdiff, ag.weight_significant, r_skip);
        status = FRIRL_HIP_UPD_SPREAD;
    }
    if (threadIdx.x == 0) *fus_e = fus;
    return status;
}

template <int NANT, int BLOCK, bool IDX>
__global__ __launch_bounds__(BLOCK) void update_sarsa_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                              double *__restrict__ rb, uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules, int maxR,
                                                              const frirl_hip_agent ag, const frirl_hip_envs ev,
                                                              const double *__restrict__ q_ant, const double *__restrict__ reward,
                                                              const double *__restrict__ cur_q_ant, const uint8_t *__restrict__ active)
{
    const int e = blockIdx.x;
    if (active && !active[e]) {
        if (threadIdx.x == 0 && ev.status) ev.status[e] = FRIRL_HIP_UPD_INACTIVE;
        return;
    }
    extern __shared__ double tab_s[];
    __shared__ StepShared sh;
    __shared__ BlockRed<BLOCK> red;
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NANT) {
        const int k = threadIdx.x;
        const double a = q_ant[(size_t)e * NANT + k], c = cur_q_ant[(size_t)e * NANT + k];
        sh.q_ant[k] = a;
        sh.cur_q_ant[k] = c;
        sh.ve1[k] = observe_ve(u, ve, U, k, a);
        sh.ve2[k] = observe_ve(u, ve, U, k, c);
    }
    __syncthreads();
    double *base = rb + (size_t)e * (NANT + 1) * maxR;
    double *rant_e = ev.rant ? ev.rant + (size_t)e * NANT * maxR : nullptr;
    uint16_t *uidx_e = uidx ? uidx + (size_t)e * NANT * maxR : nullptr;
    const auto cols = ColsSel<IDX>::make(base, uidx_e, tab_s, maxR, U);
    const int st = update_sarsa_block<NANT, BLOCK>(cols, u, ve, U, base, maxR, nrules + e, ag, sh, reward[e], false, 0.0, ev.fus + e, rant_e, red, nullptr, uidx_e,
                                                   ag.p > 0 ? ag.p : NANT, nullptr,
                                                   ev.spread_ant ? ev.spread_ant + (size_t)e * NANT : nullptr, ev.spread_R ? ev.spread_R + e : nullptr);
    if (threadIdx.x == 0 && ev.status) ev.status[e] = st;
}

// FIVE_add_rule (reference src/five/five_add_rule.c:47-95): one thread per environment.
__global__ void add_rule_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U, int nant, double *__restrict__ rb,
                                uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules, int maxR, int E, const double *__restrict__ rant,
                                const double *__restrict__ rconc, const uint8_t *__restrict__ active, double *__restrict__ rant_store,
                                int32_t *__restrict__ added)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int ok = 0;
    if (!active || active[e]) {
        const int R = nrules[e];
        if (R < maxR) {
            double *base = rb + (size_t)e * (nant + 1) * maxR;
            for (int k = 0; k < nant; k++) {
                const double v = rant[(size_t)e * nant + k];
                const double *uni = u + (size_t)k * U;
                const unsigned j = snap_index(uni, U, v, universe_div(uni, U));
                base[(size_t)k * maxR + R] = ve[(size_t)k * U + j];
                if (uidx) uidx[((size_t)e * nant + k) * maxR + R] = (uint16_t)j;
                if (rant_store) rant_store[((size_t)e * nant + k) * maxR + R] = v;
            }
            base[(size_t)nant * maxR + R] = rconc[e];
            nrules[e] = R + 1;
            ok = 1;
        }
    }
    if (added) added[e] = ok;
}

// do_action + get_reward + quantize_observations: one thread per environment.
__global__ void env_step_kernel(const frirl_hip_agent ag, int E, int ns, const double *__restrict__ action,
                                const double *__restrict__ states, double *__restrict__ new_states, double *__restrict__ reward,
                                int32_t *__restrict__ success, double *__restrict__ q_states)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double s[FRIRL_HIP_MAX_NANT], n[FRIRL_HIP_MAX_NANT], q[FRIRL_HIP_MAX_NANT];
    for (int i = 0; i < ns; i++) s[i] = states[(size_t)e * ns + i];
    env_do_action(ag.env_kind, action[e], s, n);
    double r; int f;
    env_get_reward(ag.env_kind, n, r, f);
    env_quantize(ag.env_kind, ns, ag.grid_values, ag.grid_len, ag.grid_div, n, q);
    for (int i = 0; i < ns; i++) { new_states[(size_t)e * ns + i] = n[i]; q_states[(size_t)e * ns + i] = q[i]; }
    reward[e] = r;
    success[e] = f;
}

// frirl_episode(): start of an episode (reference src/frirl/frirl_episode.c:46-82).
template <int NANT, int AMAX, int BLOCK, bool IDX, bool PN>
__global__ __launch_bounds__(BLOCK) void episode_begin_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                               const double *__restrict__ rb, const uint16_t *__restrict__ uidx,
                                                               const int32_t *__restrict__ nrules,
                                                               int maxR, const frirl_hip_agent ag, const frirl_hip_envs ev)
{
    constexpr int NS = NANT - 1;
    const int e = blockIdx.x;
    extern __shared__ double tab_s[];
    __shared__ double q_s[NS];
    __shared__ GbaScratch<AMAX, BLOCK> gs;
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x < NS) {
        const double v = ev.start_states ? ev.start_states[(size_t)e * NS + threadIdx.x] : ag.values_def[threadIdx.x];   // q_states = states = values_def (:46-48)
        ev.states[(size_t)e * NS + threadIdx.x] = v;
        ev.q_ant[(size_t)e * NANT + threadIdx.x] = v;
        q_s[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, v);
    }
    if ((int)threadIdx.x < ag.A) gs.ave[threadIdx.x] = ag.action_ve[threadIdx.x];
    __syncthreads();
    double q[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) q[k] = q_s[k];
    const double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const double *qcol = base + (size_t)NANT * maxR;
    const auto cols = ColsSel<IDX>::make(base, uidx + (IDX ? (size_t)e * NANT * maxR : 0), tab_s, maxR, U);
    const auto pw = PowSel<PN, NANT>::make(ag.p > 0 ? ag.p : NANT);
    int a0;
    if constexpr (AMAX == 24) {
        __shared__ BlockRed<BLOCK> red;
        double dummy[NANT] = {};
        a0 = sweep_gba_many<NANT, AMAX, BLOCK, false>(cols, qcol, nrules[e], q, dummy, pw, ag.A, gs, red, nullptr);   // :78
    } else if constexpr (AMAX > 8) {
        __shared__ BlockRed<BLOCK> red;
        double dummy[NANT] = {};
        a0 = sweep_gba_wide<NANT, 8, AMAX, BLOCK, false>(cols, qcol, nrules[e], q, dummy, pw, ag.A, gs, red, nullptr);   // :78
    } else {
        a0 = sweep_gba<NANT, AMAX, BLOCK>(cols, qcol, nrules[e], q, pw, ag.A, gs);   // :78
    }
    if (threadIdx.x == 0) {
        const uint32_t epi = ev.episode ? (uint32_t)(ev.episode[e] + 1) : 0u;
        if (ev.episode) ev.episode[e] = (int32_t)epi;
        a0 = e_greedy(ag, a0, (uint32_t)e, epi, 0u);
        ev.q_ant[(size_t)e * NANT + NS] = ag.grid_values[(size_t)NS * FRIRL_HIP_MAX_GRID + a0];                 // :82
        ev.done[e] = 0;
        ev.ep_steps[e] = 0;
        ev.ep_reward[e] = 0.0;
        if (ev.status) ev.status[e] = FRIRL_HIP_UPD_INACTIVE;
    }
}

// frirl_episode(): one step of the loop (reference src/frirl/frirl_episode.c:86-185), fused.
// register budget of the step kernel: 6 waves per SIMD (<= 80 VGPRs) for the 3-antecedent / <= 4-action shape (it needs ~64 now that
// the observations sit in SGPRs and exact hits need no registers; 8 waves -- all 8192 one-wave environments of the 8192 x 8192 shape
// resident at once -- measured the same 0.24 ms: the kernel is issue-bound, profiles/r02_step_timeline.txt), 4 (<= 128) for the
// others: without the bound the 5-antecedent, many-action variants sit just above 128 and lose a wave
// (the action-parallel kernel, amax > 8: 3 waves = 168 VGPRs -- its branch-free conclusion terms keep more chains in flight, and with
// cartpole's 40 KB of LDS tables only three workgroups fit a CU anyway)
#ifndef FRIRL_STEP_WAVES_N3
#define FRIRL_STEP_WAVES_N3 6
#endif
#ifndef FRIRL_STEP_SAME_CELL
#define FRIRL_STEP_SAME_CELL 1     // 0: always the full pending distance (A/B builds)
#endif
constexpr int step_min_waves(int nant, int amax) { return (nant <= 3 && amax <= 4) ? FRIRL_STEP_WAVES_N3 : (amax > 8 ? 3 : 4); }

// TRACK: the candidates of update_rules' write-back are collected during the fused sweep (sweeps.h: SpreadCand) -- for LARGE rule
// bases, where the second sweep it saves is a second pass over HBM; small slabs are re-read from L2 and the plain form is faster.
#ifdef FRIRL_STEP_TIMING
// experiment build only (tools/build_variant.sh timing -DFRIRL_STEP_TIMING): when each workgroup of the step kernel started, finished its
// fused sweep and left, in 10 ns ticks of the device wall clock -- read back with frirl_hip_debug_step_timing
__device__ long long g_step_timing[4 * 65536];
#endif

template <int NANT, int AMAX, int BLOCK, bool IDX, bool PN, bool TRACK>
__global__ __launch_bounds__(BLOCK, step_min_waves(NANT, AMAX)) void episode_step_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                              double *__restrict__ rb, uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules,
                                                              int maxR, const frirl_hip_agent ag, const frirl_hip_envs ev)
{
    constexpr int NS = NANT - 1;
    const int e = blockIdx.x;
#ifdef FRIRL_STEP_TIMING
    const long long tm0 = wall_clock64();
#endif
    if (ev.done[e]) {
        if (threadIdx.x == 0 && ev.status) ev.status[e] = FRIRL_HIP_UPD_INACTIVE;
        return;
    }
    extern __shared__ double tab_s[];
    __shared__ StepShared sh;
    __shared__ BlockRed<BLOCK> red;
    __shared__ GbaScratch<AMAX, BLOCK> gs;
    __shared__ SpreadCand cand_s[TRACK ? BLOCK : 1];       // one slot per lane: candidates of update_rules' write-back (sweeps.h)
    if (IDX) for (int i = threadIdx.x; i < NANT * U; i += BLOCK) tab_s[i] = ve[i];
    if (threadIdx.x == 0) {
        double s[FRIRL_HIP_MAX_NANT], q[FRIRL_HIP_MAX_NANT];
        for (int i = 0; i < NS; i++) s[i] = ev.states[(size_t)e * NS + i];
        for (int i = 0; i < NANT; i++) sh.q_ant[i] = ev.q_ant[(size_t)e * NANT + i];
        env_do_action(ag.env_kind, sh.q_ant[NS], s, sh.cur_states);                                   // :97
        env_get_reward(ag.env_kind, sh.cur_states, sh.reward, sh.success);                            // :106
        env_quantize(ag.env_kind, NS, ag.grid_values, ag.grid_len, ag.grid_div, sh.cur_states, q);   // :112
        for (int i = 0; i < NS; i++) sh.cur_q_ant[i] = q[i];
    }
    if ((int)threadIdx.x < ag.A) gs.ave[threadIdx.x] = ag.action_ve[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < NANT) sh.ve1[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, sh.q_ant[threadIdx.x]);
    if (threadIdx.x < NS) sh.ve2[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, sh.cur_q_ant[threadIdx.x]);
    __syncthreads();
    double q[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) q[k] = sh.ve2[k];
    double *base = rb + (size_t)e * (NANT + 1) * maxR;
    const double *qcol = base + (size_t)NANT * maxR;
    uint16_t *uidx_e = uidx ? uidx + (size_t)e * NANT * maxR : nullptr;
    const auto cols = ColsSel<IDX>::make(base, uidx_e, tab_s, maxR, U);
    double q1[NANT];
#pragma unroll
    for (int k = 0; k < NANT; k++) q1[k] = sh.ve1[k];
    const auto pw = PowSel<PN, NANT>::make(ag.p > 0 ? ag.p : NANT);
    QResult rn;
    // one pass over the slab: greedy action for s' (:148) AND Q(s,a) of the pending update (frirl_update_sarsa.c:357)
    int ap;
    if constexpr (AMAX == 24) ap = sweep_gba_many<NANT, AMAX, BLOCK, true>(cols, qcol, nrules[e], q, q1, pw, ag.A, gs, red, &rn);
    else if constexpr (AMAX > 8) ap = sweep_gba_wide<NANT, 8, AMAX, BLOCK, true, TRACK>(cols, qcol, nrules[e], q, q1, pw, ag.A, gs, red, &rn, ag.weight_significant, cand_s);
    else {
        // the agent has not left its quantisation cell (workgroup-uniform): Q(s, a) of the pending update is the greedy sweep's conclusion
        // for action a at the new observation (sweeps.h: SAMES)
        bool same_cell = NS > 0 && FRIRL_STEP_SAME_CELL != 0;
#pragma unroll
        for (int k = 0; k < NS; k++) same_cell = same_cell && (q[k] == q1[k]);
        int apend = -1;
        for (int a = ag.A - 1; a >= 0; a--) if (gs.ave[a] == q1[NS]) apend = a;
        if (same_cell && apend >= 0) ap = sweep_gba_q<NANT, AMAX, BLOCK, TRACK, true>(cols, qcol, nrules[e], q, q1, pw, ag.A, gs, red, rn, ag.weight_significant, cand_s, apend);
        else ap = sweep_gba_q<NANT, AMAX, BLOCK, TRACK, false>(cols, qcol, nrules[e], q, q1, pw, ag.A, gs, red, rn, ag.weight_significant, cand_s);
    }
#ifdef FRIRL_STEP_TIMING
    const long long tm1 = wall_clock64();
#endif
    if (threadIdx.x == 0) {
        const int chosen = e_greedy(ag, ap, (uint32_t)e, ev.episode ? (uint32_t)ev.episode[e] : 0u, (uint32_t)ev.ep_steps[e] + 1u);
        gs.best = chosen;
        sh.cur_q_ant[NS] = ag.grid_values[(size_t)NS * FRIRL_HIP_MAX_GRID + chosen];                  // :151
        sh.ve2[NS] = gs.ave[chosen];
    }
    __syncthreads();
    const double qp = gs.actconc[gs.best];     // Q(s',a') of the chosen action == FIVE_vag_concl(cur_q_ant), frirl_update_sarsa.c:356
    double *rant_e = ev.rant ? ev.rant + (size_t)e * NANT * maxR : nullptr;
    int st = FRIRL_HIP_UPD_INACTIVE;
    if (!ag.evaluate)                                                                                 // :155 (reduction_state == 0)
        st = update_sarsa_block<NANT, BLOCK, TRACK>(cols, u, ve, U, base, maxR, nrules + e, ag, sh, sh.reward, true, qp, ev.fus + e, rant_e, red, &rn, uidx_e, pw, cand_s,
                                                    ev.spread_ant ? ev.spread_ant + (size_t)e * NANT : nullptr, ev.spread_R ? ev.spread_R + e : nullptr);  // :159
    if (threadIdx.x < NS) ev.states[(size_t)e * NS + threadIdx.x] = sh.cur_states[threadIdx.x];      // :163-165
    if (threadIdx.x < NANT) ev.q_ant[(size_t)e * NANT + threadIdx.x] = sh.cur_q_ant[threadIdx.x];    // :166-168
    if (threadIdx.x == 0) {
        const int steps = ev.ep_steps[e] + 1;                                                         // :174
        ev.ep_steps[e] = steps;
        ev.ep_reward[e] = ev.ep_reward[e] + sh.reward;                                                // :107
        if (sh.success == 1 || steps >= ag.max_steps) ev.done[e] = 1;                                 // :183, :86
        if (ev.status) ev.status[e] = st;
#ifdef FRIRL_STEP_TIMING
        if (e < 65536) { g_step_timing[4 * e] = tm0; g_step_timing[4 * e + 1] = tm1; g_step_timing[4 * e + 2] = wall_clock64(); g_step_timing[4 * e + 3] = st; }
#endif
    }
}

#ifdef FRIRL_STEP_TIMING
}  // namespace frirl
extern "C" int frirl_hip_debug_step_timing(long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(frirl::g_step_timing), sizeof(long long) * 4 * (size_t)n) == hipSuccess ? 0 : -3;
}
namespace frirl {
#endif

// Persistent episode kernel for SMALL rule bases (the demos' real learning regime: <= 367 rules).
// One wave owns one environment for up to `nsteps` consecutive steps: the rule base slab, the universes / VE
// tables, the grids and the episode state are copied into LDS once and every step runs out of LDS -- no global
// round trip per step (the step kernel pays ~20 dependent global accesses, ~25 us per step).  Appends go to LDS and
// are written back with the rest of the slab when the wave leaves (episode end, nsteps exhausted, or the LDS slab
// full: then status = FRIRL_HIP_UPD_FULL with done == 0 and the caller continues with the step kernel).
// Arithmetic, lane mapping and reduction order are those of the one-wave step kernel => bit-identical results.
// PN: Shepard power = nant as a compile-time constant, exactly as in the step kernel (the two forms of the weight differ in their last bits).
template <int NANT, int AMAX, int CAP, bool PN>
__global__ __launch_bounds__(FRIRL_WAVE) void episode_run_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                  double *__restrict__ rb, uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules, int maxR,
                                                                  const frirl_hip_agent ag, const frirl_hip_envs ev, int nsteps)
{
    constexpr int NS = NANT - 1, BLOCK = FRIRL_WAVE;
    const int e = blockIdx.x;
    if (ev.done[e]) {
        if (threadIdx.x == 0 && ev.status) ev.status[e] = FRIRL_HIP_UPD_INACTIVE;
        return;
    }
    extern __shared__ double dyn_s[];                  // [2][NANT][U] universes, vague environments
    __shared__ double slab_s[(NANT + 1) * CAP];        // rule base: antecedent VE columns + consequents
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    __shared__ StepShared sh;
    __shared__ BlockRed<BLOCK> red;
    __shared__ GbaScratch<AMAX, BLOCK> gs;
    __shared__ double states_s[FRIRL_HIP_MAX_NANT];
    __shared__ int32_t nrules_s, fus_s, done_s, steps_s, status_s;
    __shared__ double reward_s;
    double *u_s = dyn_s, *ve_s = dyn_s + NANT * U;
    double *base_g = rb + (size_t)e * (NANT + 1) * maxR;
    const int R0 = nrules[e];
    if (R0 > CAP) {                                    // does not fit: leave everything to the step kernel
        if (threadIdx.x == 0 && ev.status) ev.status[e] = FRIRL_HIP_UPD_FULL;
        return;
    }
    for (int i = threadIdx.x; i < NANT * U; i += BLOCK) { u_s[i] = u[i]; ve_s[i] = ve[i]; }
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += BLOCK) grid_s[i] = ag.grid_values[i];
    for (int k = 0; k <= NANT; k++)
        for (int r = threadIdx.x; r < CAP; r += BLOCK) slab_s[k * CAP + r] = (r < R0) ? base_g[(size_t)k * maxR + r] : 0.0;
    if ((int)threadIdx.x < ag.A) gs.ave[threadIdx.x] = ag.action_ve[threadIdx.x];
    if (threadIdx.x < NS) states_s[threadIdx.x] = ev.states[(size_t)e * NS + threadIdx.x];
    if (threadIdx.x < NANT) sh.q_ant[threadIdx.x] = ev.q_ant[(size_t)e * NANT + threadIdx.x];
    if (threadIdx.x == 0) {
        nrules_s = R0; fus_s = ev.fus[e]; done_s = 0; steps_s = ev.ep_steps[e]; reward_s = ev.ep_reward[e]; status_s = FRIRL_HIP_UPD_INACTIVE;
    }
    frirl_hip_agent agl = ag;
    agl.grid_values = grid_s;
    const uint32_t episode = ev.episode ? (uint32_t)ev.episode[e] : 0u;
    double *rant_e = ev.rant ? ev.rant + (size_t)e * NANT * maxR : nullptr;
    uint16_t *uidx_e = uidx ? uidx + (size_t)e * NANT * maxR : nullptr;      // 16-bit index mirror: appends are written through (five_add_rule.c:76)
    const ColsLds cols{slab_s, CAP};
    double *qcol = slab_s + (size_t)NANT * CAP;
    const auto p = PowSel<PN, NANT>::make(ag.p > 0 ? ag.p : NANT);
    __syncthreads();

    for (int it = 0; it < nsteps; it++) {
        if (threadIdx.x == 0) {
            double q[FRIRL_HIP_MAX_NANT];
            env_do_action(ag.env_kind, sh.q_ant[NS], states_s, sh.cur_states);                        // frirl_episode.c:97
            env_get_reward(ag.env_kind, sh.cur_states, sh.reward, sh.success);                        // :106
            env_quantize(ag.env_kind, NS, grid_s, ag.grid_len, ag.grid_div, sh.cur_states, q);       // :112
            for (int i = 0; i < NS; i++) sh.cur_q_ant[i] = q[i];
        }
        __syncthreads();
        if (threadIdx.x < NANT) sh.ve1[threadIdx.x] = observe_ve(u_s, ve_s, U, threadIdx.x, sh.q_ant[threadIdx.x]);
        if (threadIdx.x < NS) sh.ve2[threadIdx.x] = observe_ve(u_s, ve_s, U, threadIdx.x, sh.cur_q_ant[threadIdx.x]);
        __syncthreads();
        double q[NS], q1[NANT];
#pragma unroll
        for (int k = 0; k < NS; k++) q[k] = sh.ve2[k];
#pragma unroll
        for (int k = 0; k < NANT; k++) q1[k] = sh.ve1[k];
        QResult rn;
        const int ap = sweep_gba_q<NANT, AMAX, BLOCK>(cols, qcol, nrules_s, q, q1, p, ag.A, gs, red, rn);   // :148 + frirl_update_sarsa.c:357
        if (threadIdx.x == 0) {
            const int chosen = e_greedy(ag, ap, (uint32_t)e, episode, (uint32_t)steps_s + 1u);
            gs.best = chosen;
            sh.cur_q_ant[NS] = grid_s[NS * FRIRL_HIP_MAX_GRID + chosen];                               // :151
            sh.ve2[NS] = gs.ave[chosen];
        }
        __syncthreads();
        const double qp = gs.actconc[gs.best];
        int st = FRIRL_HIP_UPD_INACTIVE;
        if (!ag.evaluate) st = update_sarsa_block<NANT, BLOCK>(cols, u_s, ve_s, U, slab_s, CAP, &nrules_s, agl, sh, sh.reward, true, qp, &fus_s, nullptr, red, &rn, nullptr, p, nullptr,
                                                                    ev.spread_ant ? ev.spread_ant + (size_t)e * NANT : nullptr, ev.spread_R ? ev.spread_R + e : nullptr);
        __syncthreads();
        if (st == FRIRL_HIP_UPD_INSERTED && threadIdx.x < NANT) {
            if (rant_e) rant_e[(size_t)threadIdx.x * maxR + (nrules_s - 1)] = sh.rant[threadIdx.x];
            if (uidx_e) uidx_e[(size_t)threadIdx.x * maxR + (nrules_s - 1)] = (uint16_t)sh.idx3[threadIdx.x];
        }
        if (st == FRIRL_HIP_UPD_FULL) {                // LDS slab full: hand the (unchanged) step back to the caller
            if (threadIdx.x == 0) status_s = FRIRL_HIP_UPD_FULL;
            __syncthreads();
            break;
        }
        if (threadIdx.x < NS) states_s[threadIdx.x] = sh.cur_states[threadIdx.x];                     // :163-165
        if (threadIdx.x < NANT) sh.q_ant[threadIdx.x] = sh.cur_q_ant[threadIdx.x];                    // :166-168
        if (threadIdx.x == 0) {
            steps_s = steps_s + 1;                                                                    // :174
            reward_s = reward_s + sh.reward;                                                          // :107
            status_s = st;
            if (sh.success == 1 || steps_s >= ag.max_steps) done_s = 1;                               // :183, :86
        }
        __syncthreads();
        if (done_s) break;
    }
    // write back: grown / updated slab, episode state
    const int R1 = nrules_s;
    for (int k = 0; k <= NANT; k++)
        for (int r = threadIdx.x; r < R1; r += BLOCK) base_g[(size_t)k * maxR + r] = slab_s[k * CAP + r];
    if (threadIdx.x < NS) ev.states[(size_t)e * NS + threadIdx.x] = states_s[threadIdx.x];
    if (threadIdx.x < NANT) ev.q_ant[(size_t)e * NANT + threadIdx.x] = sh.q_ant[threadIdx.x];
    if (threadIdx.x == 0) {
        nrules[e] = R1; ev.fus[e] = fus_s; ev.done[e] = done_s; ev.ep_steps[e] = steps_s; ev.ep_reward[e] = reward_s;
        if (ev.status) ev.status[e] = status_s;
    }
}


// ---- single rule base, one launch per environment step (ANSI-C drop-in path) -------------------------------------------
// Inputs by value in the kernel arguments, outputs written directly into pinned host memory (no staging copies).
// (A RESIDENT kernel serving the steps through a mailbox in host-mapped memory -- universes, VE tables, grids and the rule base held in
//  LDS, inputs polled over PCIe, results released with a flag -- was built and measured in round 2 (git history: commit "single-agent
//  drop-in step: ... experiment: resident step server"; profiles/r02_single_agent_step_server.txt): 3.6-3.9 us for the PCIe read of
//  the inputs + 5.4-11 us of LDS-resident work + 1.5 us for the results per step, i.e. no better than this kernel once the host polls
//  its completion flag -- the environment is the caller's C callback, so every step is a host round trip either way.  Dropped.)
struct MirrorStepArgs {
    double q_ant[FRIRL_HIP_MAX_NANT];
    double cur_q_states[FRIRL_HIP_MAX_NANT];
    double action_ve[FRIRL_HIP_MAX_ACTIONS];
    double action_values[FRIRL_HIP_MAX_ACTIONS];
    double reward;
    int32_t fus;
    int32_t A;
    uint32_t seq;               // step number, echoed in MirrorStepOut::seq when every result is visible to the host
};

struct MirrorStepOut {          // layout of the pinned result block
    double actconc[FRIRL_HIP_MAX_ACTIONS];
    double cur_q_ant[FRIRL_HIP_MAX_NANT];
    double new_rant[FRIRL_HIP_MAX_NANT];
    double new_rconc;
    int32_t best, fus, status, nrules;
    uint32_t seq;               // == MirrorStepArgs::seq once the step's results are complete (system-scope release): the host spins on it
    uint32_t pad;
};

template <int NANT, int AMAX, int BLOCK>
__global__ __launch_bounds__(BLOCK) void mirror_step_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                             double *__restrict__ rb, int32_t *__restrict__ nrules, int maxR,
                                                             const frirl_hip_agent ag, const MirrorStepArgs in, double *__restrict__ rant_store,
                                                             MirrorStepOut *__restrict__ out, double *__restrict__ rconc_out)
{
    constexpr int NS = NANT - 1;
    __shared__ StepShared sh;
    __shared__ BlockRed<BLOCK> red;
    __shared__ GbaScratch<AMAX, BLOCK> gs;
    __shared__ int32_t fus_s;
    if (threadIdx.x < NANT) { sh.q_ant[threadIdx.x] = in.q_ant[threadIdx.x]; sh.ve1[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, in.q_ant[threadIdx.x]); }
    if (threadIdx.x < NS) { sh.cur_q_ant[threadIdx.x] = in.cur_q_states[threadIdx.x]; sh.ve2[threadIdx.x] = observe_ve(u, ve, U, threadIdx.x, in.cur_q_states[threadIdx.x]); }
    if ((int)threadIdx.x < in.A) gs.ave[threadIdx.x] = in.action_ve[threadIdx.x];
    if (threadIdx.x == 0) fus_s = in.fus;
    __syncthreads();
    double q[NS], q1[NANT];
#pragma unroll
    for (int k = 0; k < NS; k++) q[k] = sh.ve2[k];
#pragma unroll
    for (int k = 0; k < NANT; k++) q1[k] = sh.ve1[k];
    const ColsF64 cols{rb, maxR};
    const double *qcol = rb + (size_t)NANT * maxR;
    const int p = ag.p > 0 ? ag.p : NANT;
    QResult rn;
    // (more than 8 actions: the action-parallel waves, not sweep_gba_many -- at a few hundred rules a step is ONE rule pair per lane and
    //  its time is the FP64 dependency latency of one lane's chains (~50 cycles per dependent instruction with one wave per SIMD):
    //  22 conclusions per lane measured 18 us per step, 6 per lane over four SIMDs 8 us)
    const int ap = (AMAX > 8) ? sweep_gba_wide<NANT, 8, AMAX, BLOCK, true>(cols, qcol, nrules[0], q, q1, p, in.A, gs, red, &rn)
                              : sweep_gba_q<NANT, AMAX, BLOCK>(cols, qcol, nrules[0], q, q1, p, in.A, gs, red, rn);
    if ((int)threadIdx.x < in.A) out->actconc[threadIdx.x] = gs.actconc[threadIdx.x];
    if (threadIdx.x == 0) {
        out->best = ap;
        sh.cur_q_ant[NS] = in.action_values[ap];
        sh.ve2[NS] = gs.ave[ap];
    }
    __syncthreads();
    if (threadIdx.x < NANT) out->cur_q_ant[threadIdx.x] = sh.cur_q_ant[threadIdx.x];
    const double qp = gs.actconc[ap];
    const int R_before = nrules[0];
    const int st = update_sarsa_block<NANT, BLOCK>(cols, u, ve, U, rb, maxR, nrules, ag, sh, in.reward, true, qp, &fus_s, rant_store, red, &rn, nullptr, p, nullptr);
    __syncthreads();
    const int R = (st == FRIRL_HIP_UPD_INSERTED) ? R_before + 1 : R_before;
    for (int r = threadIdx.x; r < R; r += BLOCK) rconc_out[r] = rb[(size_t)NANT * maxR + r];
    if (st == FRIRL_HIP_UPD_INSERTED && threadIdx.x < NANT) out->new_rant[threadIdx.x] = sh.rant[threadIdx.x];
    if (threadIdx.x == 0) {
        if (st == FRIRL_HIP_UPD_INSERTED) out->new_rconc = rb[(size_t)NANT * maxR + R_before];
        out->fus = fus_s; out->status = st; out->nrules = R;
    }
    // completion flag in the host-mapped result block: the host waits for it by polling its own memory instead of
    // hipStreamSynchronize (whose wake-up costs more than this whole kernel at a few hundred rules)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&out->seq, in.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// frirl_sequential_run's construct-loop bookkeeping (reference src/frirl/frirl_sequential_run.c:68-72,83-148),
// one workgroup per environment.
__global__ __launch_bounds__(256) void convergence_kernel(const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR,
                                                           int nant, const frirl_hip_agent ag, const frirl_hip_envs ev,
                                                           const frirl_hip_convergence c, int init)
{
    const int e = blockIdx.x;
    if (init != 1 && c.converged[e]) return;      // sticky: a completed rule base keeps its last report (workgroup-uniform)
    const double *qcol = rb + ((size_t)e * (nant + 1) + nant) * maxR;
    double *prev = c.prev_rconc + (size_t)e * maxR;
    const int R = nrules[e];
    __shared__ int moved_s;
    if (threadIdx.x == 0) moved_s = 0;
    __syncthreads();
    if (!init && !c.converged[e]) {
        const bool same = c.prev_nrules[e] == R && c.prev_steps[e] == ev.ep_steps[e] && ev.ep_reward[e] > ag.reward_good_above &&
                          c.prev_reward[e] == ev.ep_reward[e];                                            // :83-87
        int moved = 0;
        if (same)
            for (int r = threadIdx.x; r < R; r += blockDim.x)
                if (fabs(qcol[r] - prev[r]) >= ag.qdiff_final_tolerance) moved = 1;                    // :134-148
        if (moved) atomicOr(&moved_s, 1);
        __syncthreads();
        if (threadIdx.x == 0) {
            c.episodes[e] = c.episodes[e] + 1;
            if (same && c.epended) c.epended[e] = 1;                                                      // :88-90, before the tolerance check
            if (same && !moved_s) c.converged[e] = 1;
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < maxR; r += blockDim.x) prev[r] = qcol[r];                               // :72
    if (threadIdx.x == 0) {
        c.prev_nrules[e] = R;                                                                             // :68-70
        if (init == 1) { c.prev_steps[e] = -1; c.prev_reward[e] = -1.0; c.converged[e] = 0; c.episodes[e] = 0; }   // frirl_init.c:149-150
        else if (init == 2) { }                                     // refresh after a merge: steps / reward of the last episode stay (:66-68)
        else { c.prev_steps[e] = ev.ep_steps[e]; c.prev_reward[e] = ev.ep_reward[e]; }
    }
}

}  // namespace frirl

using namespace frirl_host;

static int check_agent(const frirl_hip_tables *t, const frirl_hip_agent *a, const char *who)
{
    if (!a || !a->grid_values) { set_error("%s: NULL agent / grid_values", who); return FRIRL_HIP_EINVAL; }
    for (int k = 0; k < t->nant; k++)
        if (a->grid_len[k] < 1 || a->grid_len[k] > FRIRL_HIP_MAX_GRID) { set_error("%s: grid_len[%d]=%d outside 1..%d", who, k, a->grid_len[k], FRIRL_HIP_MAX_GRID); return FRIRL_HIP_EINVAL; }
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_add_rule(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *rant, const double *rconc,
                                 const uint8_t *active, double *rant_store, int32_t *added, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (!rant || !rconc) { set_error("five_hip_add_rule: NULL rant/rconc"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipLaunchKernelGGL(frirl::add_rule_kernel, dim3((b->E + 255) / 256), dim3(256), 0, as_stream(stream), t->u, t->ve, t->U, t->nant, b->rb,
                       b->uidx, b->nrules, b->maxR, b->E, rant, rconc, active, rant_store, added);
    return check_launch("five_hip_add_rule");
}

#define FRIRL_Q_NANT_CASES(M) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)

extern "C" int frirl_hip_update_sarsa(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                      const frirl_hip_envs *envs, const double *q_ant, const double *reward, const double *cur_q_ant,
                                      const uint8_t *active, void *stream)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if ((rc = check_agent(t, agent, "frirl_hip_update_sarsa"))) return rc;
    if (!envs || !envs->fus || !q_ant || !reward || !cur_q_ant) { set_error("frirl_hip_update_sarsa: NULL argument"); return FRIRL_HIP_EINVAL; }
    if ((rc = check_device())) return rc;
    hipStream_t s = as_stream(stream);
    const bool big = b->E < 256;
    const bool idx = !big && frirl::use_uidx(t, b);
    const size_t tab = idx ? sizeof(double) * t->nant * (size_t)t->U : 0;
    switch (t->nant) {
#define M(N)                                                                                                                              \
    case N:                                                                                                                               \
        if (big) hipLaunchKernelGGL((frirl::update_sarsa_kernel<N, 1024, false>), dim3(b->E), dim3(1024), 0, s, t->u, t->ve, t->U, b->rb, b->uidx, b->nrules, \
                                    b->maxR, *agent, *envs, q_ant, reward, cur_q_ant, active);                                            \
        else if (idx) hipLaunchKernelGGL((frirl::update_sarsa_kernel<N, 256, true>), dim3(b->E), dim3(256), tab, s, t->u, t->ve, t->U, b->rb, b->uidx, b->nrules, \
                                b->maxR, *agent, *envs, q_ant, reward, cur_q_ant, active);                                                \
        else hipLaunchKernelGGL((frirl::update_sarsa_kernel<N, 256, false>), dim3(b->E), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->uidx, b->nrules,     \
                                b->maxR, *agent, *envs, q_ant, reward, cur_q_ant, active);                                                \
        break;
        FRIRL_Q_NANT_CASES(M)
#undef M
        default: set_error("frirl_hip_update_sarsa: nant=%d outside 2..9", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("frirl_hip_update_sarsa");
}

extern "C" int frirl_hip_env_step(const frirl_hip_agent *agent, int32_t E, int32_t nstates, const double *action, const double *states,
                                  double *new_states, double *reward, int32_t *success, double *q_states, void *stream)
{
    if (!agent || !agent->grid_values || !action || !states || !new_states || !reward || !success || !q_states) { set_error("frirl_hip_env_step: NULL argument"); return FRIRL_HIP_EINVAL; }
    if (E < 1 || nstates < 1 || nstates >= FRIRL_HIP_MAX_NANT) { set_error("frirl_hip_env_step: bad E/nstates"); return FRIRL_HIP_EINVAL; }
    const int need = agent->env_kind == FRIRL_HIP_ENV_MOUNTAINCAR ? 2 : 4;
    if (agent->env_kind < 0 || agent->env_kind > 2 || nstates != need) { set_error("frirl_hip_env_step: env_kind %d needs %d state dims", agent->env_kind, need); return FRIRL_HIP_EINVAL; }
    int rc = check_device();
    if (rc) return rc;
    hipLaunchKernelGGL(frirl::env_step_kernel, dim3((E + 255) / 256), dim3(256), 0, as_stream(stream), *agent, E, nstates, action, states,
                       new_states, reward, success, q_states);
    return check_launch("frirl_hip_env_step");
}

template <int N, int AMAX, int BLOCK, bool BEGIN>
static void launch_episode_v(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                             hipStream_t s)
{
    // compressed index mirror: large rule bases only (use_uidx); with one wave per environment only while the per-workgroup
    // LDS copy of the VE tables is small (<= 4 KiB: it does not limit the waves per CU)
    const bool idx = frirl::use_uidx(t, b) && (BLOCK >= 256 || sizeof(double) * t->nant * (size_t)t->U <= 4096);
    const size_t tab = idx ? sizeof(double) * t->nant * (size_t)t->U : 0;
    const bool pn = ag->p <= 0 || ag->p == N;                     // the Shepard power is the default nant: straight-line power (PowC<N>)
    // Spread candidates tracked in the fused sweep (sweeps.h: SpreadCand): where the second sweep would be a second pass over HBM
    // (large slabs) AND the sweep has registers to spare -- measured (tools/step_ab.py): acrobot 65 536 rules x 8 192 envs
    // 2.47 -> 2.26 ms per step; the 3-antecedent kernels (80-VGPR budget) and the 21-action kernel (already at 128) spill in the hot
    // loop with it (0.23 -> 0.57 ms, 3.0 -> 5.8 ms) and their second sweep is cheap beside A + 1 Shepard sums per rule, so they keep it.
    constexpr bool CAN_TRACK = (N >= 4 && AMAX <= 4);
    const int st_opt = frirl_host::opts().step_track;
    const bool track = CAN_TRACK && (st_opt == 1 || (st_opt < 0 && b->maxR > 16384 + 512));
#define EP_GO(KERNEL, DYN, ...)                                                                                                                  \
    hipLaunchKernelGGL((frirl::KERNEL<N, AMAX, BLOCK, __VA_ARGS__>), dim3(b->E), dim3(BLOCK), DYN, s, t->u, t->ve, t->U, b->rb, b->uidx, b->nrules, \
                       b->maxR, *ag, *ev)
    if constexpr (BEGIN) {
        if (idx) { if (pn) EP_GO(episode_begin_kernel, tab, true, true); else EP_GO(episode_begin_kernel, tab, true, false); }
        else { if (pn) EP_GO(episode_begin_kernel, 0, false, true); else EP_GO(episode_begin_kernel, 0, false, false); }
    } else {
        if constexpr (CAN_TRACK) {
            if (track) {
                if (idx) { if (pn) EP_GO(episode_step_kernel, tab, true, true, true); else EP_GO(episode_step_kernel, tab, true, false, true); }
                else { if (pn) EP_GO(episode_step_kernel, 0, false, true, true); else EP_GO(episode_step_kernel, 0, false, false, true); }
                return;
            }
        }
        if (idx) { if (pn) EP_GO(episode_step_kernel, tab, true, true, false); else EP_GO(episode_step_kernel, tab, true, false, false); }
        else { if (pn) EP_GO(episode_step_kernel, 0, false, true, false); else EP_GO(episode_step_kernel, 0, false, false, false); }
    }
#undef EP_GO
}

// Workgroup shape: 256 threads per environment for large rule bases (bandwidth); ONE wave per environment while
// the rule bases are small (<= 2048 rules: the demos' real learning regime, <= 367 rules) -- no cross-wave
// reductions or barriers on the critical path and 4x more environments resident per CU.  More than 8 actions
// always use 256 threads (the action-parallel sweep needs the waves).
template <int N, bool BEGIN>
static void launch_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                           hipStream_t s)
{
    // one wave per environment: small rule bases, or mid-size ones when the environments alone fill the chip (>= 4 waves
    // per SIMD): measured at 8192 rules x 8192 envs 0.320 -> 0.295 ms per step; at 65 536 rules the 256-thread form wins
    bool small = b->maxR <= 2048 || (b->maxR <= 16384 && b->E >= 4096);
    // (1024 threads per environment -- fewer environments in flight, fewer concurrent DRAM streams -- measured slower at 65 536 rules:
    //  2.27 -> 2.79 ms per step, tools/step_ab.py)
    const int sw = frirl_host::opts().step_wave;
    if (sw >= 0) small = sw == 1;
    // (two waves per environment -- 16 384 half-size waves instead of 8192, a finer last round -- measured 0.225 vs 0.217 ms at 8192 x 8192)
    if (ag->A <= 4) { if (small) launch_episode_v<N, 4, 64, BEGIN>(t, b, ag, ev, s); else launch_episode_v<N, 4, 256, BEGIN>(t, b, ag, ev, s); }
    else if (ag->A <= 8) { if (small) launch_episode_v<N, 8, 64, BEGIN>(t, b, ag, ev, s); else launch_episode_v<N, 8, 256, BEGIN>(t, b, ag, ev, s); }
    else if (ag->A <= 24 && !frirl_host::opts().no_many) launch_episode_v<N, 24, 256, BEGIN>(t, b, ag, ev, s);   // 9..24 actions: all in registers (sweep_gba_many)
    else launch_episode_v<N, 32, 256, BEGIN>(t, b, ag, ev, s);            // more: action-parallel waves (sweep_gba_wide)
}

// 1 when frirl_hip_episode_step streams the 16-bit index mirror for this shape (given that the caller provides one)
extern "C" int frirl_hip_step_uses_uidx(int32_t nant, int32_t U, int32_t maxR, int32_t E)
{
    frirl_hip_tables t = {nant, U, nullptr, nullptr};
    frirl_hip_rulebases b = {E, maxR, nullptr, nullptr, reinterpret_cast<uint16_t *>(16)};
    if (!frirl::use_uidx(&t, &b)) return 0;
    bool small = maxR <= 2048 || (maxR <= 16384 && E >= 4096);
    if (frirl_host::opts().step_wave >= 0) small = frirl_host::opts().step_wave == 1;
    return !small || sizeof(double) * nant * (size_t)U <= 4096;
}

static int check_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                         const frirl_hip_envs *envs, const char *who)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if ((rc = check_agent(t, agent, who))) return rc;
    if (!agent->action_ve || agent->A < 1 || agent->A > FRIRL_HIP_MAX_ACTIONS) { set_error("%s: bad action table", who); return FRIRL_HIP_EINVAL; }
    if (!envs || !envs->states || !envs->q_ant || !envs->fus || !envs->done || !envs->ep_steps || !envs->ep_reward) { set_error("%s: NULL env state", who); return FRIRL_HIP_EINVAL; }
    const int need = agent->env_kind == FRIRL_HIP_ENV_MOUNTAINCAR ? 3 : 5;
    if (agent->env_kind < 0 || agent->env_kind > 2 || t->nant != need) { set_error("%s: env_kind %d needs nant=%d", who, agent->env_kind, need); return FRIRL_HIP_EINVAL; }
    return check_device();
}

// shared with lanes.hip
int frirl_check_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs, const char *who)
{
    return check_episode(t, b, agent, envs, who);
}

extern "C" int frirl_hip_episode_begin(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                       const frirl_hip_envs *envs, void *stream)
{
    int rc = check_episode(t, b, agent, envs, "frirl_hip_episode_begin");
    if (rc) return rc;
    if (t->nant == 3) launch_episode<3, true>(t, b, agent, envs, as_stream(stream));
    else launch_episode<5, true>(t, b, agent, envs, as_stream(stream));
    return check_launch("frirl_hip_episode_begin");
}

extern "C" int frirl_hip_episode_step(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                      const frirl_hip_envs *envs, void *stream)
{
    int rc = check_episode(t, b, agent, envs, "frirl_hip_episode_step");
    if (rc) return rc;
    if (t->nant == 3) launch_episode<3, false>(t, b, agent, envs, as_stream(stream));
    else launch_episode<5, false>(t, b, agent, envs, as_stream(stream));
    return check_launch("frirl_hip_episode_step");
}

extern "C" int frirl_hip_episode_steps(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                       const frirl_hip_envs *envs, int32_t nsteps, void *stream)
{
    int rc = check_episode(t, b, agent, envs, "frirl_hip_episode_steps");
    if (rc) return rc;
    for (int i = 0; i < nsteps; i++) {
        if (t->nant == 3) launch_episode<3, false>(t, b, agent, envs, as_stream(stream));
        else launch_episode<5, false>(t, b, agent, envs, as_stream(stream));
    }
    return check_launch("frirl_hip_episode_steps");
}

static int check_convergence(const frirl_hip_rulebases *b, const frirl_hip_convergence *c, const char *who)
{
    if (!b || !b->rb || !b->nrules || b->E < 1) { set_error("%s: bad rule bases", who); return FRIRL_HIP_EINVAL; }
    if (!c || !c->prev_nrules || !c->prev_steps || !c->prev_reward || !c->prev_rconc || !c->converged || !c->episodes) { set_error("%s: NULL convergence state", who); return FRIRL_HIP_EINVAL; }
    return check_device();
}

extern "C" int frirl_hip_convergence_init(const frirl_hip_rulebases *b, int nant, const frirl_hip_convergence *c, void *stream)
{
    int rc = check_convergence(b, c, "frirl_hip_convergence_init");
    if (rc) return rc;
    frirl_hip_agent ag;
    frirl_hip_envs ev;
    memset(&ag, 0, sizeof ag);
    memset(&ev, 0, sizeof ev);
    hipLaunchKernelGGL(frirl::convergence_kernel, dim3(b->E), dim3(256), 0, as_stream(stream), b->rb, b->nrules, b->maxR, nant, ag, ev, *c, 1);
    return check_launch("frirl_hip_convergence_init");
}

// after the rule bases were changed from outside an episode (rule-base merge): the snapshot the next "same as the previous
// episode" test compares with is retaken (rule count, consequents; previous steps / reward invalidated); converged agents keep theirs
extern "C" int frirl_hip_convergence_refresh(const frirl_hip_rulebases *b, int nant, const frirl_hip_convergence *c, void *stream)
{
    int rc = check_convergence(b, c, "frirl_hip_convergence_refresh");
    if (rc) return rc;
    frirl_hip_agent ag;
    frirl_hip_envs ev;
    memset(&ag, 0, sizeof ag);
    memset(&ev, 0, sizeof ev);
    hipLaunchKernelGGL(frirl::convergence_kernel, dim3(b->E), dim3(256), 0, as_stream(stream), b->rb, b->nrules, b->maxR, nant, ag, ev, *c, 2);
    return check_launch("frirl_hip_convergence_refresh");
}

extern "C" int frirl_hip_convergence_update(const frirl_hip_rulebases *b, int nant, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                                            const frirl_hip_convergence *c, void *stream)
{
    int rc = check_convergence(b, c, "frirl_hip_convergence_update");
    if (rc) return rc;
    if (!agent || !envs || !envs->ep_steps || !envs->ep_reward) { set_error("frirl_hip_convergence_update: NULL agent/envs"); return FRIRL_HIP_EINVAL; }
    hipLaunchKernelGGL(frirl::convergence_kernel, dim3(b->E), dim3(256), 0, as_stream(stream), b->rb, b->nrules, b->maxR, nant, *agent, *envs, *c, 0);
    return check_launch("frirl_hip_convergence_update");
}

template <int N, int AMAX>
static void launch_run(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev, int nsteps,
                       int lds_rules, hipStream_t s)
{
    const size_t dyn = 2 * sizeof(double) * t->nant * (size_t)t->U;
    const bool pn = ag->p <= 0 || ag->p == N;
#define RUN1(CAP_, PN_) hipLaunchKernelGGL((frirl::episode_run_kernel<N, AMAX, CAP_, PN_>), dim3(b->E), dim3(FRIRL_WAVE), dyn, s, t->u, t->ve, t->U, b->rb, b->uidx, \
                                           b->nrules, b->maxR, *ag, *ev, nsteps)
#define RUN(CAP_) do { if (pn) RUN1(CAP_, true); else RUN1(CAP_, false); } while (0)
    if (lds_rules <= 256) RUN(256);
    else if (lds_rules <= 512) RUN(512);
    else RUN(1024);
#undef RUN
#undef RUN1
}

extern "C" int frirl_hip_episode_run(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                     const frirl_hip_envs *envs, int32_t nsteps, int32_t lds_rules, void *stream)
{
    int rc = check_episode(t, b, agent, envs, "frirl_hip_episode_run");
    if (rc) return rc;
    if (agent->A > 8 || lds_rules < 2 || lds_rules > 1024 || 2 * sizeof(double) * t->nant * (size_t)t->U > 16 * 1024) {
        set_error("frirl_hip_episode_run: needs A <= 8, lds_rules <= 1024 and universes/VE tables <= 16 KiB (got A=%d, lds_rules=%d, nant*U=%d)",
                  agent->A, lds_rules, t->nant * t->U);
        return FRIRL_HIP_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    if (t->nant == 3) { if (agent->A <= 4) launch_run<3, 4>(t, b, agent, envs, nsteps, lds_rules, s); else launch_run<3, 8>(t, b, agent, envs, nsteps, lds_rules, s); }
    else { if (agent->A <= 4) launch_run<5, 4>(t, b, agent, envs, nsteps, lds_rules, s); else launch_run<5, 8>(t, b, agent, envs, nsteps, lds_rules, s); }
    return check_launch("frirl_hip_episode_run");
}

// launcher used by mirror.hip (single rule base): see five_hip_mirror_greedy_step
int frirl_launch_mirror_step(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const double *q_ant, double reward,
                             const double *cur_q_states, const double *action_ve, const double *action_values, int A, int fus, double *rant_store,
                             void *out_dev, double *rconc_out_dev, uint32_t seq, hipStream_t s)
{
    frirl::MirrorStepArgs in;
    memset(&in, 0, sizeof in);
    memcpy(in.q_ant, q_ant, sizeof(double) * t->nant);
    memcpy(in.cur_q_states, cur_q_states, sizeof(double) * (t->nant - 1));
    memcpy(in.action_ve, action_ve, sizeof(double) * A);
    memcpy(in.action_values, action_values, sizeof(double) * A);
    in.reward = reward; in.fus = fus; in.A = A; in.seq = seq;
    frirl::MirrorStepOut *out = static_cast<frirl::MirrorStepOut *>(out_dev);
    switch (t->nant) {
#define M(N)                                                                                                                               \
    case N:                                                                                                                                \
        if (A <= 4) hipLaunchKernelGGL((frirl::mirror_step_kernel<N, 4, 256>), dim3(1), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, in, rant_store, out, rconc_out_dev); \
        else if (A <= 8) hipLaunchKernelGGL((frirl::mirror_step_kernel<N, 8, 256>), dim3(1), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, in, rant_store, out, rconc_out_dev); \
        else hipLaunchKernelGGL((frirl::mirror_step_kernel<N, 32, 256>), dim3(1), dim3(256), 0, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, in, rant_store, out, rconc_out_dev); \
        break;
        FRIRL_Q_NANT_CASES(M)
#undef M
        default: set_error("five_hip_mirror_greedy_step: nant=%d outside 2..9", t->nant); return FRIRL_HIP_EINVAL;
    }
    return check_launch("five_hip_mirror_greedy_step");
}

size_t frirl_mirror_step_out_bytes() { return sizeof(frirl::MirrorStepOut); }
bool frirl_mirror_step_done(const void *out_host, uint32_t seq)
{
    return __atomic_load_n(&static_cast<const frirl::MirrorStepOut *>(out_host)->seq, __ATOMIC_ACQUIRE) == seq;
}
void frirl_mirror_step_unpack(const void *out_host, int nant, int A, uint32_t *best, double *actconc, double *cur_q_ant, int32_t *fus, int32_t *status,
                              int32_t *nrules, double *new_rant, double *new_rconc)
{
    const frirl::MirrorStepOut *o = static_cast<const frirl::MirrorStepOut *>(out_host);
    *best = (uint32_t)o->best; *fus = o->fus; *status = o->status; *nrules = o->nrules;
    memcpy(actconc, o->actconc, sizeof(double) * A);
    memcpy(cur_q_ant, o->cur_q_ant, sizeof(double) * nant);
    if (new_rant) memcpy(new_rant, o->new_rant, sizeof(double) * nant);
    if (new_rconc) *new_rconc = o->new_rconc;
}
