// lanes.hip -- frirl_episode's loop for MANY agents with SMALL rule bases (the demos' real learning regime, <= a few
// hundred rules per agent): a group of G consecutive lanes owns one environment, 64/G environments per wave, the whole
// episode in one launch.
//
// Why another mapping: with one wave (or workgroup) per environment a 100-rule base gives every lane 1-2 rules and the
// step is all reductions, barriers and single-lane sections.  Here nothing is reduced: the lanes of a group split the
// A + 1 conclusions a step needs -- Q(s',a) for each action a and Q(s,a) of the pending update -- and each lane walks ALL
// rules in index order, accumulating its Shepard sums sequentially (the reference's own summation order,
// FIVEVagConcl.c:224-235).  The only cross-lane traffic is a handful of __shfl's per step inside the group.
//
// Layout: the rule bases are transposed into `T[tile][k][r][i]` (tile = wave, k = column 0..nant, r = rule, i =
// environment within the tile), so that at a given rule the 64/G environments of a wave read one contiguous run; the
// G lanes of a group read the same address.  frirl_hip_episode_run_lanes imports the canonical slabs, runs, and exports
// them again (two small transposes per call: only the first nrules[e] rules move).
#include "sweeps.h"
#include "envs.h"
#include <cstdlib>

namespace frirl {

// canonical rb[e][k][r]  ->  T[((tile*(nant+1) + k)*maxR + r)*EPW + i]   (e = tile*EPW + i), rules r < nrules[e]
__global__ __launch_bounds__(256) void lanes_import_kernel(const double *__restrict__ rb, const int32_t *__restrict__ nrules, int E, int nant1, int maxR,
                                                            int EPW, double *__restrict__ T)
{
    __shared__ double s[16][65];
    const int tile = blockIdx.x, k = blockIdx.y, e0 = tile * EPW;
    int rmax = 0;
    for (int i = 0; i < EPW; i++) if (e0 + i < E) { const int r = nrules[e0 + i]; rmax = r > rmax ? r : rmax; }
    for (int r0 = 0; r0 < rmax; r0 += 64) {
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, e = e0 + i, r = r0 + j;
            s[i][j] = (e < E && r < maxR) ? rb[((size_t)e * nant1 + k) * maxR + r] : 0.0;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int j = idx / EPW, i = idx - j * EPW, r = r0 + j;
            if (r < maxR) T[(((size_t)tile * nant1 + k) * maxR + r) * EPW + i] = s[i][j];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void lanes_export_kernel(double *__restrict__ rb, const int32_t *__restrict__ nrules, int E, int nant1, int maxR, int EPW,
                                                            const double *__restrict__ T)
{
    __shared__ double s[16][65];
    __shared__ int nr[16];
    const int tile = blockIdx.x, k = blockIdx.y, e0 = tile * EPW;
    if ((int)threadIdx.x < EPW) nr[threadIdx.x] = (e0 + (int)threadIdx.x < E) ? nrules[e0 + threadIdx.x] : 0;
    __syncthreads();
    int rmax = 0;
    for (int i = 0; i < EPW; i++) rmax = nr[i] > rmax ? nr[i] : rmax;
    for (int r0 = 0; r0 < rmax; r0 += 64) {
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int j = idx / EPW, i = idx - j * EPW, r = r0 + j;
            s[i][j] = (r < maxR) ? T[(((size_t)tile * nant1 + k) * maxR + r) * EPW + i] : 0.0;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < EPW * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63, e = e0 + i, r = r0 + j;
            if (e < E && r < nr[i]) rb[((size_t)e * nant1 + k) * maxR + r] = s[i][j];
        }
        __syncthreads();
    }
}

// The per-lane rule loops are chains of dependent global loads (~0.5 us each with one or two waves per SIMD): rules are
// fetched UR at a time into registers, two batches in flight (the next one is requested before the current one is
// consumed).  Out-of-range slots re-read the last rule (never consumed).
constexpr int LN_UR = 4;

template <int NCOL>
struct RuleBatch {
    double c[LN_UR][NCOL];
    __device__ __forceinline__ void load(const double *__restrict__ Te, int maxR, int EPW, int r0, int stride, int R)
    {
#pragma unroll
        for (int j = 0; j < LN_UR; j++) {
            int r = r0 + j * stride;
            r = r < R ? r : R - 1;
            r = r < 0 ? 0 : r;
#pragma unroll
            for (int k = 0; k < NCOL; k++) c[j][k] = Te[(unsigned)((k * maxR + r) * EPW)];
        }
    }
};

// for r = first, first + stride, ... < R (in order): f(r, columns of rule r)
template <int NCOL, class F>
__device__ __forceinline__ void for_rules(const double *__restrict__ Te, int maxR, int EPW, int first, int stride, int R, F &&f)
{
    RuleBatch<NCOL> a, b;
    a.load(Te, maxR, EPW, first, stride, R);
    for (int r0 = first; r0 < R; r0 += 2 * LN_UR * stride) {
        b.load(Te, maxR, EPW, r0 + LN_UR * stride, stride, R);
#pragma unroll
        for (int j = 0; j < LN_UR; j++) { const int r = r0 + j * stride; if (r < R) f(r, a.c[j]); }
        a.load(Te, maxR, EPW, r0 + 2 * LN_UR * stride, stride, R);
#pragma unroll
        for (int j = 0; j < LN_UR; j++) { const int r = r0 + (LN_UR + j) * stride; if (r < R) f(r, b.c[j]); }
    }
}

struct LaneQ {              // one conclusion's raw result
    double v, w;
    unsigned hit;
};

// FIVE_vag_concl's sums for one VE point, all rules, on one lane (sequential, rule order)
template <int NANT>
__device__ __forceinline__ LaneQ lane_sweep_q(const double *__restrict__ Te, int maxR, int EPW, int R, const double (&q)[NANT], int p)
{
    LaneQ o{0.0, 0.0, FRIRL_HIP_NO_HIT};
    for_rules<NANT + 1>(Te, maxR, EPW, 0, 1, R, [&](int r, const double (&c)[NANT + 1]) {
        const double d0 = q[0] - c[0];
        double s = d0 * d0;
#pragma unroll
        for (int k = 1; k < NANT; k++) { const double d = q[k] - c[k]; const double t = d * d; s = s + t; }
        if (s == 0.0) { if (o.hit == FRIRL_HIP_NO_HIT) o.hit = (unsigned)r; }
        else { const double wi = inv_dist_pow(s, p); const double t = wi * c[NANT]; o.v = o.v + t; o.w = o.w + wi; }
    });
    return o;
}

// One wave = 64/G environments.  APL = conclusions per lane: lanes 0..G-2 hold APL actions each ((G-1)*APL >= A), lane
// G-1 holds Q(s,a).
template <int NANT, int APL, int G, int WPE>
__global__ __launch_bounds__(FRIRL_WAVE, WPE) void episode_run_lanes_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                        double *__restrict__ T, uint16_t *__restrict__ uidx, int32_t *__restrict__ nrules,
                                                                        int E, int maxR, const frirl_hip_agent ag, const frirl_hip_envs ev, int nsteps)
{
    constexpr int NS = NANT - 1, EPW = FRIRL_WAVE / G;
    extern __shared__ double tab_s[];                          // [2][NANT][U] when the tables fit, else unused
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    __shared__ double ave_s[FRIRL_HIP_MAX_ACTIONS];
    const int lane = threadIdx.x, sub = lane % G, il = lane / G, base = lane - sub;
    const int tile = blockIdx.x, e = tile * EPW + il;
    const bool exists = e < E;
    const bool in_lds = 2 * sizeof(double) * NANT * (size_t)U <= 16 * 1024;
    if (in_lds) for (int i = lane; i < NANT * U; i += FRIRL_WAVE) { tab_s[i] = u[i]; tab_s[NANT * U + i] = ve[i]; }
    for (int i = lane; i < NANT * FRIRL_HIP_MAX_GRID; i += FRIRL_WAVE) grid_s[i] = ag.grid_values[i];
    if (lane < ag.A) ave_s[lane] = ag.action_ve[lane];
    __syncthreads();
    const double *us = in_lds ? tab_s : u, *ves = in_lds ? tab_s + NANT * U : ve;
    double *Te = T + (size_t)tile * (NANT + 1) * maxR * EPW + il;          // element (k, r) at Te[(k*maxR + r)*EPW]
    const int p = ag.p > 0 ? ag.p : NANT;
    const bool has_q = (sub == G - 1);

    double states[NS], q_ant[NANT], total = 0.0;
    int R = 0, fus = 0, steps = 0, status = FRIRL_HIP_UPD_INACTIVE;
    bool active = false;
    uint32_t episode = 0;
    if (exists) {
#pragma unroll
        for (int k = 0; k < NS; k++) states[k] = ev.states[(size_t)e * NS + k];
#pragma unroll
        for (int k = 0; k < NANT; k++) q_ant[k] = ev.q_ant[(size_t)e * NANT + k];
        R = nrules[e]; fus = ev.fus[e]; steps = ev.ep_steps[e]; total = ev.ep_reward[e];
        active = ev.done[e] == 0;
        episode = ev.episode ? (uint32_t)ev.episode[e] : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < NS; k++) states[k] = 0.0;
#pragma unroll
        for (int k = 0; k < NANT; k++) q_ant[k] = 0.0;
    }
    const bool was_active = active;

    for (int it = 0; it < nsteps; it++) {
        if (!__any(active ? 1 : 0)) break;
        if (active) {
            double cur[NS], cur_q[NANT], reward;
            int success;
            env_do_action(ag.env_kind, q_ant[NS], states, cur);                                         // frirl_episode.c:97
            env_get_reward(ag.env_kind, cur, reward, success);                                          // :106
            env_quantize(ag.env_kind, NS, grid_s, ag.grid_len, ag.grid_div, cur, cur_q);                // :112
            double ve1[NANT], ve2[NS];
#pragma unroll
            for (int k = 0; k < NANT; k++) ve1[k] = observe_ve(us, ves, U, k, q_ant[k]);
#pragma unroll
            for (int k = 0; k < NS; k++) ve2[k] = observe_ve(us, ves, U, k, cur_q[k]);

            // ---- one pass over the rules: this lane's conclusions (frirl_get_best_action :148 / frirl_update_sarsa.c:357)
            double qsel[NS], apt[APL], sv[APL], sw[APL], conc[APL];
            unsigned hit[APL];
#pragma unroll
            for (int k = 0; k < NS; k++) qsel[k] = has_q ? ve1[k] : ve2[k];
            int nacc = has_q ? 1 : ag.A - sub * APL;
            nacc = nacc < 0 ? 0 : (nacc > APL ? APL : nacc);
#pragma unroll
            for (int i = 0; i < APL; i++) {
                const int a = sub * APL + i;
                apt[i] = has_q ? ve1[NS] : ave_s[a < ag.A ? a : 0];
                sv[i] = 0.0; sw[i] = 0.0; hit[i] = FRIRL_HIP_NO_HIT;
            }
            for_rules<NANT + 1>(Te, maxR, EPW, 0, 1, R, [&](int r, const double (&c)[NANT + 1]) {
                const double d0 = qsel[0] - c[0];
                double s = d0 * d0;
#pragma unroll
                for (int k = 1; k < NS; k++) { const double d = qsel[k] - c[k]; const double t = d * d; s = s + t; }
                const double va = c[NS], cq = c[NANT];
#pragma unroll
                for (int i = 0; i < APL; i++) {
                    if (i < nacc) {
                        const double ea = apt[i] - va;
                        const double f = ea * ea;
                        const double d2 = f + s;
                        if (d2 == 0.0) { if (hit[i] == FRIRL_HIP_NO_HIT) hit[i] = (unsigned)r; }
                        else { const double wi = inv_dist_pow(d2, p); const double t = wi * cq; sv[i] = sv[i] + t; sw[i] = sw[i] + wi; }
                    }
                }
            });
            double bv = -__builtin_inf();
            int bi = ag.A;
#pragma unroll
            for (int i = 0; i < APL; i++) {
                conc[i] = 0.0;
                if (i < nacc) {
                    conc[i] = (hit[i] != FRIRL_HIP_NO_HIT) ? Te[((size_t)NANT * maxR + hit[i]) * EPW] : sv[i] / sw[i];
                    const int a = sub * APL + i;
                    if (!has_q && (a == 0 || bv < conc[i])) { bv = conc[i]; bi = a; }                   // first maximum, max.inl:21
                }
            }
            // greedy action over the group's action lanes, in action order
            double cb = __shfl(bv, base);
            int ci = __shfl(bi, base);
#pragma unroll
            for (int g = 1; g < G - 1; g++) {
                const double v = __shfl(bv, base + g);
                const int i2 = __shfl(bi, base + g);
                if (cb < v) { cb = v; ci = i2; }
            }
            const int chosen = e_greedy(ag, ci, (uint32_t)e, episode, (uint32_t)steps + 1u);
            const int slot = chosen % APL;
            double mine = conc[0];
#pragma unroll
            for (int i = 1; i < APL; i++) if (i == slot) mine = conc[i];
            const double qp = __shfl(mine, base + chosen / APL);                                        // Q(s',a'), frirl_update_sarsa.c:356
            const double qnow = __shfl(conc[0], base + G - 1);                                          // Q(s,a), :357
            const double ws1 = __shfl(sw[0], base + G - 1);
            const double vs1 = __shfl(sv[0], base + G - 1);
            const unsigned hit1 = (unsigned)__shfl((int)hit[0], base + G - 1);
            cur_q[NS] = grid_s[NS * FRIRL_HIP_MAX_GRID + chosen];                                        // frirl_episode.c:151

            // ---- frirl_update_sarsa + update_rules (frirl_update_sarsa.c:348-385, :22-143); every lane of the group follows
            //      the same branch, stores are issued by one lane (or split over the lanes for the weighted spread)
            status = FRIRL_HIP_UPD_INACTIVE;
            if (!ag.evaluate) {                                                                         // frirl_episode.c:155
                const double qdiff = ag.alpha * (reward + ag.gamma * qp - qnow);                        // :358
                bool finished = false;
                if (qdiff > ag.qdiff_pos_boundary || qdiff < ag.qdiff_neg_boundary) {                   // :363
                    double rant[NANT], ve3[NANT];
                    unsigned idx3[NANT];
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < NANT; k++) {
                        rant[k] = check_possible_states(q_ant[k], grid_s + k * FRIRL_HIP_MAX_GRID, ag.grid_len[k]);   // :146-170
                        const double *uni = us + (size_t)k * U;
                        idx3[k] = snap_index(uni, U, rant[k], universe_div(uni, U));
                        ve3[k] = ves[(size_t)k * U + idx3[k]];
                        same = same && (ve3[k] == ve1[k]);
                    }
                    LaneQ rr{vs1, ws1, hit1};                                                           // :370 (same VE point => same sums)
                    if (!same) rr = lane_sweep_q<NANT>(Te, maxR, EPW, R, ve3, p);
                    if (rr.hit == FRIRL_HIP_NO_HIT) {                                                   // :373-377 append and leave
                        if (R >= maxR) {
                            status = FRIRL_HIP_UPD_FULL;
                        } else {
                            if (sub == 0) {
#pragma unroll
                                for (int k = 0; k < NANT; k++) {
                                    Te[((size_t)k * maxR + R) * EPW] = ve3[k];                           // five_add_rule.c:80-81
                                    if (uidx) uidx[((size_t)e * NANT + k) * maxR + R] = (uint16_t)idx3[k];   // :76
                                    if (ev.rant) ev.rant[((size_t)e * NANT + k) * maxR + R] = rant[k];
                                }
                                Te[((size_t)NANT * maxR + R) * EPW] = rr.v / rr.w + qdiff;
                            }
                            R++;
                            fus = 1;
                            status = FRIRL_HIP_UPD_INSERTED;
                        }
                        finished = true;
                    } else {
                        fus = 0;                                                                        // :378
                    }
                }
                if (!finished) {
                    const int rules = fus ? R - 1 : R;                                                  // :30-33
                    if (hit1 != FRIRL_HIP_NO_HIT && (ag.skip_rules == 0 || (ag.skip_rules == 1 && (int)hit1 < rules))) {
                        if (sub == 0) Te[((size_t)NANT * maxR + hit1) * EPW] = qnow + qdiff;             // :55
                        status = FRIRL_HIP_UPD_EXACT;
                    } else if (ag.skip_rules == 1 && hit1 != FRIRL_HIP_NO_HIT && (int)hit1 == rules) {
                        status = FRIRL_HIP_UPD_SKIPPED;                                                 // :61-63
                    } else {
                        if (ag.skip_rules == 0) fus = 0;                                                // :70-73
                        const int r_skip = fus ? R - 1 : -1;                                            // :76,124-126
                        const double iws = 1.0 / ws1;
                        double *qc = Te + (size_t)NANT * maxR * EPW;
                        for_rules<NANT>(Te, maxR, EPW, sub, G, R, [&](int r, const double (&c)[NANT]) {      // K6 + K7, rules split over the group
                            const double d0 = ve1[0] - c[0];
                            double s = d0 * d0;
#pragma unroll
                            for (int k = 1; k < NANT; k++) { const double d = ve1[k] - c[k]; const double t = d * d; s = s + t; }
                            const double w = inv_dist_pow(s, p) * iws;
                            if (w > ag.weight_significant && r != r_skip) { const double t = qdiff * w; qc[(size_t)r * EPW] = qnow + t; }
                        });
                        status = FRIRL_HIP_UPD_SPREAD;
                    }
                }
                __threadfence_block();      // the group's stores are visible to its other lanes before the next sweep
            }
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = cur[k]; q_ant[k] = cur_q[k]; }                   // :163-168
            q_ant[NS] = cur_q[NS];
            steps++;                                                                                    // :174
            total = total + reward;                                                                     // :107
            if (success == 1 || steps >= ag.max_steps) active = false;                                  // :183, :86
        }
    }
    if (!exists || sub != 0) return;
    if (ev.status) ev.status[e] = was_active ? status : FRIRL_HIP_UPD_INACTIVE;
    if (!was_active) return;
#pragma unroll
    for (int k = 0; k < NS; k++) ev.states[(size_t)e * NS + k] = states[k];
#pragma unroll
    for (int k = 0; k < NANT; k++) ev.q_ant[(size_t)e * NANT + k] = q_ant[k];
    ev.fus[e] = fus;
    ev.ep_steps[e] = steps;
    ev.ep_reward[e] = total;
    ev.done[e] = active ? 0 : 1;
    nrules[e] = R;
}

}  // namespace frirl

using namespace frirl_host;
int frirl_check_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs, const char *who);

static int lanes_group(int A) { return A <= 3 ? 4 : 8; }
static int lanes_apl(int A) { const int g = lanes_group(A); const int apl = (A + g - 2) / (g - 1); return apl <= 1 ? 1 : (apl <= 3 ? 3 : 5); }

extern "C" size_t frirl_hip_lanes_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A)
{
    if (nant < 1 || E < 1 || maxR < 1 || A < 1) return 0;
    const int epw = FRIRL_WAVE / lanes_group(A);
    const size_t tiles = ((size_t)E + epw - 1) / epw;
    return tiles * epw * (size_t)(nant + 1) * (size_t)maxR * sizeof(double);
}

template <int N, int APL, int G>
static void launch_lanes(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev, int nsteps,
                         double *T, hipStream_t s)
{
    constexpr int EPW = FRIRL_WAVE / G;
    const int tiles = (b->E + EPW - 1) / EPW;
    const size_t tab = 2 * sizeof(double) * N * (size_t)t->U;
    const size_t dyn = tab <= 16 * 1024 ? tab : 0;
    hipLaunchKernelGGL(frirl::lanes_import_kernel, dim3(tiles, N + 1), dim3(256), 0, s, b->rb, b->nrules, b->E, N + 1, b->maxR, EPW, T);
    // registers: 2 waves per SIMD keep both rule batches and the environment state in VGPRs; beyond ~2048 waves (more
    // environments than that can hold at once) 4 waves per SIMD with a few cold values in scratch win
    int wpe = tiles > 2048 ? 4 : 2;
    if (const char *e = getenv("FRIRL_HIP_LANES_WPE")) { const int v = atoi(e); if (v == 2 || v == 3 || v == 4) wpe = v; }
#define LANES_GO(W)                                                                                                                                \
    hipLaunchKernelGGL((frirl::episode_run_lanes_kernel<N, APL, G, W>), dim3(tiles), dim3(FRIRL_WAVE), dyn, s, t->u, t->ve, t->U, T, b->uidx, b->nrules, \
                       b->E, b->maxR, *ag, *ev, nsteps)
    if (wpe == 4) LANES_GO(4); else if (wpe == 3) LANES_GO(3); else LANES_GO(2);
#undef LANES_GO
    hipLaunchKernelGGL(frirl::lanes_export_kernel, dim3(tiles, N + 1), dim3(256), 0, s, b->rb, b->nrules, b->E, N + 1, b->maxR, EPW, T);
}

extern "C" int frirl_hip_episode_run_lanes(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                           const frirl_hip_envs *envs, int32_t nsteps, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = frirl_check_episode(t, b, agent, envs, "frirl_hip_episode_run_lanes");
    if (rc) return rc;
    if (nsteps < 0 || !workspace) { set_error("frirl_hip_episode_run_lanes: nsteps=%d / workspace=%p", nsteps, workspace); return FRIRL_HIP_EINVAL; }
    const size_t need = frirl_hip_lanes_workspace_bytes(t->nant, b->E, b->maxR, agent->A);
    if (workspace_bytes < need) { set_error("frirl_hip_episode_run_lanes: workspace %zu B < %zu B (frirl_hip_lanes_workspace_bytes)", workspace_bytes, need); return FRIRL_HIP_EINVAL; }
    hipStream_t s = as_stream(stream);
    double *T = static_cast<double *>(workspace);
    const int G = lanes_group(agent->A), apl = lanes_apl(agent->A);
#define RUN(N)                                                                     \
    do {                                                                           \
        if (G == 4) launch_lanes<N, 1, 4>(t, b, agent, envs, nsteps, T, s);        \
        else if (apl == 1) launch_lanes<N, 1, 8>(t, b, agent, envs, nsteps, T, s); \
        else if (apl == 3) launch_lanes<N, 3, 8>(t, b, agent, envs, nsteps, T, s); \
        else launch_lanes<N, 5, 8>(t, b, agent, envs, nsteps, T, s);               \
    } while (0)
    if (t->nant == 3) RUN(3); else RUN(5);
#undef RUN
    return check_launch("frirl_hip_episode_run_lanes");
}
