// shared.hip -- many observations / environments against ONE read-only rule base (evaluation mode, SURVEY 8f #3, and
// the "try-remove" replays of the rule-base reduction, 8f #1).
//
// Lane = observation (or environment).  A workgroup stages tiles of the rule base in LDS (every rule is fetched from
// HBM/L2 once per workgroup and reused by its 256 lanes), each lane walks the tile's rules in index order and
// accumulates its own Shepard sums sequentially -- the reference's summation order (FIVEVagConcl.c:224-235,
// FIVEVagConcl_FRIRL_BestAct.c:212-217) -- so no reductions are needed.  Compute-bound: ~24 FP64 instructions per
// (lane, rule, action).
#include "sweeps.h"
#include "envs.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <numeric>
#include <vector>

namespace frirl {

constexpr int SH_TILE = 256;   // rules per LDS tile
constexpr int SH_BLOCK = 256;

template <int NANT>
struct SharedTile {
    double col[(NANT + 1) * SH_TILE];
    uint8_t slot[SH_TILE];
    double ave[FRIRL_HIP_MAX_ACTIONS];
};

// One lane's conclusions against the whole shared rule base; every lane of the workgroup must call it (barriers).
//   GBA: q[] holds the nant-1 state VE points, the action VE points come from tl.ave; conclusions of all A actions go
//        to conc[0..A) (if non-NULL) and the first maximum (frirl_get_best_action.c:60-75) is returned in bi.
//   !GBA: q[] holds all nant VE points; conc[0] / hit0 are FIVE_vag_concl's result.
//   EXCL: rules whose candidate slot s (slot_g[r], 255 = none) has bit s set in `mask` are treated as removed
//        (same sums and same first-hit ORDER as the compacted rule base: removal keeps the relative rule order,
//        five_remove_rule.c:29-85).
//   H > 1: H lanes (G apart, slice index h) share every conclusion of this lane: lane h takes the rules r = h (mod H); the
//        partial sums are added in slice order and the lowest exact hit wins (latency form for few environments).
template <int NANT, int AMAX, bool GBA, bool EXCL, int G = 1, int H = 1, class POW = PowU>
__device__ __forceinline__ void shared_sweep(SharedTile<NANT> &tl, const double *__restrict__ rb, const uint8_t *__restrict__ slot_g, int R,
                                             int maxR, POW p, int abeg, int aend, int nchunks, const double *q, bool live, uint32_t mask,
                                             double *conc, unsigned &hit0, int &bi, double &bvout, int h = 0)
{
    constexpr int NS = NANT - 1;
    constexpr int ND = GBA ? NS : NANT;
    const double *qcol = rb + (size_t)NANT * maxR;
    const int nact = GBA ? aend : 1;
    // running first maximum over this lane's actions [abeg, aend): `bv < c` as max.inl:21; action 0 always seeds it (so a
    // NaN there sticks, as in the reference), a lane that starts later seeds with -inf and skips NaNs
    double bv = -__builtin_inf();
    bi = abeg;
    hit0 = FRIRL_HIP_NO_HIT;
    const auto pk = pin_pow(p);            // series coefficients of the Shepard weight in registers (sweeps.h)
    // actions in chunks of AMAX accumulators (A = 21: three passes over the L2-resident rule base keep the kernel at
    // ~90 VGPRs instead of 254)
    // `nchunks` is uniform over the workgroup (the tile staging below has barriers); a lane with fewer actions idles
    for (int c = 0; c < nchunks; c++) {
        const int a0 = (GBA ? abeg : 0) + c * AMAX;
        const int left = nact - a0;
        const int nacc = left < 0 ? 0 : (left < AMAX ? left : AMAX);
        double sv[AMAX], sw[AMAX];
        unsigned sh[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; a++) { sv[a] = 0.0; sw[a] = 0.0; sh[a] = FRIRL_HIP_NO_HIT; }
        for (int r0 = 0; r0 < R; r0 += SH_TILE) {
            const int n = (R - r0 < SH_TILE) ? R - r0 : SH_TILE;
            __syncthreads();
            for (int i = threadIdx.x; i < (NANT + 1) * SH_TILE; i += SH_BLOCK) {
                const int k = i / SH_TILE, r = i - k * SH_TILE;
                tl.col[i] = (r < n) ? rb[(size_t)k * maxR + r0 + r] : 0.0;
            }
            if (EXCL) for (int r = threadIdx.x; r < SH_TILE; r += SH_BLOCK) tl.slot[r] = (r < n) ? slot_g[r0 + r] : (uint8_t)255;
            __syncthreads();
            if (live && nacc > 0) {
                // branch-free body (selects on the exact-hit test and on the try-remove mask): straight-line code per rule
                for (int r = h; r < n; r += H) {
                    bool valid = true;
                    if (EXCL) { const unsigned sl = tl.slot[r]; valid = !(sl < 32u && ((mask >> sl) & 1u)); }
                    double d0 = q[0] - tl.col[r];
                    double s = d0 * d0;
#pragma unroll
                    for (int k = 1; k < ND; k++) { const double d = q[k] - tl.col[k * SH_TILE + r]; s = __fma_rn(d, d, s); }
                    const double cq = tl.col[NANT * SH_TILE + r];
                    // a removed rule gets a huge squared distance once (its weight vanishes: the sums keep the bits of the compacted
                    // rule base); an exact hit is noted with a select and poisons the sums of its own conclusion, which are then not
                    // read (sweeps.h: q_pair) -- no select around the weight
                    if (EXCL) s = valid ? s : NO_RULE_STATE_PART;
                    if (GBA) {
                        const double va = tl.col[NS * SH_TILE + r];
#pragma unroll
                        for (int a = 0; a < AMAX; a++) {
                            if (a < nacc) {
                                const double e = tl.ave[a0 + a] - va;
                                const double d2 = __fma_rn(e, e, s);
                                const double wi = shepard_w(d2, pk);
                                sv[a] = __fma_rn(wi, cq, sv[a]);
                                sw[a] = sw[a] + wi;
                                sh[a] = (d2 == 0.0 && sh[a] == FRIRL_HIP_NO_HIT) ? (unsigned)(r0 + r) : sh[a];
                            }
                        }
                    } else {
                        const double wi = shepard_w(s, pk);
                        sv[0] = __fma_rn(wi, cq, sv[0]);
                        sw[0] = sw[0] + wi;
                        sh[0] = (s == 0.0 && sh[0] == FRIRL_HIP_NO_HIT) ? (unsigned)(r0 + r) : sh[0];
                    }
                }
            }
        }
        if (H > 1 && live) {             // combine the H rule slices (all lanes of an environment are live together)
            const int lane = threadIdx.x & (FRIRL_WAVE - 1);
            const int first = lane - h * G;
#pragma unroll
            for (int a = 0; a < AMAX; a++) {
                double tv = __shfl(sv[a], first), tw = __shfl(sw[a], first);
                unsigned th = (unsigned)__shfl((int)sh[a], first);
#pragma unroll
                for (int hh = 1; hh < H; hh++) {
                    const double v = __shfl(sv[a], first + hh * G), w = __shfl(sw[a], first + hh * G);
                    const unsigned x = (unsigned)__shfl((int)sh[a], first + hh * G);
                    tv = tv + v;
                    tw = tw + w;
                    th = x < th ? x : th;
                }
                sv[a] = tv; sw[a] = tw; sh[a] = th;
            }
        }
        if (live) {
#pragma unroll
            for (int a = 0; a < AMAX; a++) {
                if (a < nacc) {
                    const double c = (sh[a] != FRIRL_HIP_NO_HIT) ? qcol[sh[a]] : sv[a] / sw[a];
                    if (conc) conc[a0 + a] = c;
                    if (a0 + a == 0 || bv < c) { bv = c; bi = a0 + a; }
                }
            }
            if (c == 0) hit0 = sh[0];
        }
    }
    bvout = bv;
}

template <int NANT, int AMAX, bool GBA>
__global__ __launch_bounds__(SH_BLOCK) void shared_q_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                             const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR, int p,
                                                             int Q, const double *__restrict__ x, const double *__restrict__ action_ve, int A,
                                                             double *__restrict__ conc, uint32_t *__restrict__ hit, int32_t *__restrict__ best)
{
    constexpr int ND = GBA ? NANT - 1 : NANT;    // observation dimensions supplied by the caller
    __shared__ SharedTile<NANT> tl;
    const int qi = blockIdx.x * SH_BLOCK + threadIdx.x;
    const bool live = qi < Q;
    double q[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) q[k] = live ? observe_ve(u, ve, U, k, x[(size_t)qi * ND + k]) : 0.0;
    if (GBA && (int)threadIdx.x < A) tl.ave[threadIdx.x] = action_ve[threadIdx.x];
    unsigned h0;
    int bi;
    double bv;
    shared_sweep<NANT, AMAX, GBA, false>(tl, rb, nullptr, nrules[0], maxR, PowU{p}, 0, A, GBA ? (A + AMAX - 1) / AMAX : 1, q, live, 0u,
                                         live ? conc + (size_t)qi * (GBA ? A : 1) : nullptr, h0, bi, bv);
    if (!live) return;
    if (GBA) best[qi] = bi;
    else hit[qi] = h0;
}

// First maximum over the G lanes that share one environment (consecutive lanes of one wave, each holding the first
// maximum of its own block of actions): combined in block order with the reference's `bv < c` (max.inl:21).
template <int G>
__device__ __forceinline__ void group_first_max(double &bv, int &bi)
{
    if (G == 1) return;
    const int lane = threadIdx.x & (FRIRL_WAVE - 1);
    const int base = lane - (lane % G);
    double cb = __shfl(bv, base);
    int ci = __shfl(bi, base);
#pragma unroll
    for (int g = 1; g < G; g++) {
        const double v = __shfl(bv, base + g);
        const int i = __shfl(bi, base + g);
        if (cb < v) { cb = v; ci = i; }
    }
    bv = cb;
    bi = ci;
}

// frirl_test_run's episode (src/frirl/frirl_test_run.c:66-70: construct_rb = 0, reduction_state = 1, frirl_episode) for
// Q environments sharing one rule base, the whole roll-out in one launch.  No SARSA update (frirl_episode.c:155), so the
// rule base stays read-only and can be shared.  G consecutive lanes serve one environment, each evaluating its own
// block of actions (G = 1 for throughput when Q fills the chip; G = 4 / 8 cut the per-step latency when Q is small, the
// case of the reduction's replays); the environment state is kept redundantly by all G lanes.
// PN: Shepard power = the default nant as a compile-time constant; else the agent's run-time power (instantiated without rule slices)
template <int NANT, int AMAX, int G, int H, bool EXCL, bool PN = true>
__global__ __launch_bounds__(SH_BLOCK) void rollout_shared_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                   const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR,
                                                                   const frirl_hip_agent ag, int Q, const frirl_hip_rollout ro,
                                                                   const unsigned *__restrict__ run_if)
{
    constexpr int NS = NANT - 1, GH = G * H, EPB = SH_BLOCK / GH;
    if (run_if && *run_if == 0u) return;          // the resident form (rollout.hip) has served this call
    __shared__ SharedTile<NANT> tl;
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    const int gl = threadIdx.x % GH, sub = gl % G, h = gl / G;      // group lane = (rule slice h, action slot sub)
    const int qi = blockIdx.x * EPB + threadIdx.x / GH;
    const bool exists = qi < Q;
    const int R = nrules[0];
    using POW = typename std::conditional<PN, PowC<NANT>, PowU>::type;
    POW p;
    if constexpr (!PN) p.p = ag.p > 0 ? ag.p : NANT;
    const int apl = (ag.A + G - 1) / G;                              // actions per lane
    const int abeg = (sub * apl < ag.A) ? sub * apl : ag.A;
    const int aend = (abeg + apl < ag.A) ? abeg + apl : ag.A;
    const int nchunks = (apl + AMAX - 1) / AMAX;
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += SH_BLOCK) grid_s[i] = ag.grid_values[i];
    if ((int)threadIdx.x < ag.A) tl.ave[threadIdx.x] = ag.action_ve[threadIdx.x];
    const uint32_t mask = (EXCL && exists && ro.exclude_mask) ? ro.exclude_mask[qi] : 0u;
    double states[NS], cur[NS], qs[NS], q[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        states[k] = (exists && ro.start_states) ? ro.start_states[(size_t)qi * NS + k] : ag.values_def[k];   // frirl_episode.c:46-48
        q[k] = observe_ve(u, ve, U, k, states[k]);
    }
    unsigned h0;
    int a;
    double bv;
    shared_sweep<NANT, AMAX, true, EXCL, G, H, POW>(tl, rb, ro.rule_slot, R, maxR, p, abeg, aend, nchunks, q, exists, mask, nullptr, h0, a, bv, h);   // :78 (un-quantised start state)
    group_first_max<G>(bv, a);
    a = e_greedy(ag, a, (uint32_t)qi, 0u, 0u);
    double action = grid_s[NS * FRIRL_HIP_MAX_GRID + a];                                                 // :82
    int steps = 0, success = 0;
    double total = 0.0;
    bool active = exists;
    for (int step = 1; step <= ag.max_steps; step++) {                                                   // :86
        if (__syncthreads_count(active ? 1 : 0) == 0) break;                                             // every lane's episode has ended
        if (active) {
            double r;
            env_do_action(ag.env_kind, action, states, cur);                                             // :97
            env_get_reward(ag.env_kind, cur, r, success);                                                // :106
            total = total + r;                                                                           // :107
            env_quantize(ag.env_kind, NS, grid_s, ag.grid_len, ag.grid_div, cur, qs);                    // :112
#pragma unroll
            for (int k = 0; k < NS; k++) q[k] = observe_ve(u, ve, U, k, qs[k]);
        }
        int pa;
        shared_sweep<NANT, AMAX, true, EXCL, G, H, POW>(tl, rb, ro.rule_slot, R, maxR, p, abeg, aend, nchunks, q, active, mask, nullptr, h0, pa, bv, h);   // :148
        group_first_max<G>(bv, pa);
        if (active) {
            pa = e_greedy(ag, pa, (uint32_t)qi, 0u, (uint32_t)step);
            action = grid_s[NS * FRIRL_HIP_MAX_GRID + pa];                                               // :151
#pragma unroll
            for (int k = 0; k < NS; k++) states[k] = cur[k];                                             // :163-165
            steps++;                                                                                     // :174
            if (success == 1) active = false;                                                            // :183
        }
    }
    if (!exists || gl != 0) return;
    ro.steps[qi] = steps;
    ro.reward[qi] = total;
    if (ro.success) ro.success[qi] = success;
    if (ro.final_states)
        for (int k = 0; k < NS; k++) ro.final_states[(size_t)qi * NS + k] = states[k];
}

}  // namespace frirl

using namespace frirl_host;

static int check_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int Q, const char *who)
{
    int rc = check_rulebases(t, b);
    if (rc) return rc;
    if (b->E != 1) { set_error("%s: needs ONE shared rule base (E == 1), got E=%d", who, b->E); return FRIRL_HIP_EINVAL; }
    if (Q < 1) { set_error("%s: Q=%d < 1", who, Q); return FRIRL_HIP_EINVAL; }
    if (t->nant < 2 || t->nant > 9) { set_error("%s: nant=%d outside 2..9", who, t->nant); return FRIRL_HIP_EINVAL; }
    return check_device();
}

extern "C" int five_hip_vag_concl_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, int32_t Q, const double *x,
                                         double *conc, uint32_t *hit, void *stream)
{
    int rc = check_shared(t, b, Q, "five_hip_vag_concl_shared");
    if (rc) return rc;
    if (!x || !conc || !hit) { set_error("five_hip_vag_concl_shared: NULL argument"); return FRIRL_HIP_EINVAL; }
    const int pp = p > 0 ? p : t->nant;
    const dim3 grid((Q + frirl::SH_BLOCK - 1) / frirl::SH_BLOCK);
    switch (t->nant) {
#define M(N) case N: hipLaunchKernelGGL((frirl::shared_q_kernel<N, 1, false>), grid, dim3(frirl::SH_BLOCK), 0, as_stream(stream), t->u, t->ve, t->U, b->rb, \
                                        b->nrules, b->maxR, pp, Q, x, nullptr, 1, conc, hit, nullptr); break;
        M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#undef M
    }
    return check_launch("five_hip_vag_concl_shared");
}

extern "C" int frirl_hip_get_best_action_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, int32_t Q, const double *states,
                                                const double *action_ve, int A, double *actconc, int32_t *best, void *stream)
{
    int rc = check_shared(t, b, Q, "frirl_hip_get_best_action_shared");
    if (rc) return rc;
    if (!states || !action_ve || !actconc || !best || A < 1 || A > FRIRL_HIP_MAX_ACTIONS) { set_error("frirl_hip_get_best_action_shared: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int pp = p > 0 ? p : t->nant;
    const dim3 grid((Q + frirl::SH_BLOCK - 1) / frirl::SH_BLOCK);
    switch (t->nant) {
#define M(N)                                                                                                                                              \
    case N:                                                                                                                                               \
        if (A <= 4) hipLaunchKernelGGL((frirl::shared_q_kernel<N, 4, true>), grid, dim3(frirl::SH_BLOCK), 0, as_stream(stream), t->u, t->ve, t->U, b->rb,   \
                                       b->nrules, b->maxR, pp, Q, states, action_ve, A, actconc, nullptr, best);                                          \
        else hipLaunchKernelGGL((frirl::shared_q_kernel<N, 8, true>), grid, dim3(frirl::SH_BLOCK), 0, as_stream(stream), t->u, t->ve, t->U, b->rb,         \
                                b->nrules, b->maxR, pp, Q, states, action_ve, A, actconc, nullptr, best);                                                 \
        break;
        M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#undef M
    }
    return check_launch("frirl_hip_get_best_action_shared");
}

template <int N, int AMAX, int G, int H, bool PN = true>
static void launch_rollout(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                           hipStream_t s, const unsigned *run_if)
{
    constexpr int EPB = frirl::SH_BLOCK / (G * H);
    const dim3 grid((Q + EPB - 1) / EPB);
    if (ro->exclude_mask && ro->rule_slot)
        hipLaunchKernelGGL((frirl::rollout_shared_kernel<N, AMAX, G, H, true, PN>), grid, dim3(frirl::SH_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, Q, *ro, run_if);
    else
        hipLaunchKernelGGL((frirl::rollout_shared_kernel<N, AMAX, G, H, false, PN>), grid, dim3(frirl::SH_BLOCK), 0, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, Q, *ro, run_if);
}

// lanes per environment: 1 once the environments alone fill the chip, else split the actions over 4 (A <= 4) or 8 lanes
static int rollout_group(int Q, int A)
{
    { const int g = frirl_host::opts().rollout_group; if (g == 1 || (g == 4 && A <= 4) || (g == 8 && A > 4)) return g; }
    if (A < 2 || Q >= 131072) return 1;
    return A <= 4 ? 4 : 8;
}

// rule slices per conclusion (G > 1 only): the replays of the reduction run ~1000 environments, one step is then a
// latency chain over the rules -- 4 or 8 lanes share it while the launch stays under ~2048 waves
static int rollout_slices(int Q, int G)
{
    if (G == 1) return 1;
    { const int v = frirl_host::opts().rollout_slices; if (v == 1 || v == 4 || v == 8) return v; }
    const long waves1 = ((long)Q * G + 63) / 64;
    return waves1 * 8 <= 2048 ? 8 : (waves1 * 4 <= 2048 ? 4 : 1);
}

template <int N, int AMAX, int G>
static void launch_rollout_h(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                             hipStream_t s, const unsigned *run_if)
{
    const int H = rollout_slices(Q, G);
    if (H == 8) launch_rollout<N, AMAX, G, 8>(t, b, ag, Q, ro, s, run_if);
    else if (H == 4) launch_rollout<N, AMAX, G, 4>(t, b, ag, Q, ro, s, run_if);
    else launch_rollout<N, AMAX, G, 1>(t, b, ag, Q, ro, s, run_if);
}

template <int N>
static void launch_rollout_n(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                             hipStream_t s, const unsigned *run_if)
{
    const int G = rollout_group(Q, ag->A);
    if (ag->p > 0 && ag->p != N) {          // run-time Shepard power: the variants without rule slices
        if (G == 4) launch_rollout<N, 1, 4, 1, false>(t, b, ag, Q, ro, s, run_if);
        else if (G == 8) launch_rollout<N, 4, 8, 1, false>(t, b, ag, Q, ro, s, run_if);
        else if (ag->A <= 4) launch_rollout<N, 4, 1, 1, false>(t, b, ag, Q, ro, s, run_if);
        else launch_rollout<N, 8, 1, 1, false>(t, b, ag, Q, ro, s, run_if);
        return;
    }
    if (G == 4) launch_rollout_h<N, 1, 4>(t, b, ag, Q, ro, s, run_if);
    else if (G == 8) launch_rollout_h<N, 4, 8>(t, b, ag, Q, ro, s, run_if);
    else if (ag->A <= 4) launch_rollout<N, 4, 1, 1>(t, b, ag, Q, ro, s, run_if);
    else launch_rollout<N, 8, 1, 1>(t, b, ag, Q, ro, s, run_if);
}

int frirl_rollout_resident(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                           hipStream_t s, const unsigned **too_big_flag, void **workspace);      // rollout.hip

extern "C" int frirl_hip_rollout_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, int32_t Q,
                                        const frirl_hip_rollout *ro, void *stream)
{
    int rc = check_shared(t, b, Q, "frirl_hip_rollout_shared");
    if (rc) return rc;
    if (!agent || !ro || !ro->steps || !ro->reward || !agent->grid_values || !agent->action_ve) { set_error("frirl_hip_rollout_shared: NULL argument"); return FRIRL_HIP_EINVAL; }
    if (agent->A < 1 || agent->A > FRIRL_HIP_MAX_ACTIONS || agent->max_steps < 0) { set_error("frirl_hip_rollout_shared: A=%d / max_steps=%d out of range", agent->A, agent->max_steps); return FRIRL_HIP_EINVAL; }
    for (int k = 0; k < t->nant; k++)
        if (agent->grid_len[k] < 1 || agent->grid_len[k] > FRIRL_HIP_MAX_GRID) { set_error("frirl_hip_rollout_shared: grid_len[%d]=%d outside 1..%d", k, agent->grid_len[k], FRIRL_HIP_MAX_GRID); return FRIRL_HIP_EINVAL; }
    if ((ro->exclude_mask == nullptr) != (ro->rule_slot == nullptr)) { set_error("frirl_hip_rollout_shared: exclude_mask and rule_slot go together"); return FRIRL_HIP_EINVAL; }
    if (agent->env_kind == FRIRL_HIP_ENV_MOUNTAINCAR ? t->nant != 3 : t->nant != 5) {
        set_error("frirl_hip_rollout_shared: env_kind %d does not match nant=%d", agent->env_kind, t->nant);
        return FRIRL_HIP_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    // small rule bases: the LDS-resident, queue-fed form (rollout.hip); the tiled kernel below serves every other shape, and a rule
    // base that turns out not to fit the LDS image (known on the device only: `too_big`)
    const unsigned *too_big = nullptr;
    void *ws = nullptr;
    if (frirl_rollout_resident(t, b, agent, Q, ro, s, &too_big, &ws)) {
        if (too_big) {
            if (t->nant == 3) launch_rollout_n<3>(t, b, agent, Q, ro, s, too_big);
            else launch_rollout_n<5>(t, b, agent, Q, ro, s, too_big);
        }
        if (ws) (void)hipFreeAsync(ws, s);
        return check_launch("frirl_hip_rollout_shared");
    }
    if (t->nant == 3) launch_rollout_n<3>(t, b, agent, Q, ro, s, nullptr);
    else launch_rollout_n<5>(t, b, agent, Q, ro, s, nullptr);
    return check_launch("frirl_hip_rollout_shared");
}

// ---- speculative try-remove reduction (frirl_sequential_run.c:170-350); see include/frirl_hip.h ----------------------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess; }
};
}  // namespace

extern "C" int frirl_hip_reduce_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, double *rant, int strategy,
                                       double reward_tolerance, int depth, int32_t *kept, frirl_hip_reduce_result *result, void *stream)
{
    int rc = check_shared(t, b, 1, "frirl_hip_reduce_shared");
    if (rc) return rc;
    if (!agent || !result) { set_error("frirl_hip_reduce_shared: NULL argument"); return FRIRL_HIP_EINVAL; }
    if (strategy != 1 && strategy != 2) { set_error("frirl_hip_reduce_shared: strategy %d (1 = smallest |Q| first, 2 = largest |Q| first)", strategy); return FRIRL_HIP_EINVAL; }
    if (depth == 0) depth = 10;
    if (depth < 1 || depth > 12) { set_error("frirl_hip_reduce_shared: depth %d outside 1..12", depth); return FRIRL_HIP_EINVAL; }
    hipStream_t s = as_stream(stream);
    const int nant = t->nant, maxR = b->maxR;
    const size_t col = (size_t)maxR;
#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) { set_error("frirl_hip_reduce_shared: %s: %s", #expr, hipGetErrorString(e_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)

    int32_t R0 = 0;
    HIP_TRY(hipMemcpyAsync(&R0, b->nrules, sizeof R0, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (R0 < 1 || R0 > maxR) { set_error("frirl_hip_reduce_shared: nrules=%d outside 1..maxR=%d", R0, maxR); return FRIRL_HIP_EINVAL; }
    std::vector<double> slab((size_t)(nant + 1) * col), rants;
    std::vector<uint16_t> idx;
    HIP_TRY(hipMemcpyAsync(slab.data(), b->rb, slab.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    if (rant) { rants.resize((size_t)nant * col); HIP_TRY(hipMemcpyAsync(rants.data(), rant, rants.size() * sizeof(double), hipMemcpyDeviceToHost, s)); }
    if (b->uidx) { idx.resize((size_t)nant * col); HIP_TRY(hipMemcpyAsync(idx.data(), b->uidx, idx.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, s)); }
    HIP_TRY(hipStreamSynchronize(s));

    // candidate order: the reference rescans for the first minimum (strategy 1, `mvalue > fabs(..)` :268) or the first
    // maximum (strategy 2, :286) of the not-yet-tested consequents after every episode; the consequents never change and
    // removals keep the relative rule order, so that is a stable sort, fixed up front
    const double *qcol = slab.data() + (size_t)nant * col;
    std::vector<int> order(R0);
    std::iota(order.begin(), order.end(), 0);
    if (strategy == 1) std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return std::fabs(qcol[a]) < std::fabs(qcol[c]); });
    else std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return std::fabs(qcol[a]) > std::fabs(qcol[c]); });

    const int lanes_max = (1 << depth) - 1;
    DevBuf d_slot, d_mask, d_steps, d_reward;
    if (!d_slot.alloc(col) || !d_mask.alloc(sizeof(uint32_t) * lanes_max) || !d_steps.alloc(sizeof(int32_t) * lanes_max) ||
        !d_reward.alloc(sizeof(double) * lanes_max)) { set_error("frirl_hip_reduce_shared: hipMalloc failed"); return FRIRL_HIP_ELAUNCH; }
    std::vector<uint8_t> slot(col);
    std::vector<uint32_t> mask(lanes_max);
    std::vector<int32_t> steps(lanes_max);
    std::vector<double> reward(lanes_max);

    // baseline replay on the un-reduced rule base (:196-198 and the first loop iteration, :204-206)
    frirl_hip_rollout ro = {};
    ro.steps = static_cast<int32_t *>(d_steps.p);
    ro.reward = static_cast<double *>(d_reward.p);
    frirl_hip_agent greedy = *agent;
    greedy.no_random = 1;                                             // the replays are greedy (reduction_state == 1 keeps epsilon at 0 in every demo)
    rc = frirl_hip_rollout_shared(t, b, &greedy, 1, &ro, stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(steps.data(), d_steps.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(reward.data(), d_reward.p, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int steps_inc = steps[0];
    double prev_reward = reward[0];
    frirl_hip_agent capped = greedy;
    if (capped.max_steps > steps_inc + 1) capped.max_steps = steps_inc + 1;

    std::vector<int> alive(R0);                                       // original index of the rule in each current slot
    std::iota(alive.begin(), alive.end(), 0);
    std::vector<int> where(R0);                                       // current slot of each original rule, -1 = removed
    int R = R0, rounds = 0, rollouts = 1;
    ro.exclude_mask = static_cast<const uint32_t *>(d_mask.p);
    ro.rule_slot = static_cast<const uint8_t *>(d_slot.p);
    for (int j = 0; j < R0;) {
        const int d = std::min(depth, R0 - j);
        const int lanes = (1 << d) - 1;
        std::fill(where.begin(), where.end(), -1);
        for (int i = 0; i < R; i++) where[alive[i]] = i;
        std::fill(slot.begin(), slot.end(), (uint8_t)255);
        for (int i = 0; i < d; i++) slot[where[order[j + i]]] = (uint8_t)i;
        for (int k = 0; k < d; k++)                                   // node (k, bits): candidates j..j+k-1 had outcomes `bits`, candidate j+k is on trial
            for (uint32_t bits = 0; bits < (1u << k); bits++) mask[(1u << k) - 1 + bits] = bits | (1u << k);
        HIP_TRY(hipMemcpyAsync(d_slot.p, slot.data(), col, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_mask.p, mask.data(), sizeof(uint32_t) * lanes, hipMemcpyHostToDevice, s));
        rc = frirl_hip_rollout_shared(t, b, &capped, lanes, &ro, stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(steps.data(), d_steps.p, sizeof(int32_t) * lanes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(reward.data(), d_reward.p, sizeof(double) * lanes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        rounds++;
        rollouts += lanes;
        uint32_t bits = 0;
        for (int k = 0; k < d; k++) {
            const uint32_t lane = (1u << k) - 1 + bits;
            const double diff = prev_reward - reward[lane];
            if (reward[lane] > agent->reward_good_above && steps[lane] == steps_inc && std::fabs(diff) <= reward_tolerance) {   // :212
                bits |= 1u << k;
                prev_reward = reward[lane];                           // :222
            }
        }
        if (bits) {                                                   // five_remove_rule of every accepted candidate: compact all columns
            std::vector<char> drop(R, 0);
            for (int i = 0; i < d; i++) if ((bits >> i) & 1u) drop[where[order[j + i]]] = 1;
            int w = 0;
            for (int r = 0; r < R; r++) {
                if (drop[r]) continue;
                if (w != r) {
                    for (int k = 0; k <= nant; k++) slab[(size_t)k * col + w] = slab[(size_t)k * col + r];
                    if (rant) for (int k = 0; k < nant; k++) rants[(size_t)k * col + w] = rants[(size_t)k * col + r];
                    if (b->uidx) for (int k = 0; k < nant; k++) idx[(size_t)k * col + w] = idx[(size_t)k * col + r];
                    alive[w] = alive[r];
                }
                w++;
            }
            for (int r = w; r < R; r++) {                             // vacated tail: zero like the reference's memset (five_remove_rule.c:64-80)
                for (int k = 0; k <= nant; k++) slab[(size_t)k * col + r] = 0.0;
                if (rant) for (int k = 0; k < nant; k++) rants[(size_t)k * col + r] = 0.0;
                if (b->uidx) for (int k = 0; k < nant; k++) idx[(size_t)k * col + r] = 0;
            }
            R = w;
            alive.resize(R);
            const int32_t Rn = R;
            HIP_TRY(hipMemcpyAsync(b->rb, slab.data(), slab.size() * sizeof(double), hipMemcpyHostToDevice, s));
            if (rant) HIP_TRY(hipMemcpyAsync(rant, rants.data(), rants.size() * sizeof(double), hipMemcpyHostToDevice, s));
            if (b->uidx) HIP_TRY(hipMemcpyAsync(b->uidx, idx.data(), idx.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(b->nrules, &Rn, sizeof Rn, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        j += d;
    }
#undef HIP_TRY
    if (kept) for (int i = 0; i < R; i++) kept[i] = alive[i];
    result->rules_before = R0;
    result->rules_after = R;
    result->rounds = rounds;
    result->rollouts = rollouts;
    result->steps_incremental = steps_inc;
    result->reserved = 0;
    result->reward = prev_reward;
    return FRIRL_HIP_OK;
}
