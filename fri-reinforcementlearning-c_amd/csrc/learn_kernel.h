#pragma once
// learn_kernel.h -- the construct loop of frirl_sequential_run (reference src/frirl/frirl_sequential_run.c:55-165: episodes until the
// rule base is "considered complete") for MANY agents with small rule bases, as ONE persistent launch per chunk of work: every
// agent runs episode after episode at its own pace -- frirl_episode's loop (frirl_episode.c:28-194), the SARSA update
// (frirl_update_sarsa.c:22-143,348-385), and at each episode's end the loop's own bookkeeping (:68-72,83-148) -- until it has
// converged, has used its step budget for this launch, or has run max_episodes - 1 episodes.
//
// Why not one episode per launch (lanes.hip): agents of a real many-agent run (start states diversified per agent, reference
// frirl_agent.c:121-139) are never in step -- at any moment some are still in their 1000-step failing episodes while others
// finish in 80 steps, and they converge after anything between 300 and 30 000 steps.  A launch that waits for the longest
// episode of the batch idles most lanes most of the time.  Here nothing waits: the launch ends when every agent has spent its
// budget, and between launches the host compacts the agents that are still learning (`live` list) and gives them more lanes
// each (H) as their number shrinks.
//
// Mapping: a GROUP of H consecutive lanes owns one agent; lane h holds the rules r = h (mod H) and evaluates ALL A + 1
// conclusions of a step for them -- Q(s', a) for every action (state part of the squared distance once per rule) and Q(s, a) of
// the pending update -- so nothing is computed twice inside a group except the environment's own dynamics; the H partial sums
// are combined by a butterfly.  Rule bases live in per-wave tiles T[tile][j][lane] (rule r = j H + h of the lane's agent):
// every load of the sweep is one contiguous 64-lane run.  A rule is a packed record of BITS-bit universe indices (4 B for the
// 41-point universes of mountaincar / acrobot, 8 B for cartpole's 1001 points) + its consequent; VE values are gathered from an
// LDS copy of the tables (rb[e][k][r] == ve[k][uidx[e][k][r]] exactly, five_add_rule.c:76-81): 12 B per rule and step from
// HBM / L2 instead of 48.  Rules are walked from the highest index down so that "the lowest exact hit" is simply the last one
// seen.  Sums: per lane in (descending) index order, slices added in butterfly order -- interpolated values, <= 1e-6 contract.
//
// What a step does NOT repeat (round 3, second half; each measured in profiles/r03d_learner.md):
//  * the weighted spread visits only rules the fused sweep FLAGGED as possibly significant (one bit per rule against a lower bound
//    of the weight sum that is known before the sweep: `thr`), not every rule a second time;
//  * the pending point (s, a) is the (s', a') of the step before: its VE values, packed indices and "is a grid point" flag are
//    carried, not re-derived; frirl_check_possible_states of a grid point is the point itself;
//  * a launch ends an agent after the WORK of `budget` mean-agent steps, so that the waves of the large rule bases do not outlast
//    the others (all waves of a launch are resident together: the launch lasts as long as its slowest wave);
//  * more than 8 actions (cartpole: 21): the A + 1 conclusions do not fit the register file at once; the rules are walked in parts of
//    <= 11 conclusions, each part reduced to "best action so far" before the next (`part`).
#include "sweeps.h"
#include "envs.h"
#include <type_traits>

namespace frirl {

#ifndef LEARN_UR
#define LEARN_UR 2      // rules fetched per batch, two batches in flight
#endif
constexpr int LR_BLOCK = 256;
constexpr int LR_PADROWS = 16;     // rule rows in front of every tile that nothing consumes: the prefetch of the descending walk may run past row 0
constexpr int LR_WPB = LR_BLOCK / FRIRL_WAVE;
constexpr int LR_MW = 16;          // 32-bit words of spread-candidate flags per lane (one bit per rule of the lane's slice: 512 rules)

struct LearnArgs {
    const double *u, *ve;        // tables [nant][U]
    int U;
    uint32_t *Ti;                // [tiles][njmax][64][W] packed universe indices
    double *Tq;                  // [tiles][njmax][64]    consequents
    double *Tp;                  // [tiles][njmax][64]    consequents after the previous episode (frirl_sequential_run.c:72)
    const int32_t *live;         // [nlive] agents of this launch, or NULL = 0..nlive-1
    int nlive, tiles, njmax;
    double *rb;                  // canonical slabs: VE columns of appended rules are written through
    uint16_t *uidx;
    int32_t *nrules;
    int maxR;
    int64_t *work;               // [E][2] rule visits of the main sweeps / of the extra sweeps (snapped point, weighted spread), or NULL
    int64_t *steps_total;        // [E] environment steps so far, or NULL
    int budget;                  // environment steps in this launch for an agent with the launch's mean rule count (see work budget)
    long long *sum_rules;        // sum of the live agents' rule counts at the start of the launch (import kernel)
    int max_episodes;            // the loop runs episodes 1 .. max_episodes - 1 (frirl_sequential_run.c:51,59)
#ifdef LEARN_TIMING
    unsigned long long *timing;  // [16] cycles per section of the step, summed over waves (tools/exp builds only)
#endif
};
#ifdef LEARN_TIMING
#define LT_MARK(i) do { const unsigned long long now_ = clock64(); if ((__ffsll((long long)__ballot(1)) - 1) == lane) { tim_s[wave][i] += now_ - tim_s[wave][15]; tim_s[wave][15] = now_; } } while (0)
#else
#define LT_MARK(i) do { } while (0)
#endif

template <int BITS>
struct Packed {
    static constexpr int FPW = 32 / BITS;                       // fields per 32-bit word
    static constexpr int words(int nant) { return (nant + FPW - 1) / FPW; }
};

// ---- import / export: canonical rb[e][nant][r], uidx[e][k][r], prev_rconc[e][r]  <->  tiles --------------------------------
template <int NANT, int BITS>
__global__ __launch_bounds__(256) void learn_import_kernel(const LearnArgs la, int H, const double *__restrict__ prev_rconc)
{
    constexpr int FPW = Packed<BITS>::FPW, W = Packed<BITS>::words(NANT);
    const int tile = blockIdx.x, EPW = FRIRL_WAVE / H;
    const int lane = threadIdx.x & 63, il = lane / H, h = lane % H;
    const int slot = tile * EPW + il;
    const bool exists = slot < la.nlive;
    const int e = exists ? (la.live ? la.live[slot] : slot) : 0;
    const int R = exists ? la.nrules[e] : 0;
    const int nj = (R + H - 1) / H;
    if (threadIdx.x < 64 && h == 0 && exists) atomicAdd(reinterpret_cast<unsigned long long *>(la.sum_rules), (unsigned long long)R);
    int njw = nj;                                               // rounds needed by any agent of the tile
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(njw, off, FRIRL_WAVE); njw = o > njw ? o : njw; }
    for (int j = threadIdx.x / 64; j < njw; j += 4) {
        const int r = j * H + h;
        uint32_t w[W];
#pragma unroll
        for (int x = 0; x < W; x++) w[x] = 0u;
        double q = 0.0, p = 0.0;
        if (r < R) {
#pragma unroll
            for (int k = 0; k < NANT; k++) w[k / FPW] |= (uint32_t)la.uidx[((size_t)e * NANT + k) * la.maxR + r] << (BITS * (k % FPW));
            q = la.rb[((size_t)e * (NANT + 1) + NANT) * la.maxR + r];
            p = prev_rconc[(size_t)e * la.maxR + r];
        }
        const size_t o = ((size_t)tile * (la.njmax + LR_PADROWS) + LR_PADROWS + j) * 64 + lane;
#pragma unroll
        for (int x = 0; x < W; x++) la.Ti[o * W + x] = w[x];
        la.Tq[o] = q;
        la.Tp[o] = p;
    }
}

template <int NANT>
__global__ __launch_bounds__(256) void learn_export_kernel(const LearnArgs la, int H, double *__restrict__ prev_rconc)
{
    constexpr int nant = NANT;
    const int tile = blockIdx.x, EPW = FRIRL_WAVE / H;
    const int lane = threadIdx.x & 63, il = lane / H, h = lane % H;
    const int slot = tile * EPW + il;
    const bool exists = slot < la.nlive;
    const int e = exists ? (la.live ? la.live[slot] : slot) : 0;
    const int R = exists ? la.nrules[e] : 0;
    const int nj = (R + H - 1) / H;
    int njw = nj;
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(njw, off, FRIRL_WAVE); njw = o > njw ? o : njw; }
    for (int j = threadIdx.x / 64; j < njw; j += 4) {
        const int r = j * H + h;
        if (r < R) {
            const size_t o = ((size_t)tile * (la.njmax + LR_PADROWS) + LR_PADROWS + j) * 64 + lane;
            la.rb[((size_t)e * (nant + 1) + nant) * la.maxR + r] = la.Tq[o];
            prev_rconc[(size_t)e * la.maxR + r] = la.Tp[o];
        }
    }
}

// ---- the learner ------------------------------------------------------------------------------------------------------
template <int NANT, int NA, int KIND, int H, int BITS, int WPS>
__global__ __launch_bounds__(LR_BLOCK, WPS) void learn_kernel(const LearnArgs la, const frirl_hip_agent ag, const frirl_hip_envs ev,
                                                              const frirl_hip_convergence cv)
{
    constexpr int NS = NANT - 1, EPW = FRIRL_WAVE / H;                        // NA + 1 conclusions per step: A actions at s', Q(s, a)
    constexpr int FPW = Packed<BITS>::FPW, W = Packed<BITS>::words(NANT);
    constexpr int UR = LEARN_UR;                                        // rules fetched per batch, two batches in flight
    using RecI = typename std::conditional<W == 1, uint32_t, typename std::conditional<W == 2, uint2, uint4>::type>::type;
    // STATIC LDS, so that every address is a compile-time constant: a table entry is read at  8 * index  with the table row's address
    // as the instruction's immediate offset (no base register to add).  Tables have a fixed row stride of 64 entries (6-bit indices).
    static_assert(BITS == 6 || BITS == 10, "the LDS tables are laid out for 6-bit (<= 64 points) or 10-bit (<= 1024 points: cartpole) universe indices");
    constexpr int TS = 1 << BITS;
    constexpr int MW = BITS == 6 ? LR_MW : LR_MW / 2;                         // flag words per lane (the 40 KB tables of cartpole leave less LDS)
    constexpr bool LU = KIND != FRIRL_HIP_ENV_CARTPOLE;                          // small tables: the universes in LDS too
    constexpr int NCOLD = 2 * NS + 2 * NANT + 3 + 4;                          // cold per-agent state parked during every sweep (see below)
    __shared__ double tab_s[(LU ? 2 : 1) * NANT * TS];
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    __shared__ double udiv[8];                                                // FIVEInit.c:244-248, once instead of per observation
    __shared__ unsigned aidx_s[NA];                                           // universe index of every action value
    extern __shared__ __attribute__((aligned(16))) double cold_s[];          // [LR_WPB * EPW][NCOLD]: dynamic (51 KB at one lane per agent: beyond the static limit)
    __shared__ uint32_t mask_s[MW * LR_BLOCK];                             // [word][thread]: spread candidates of the sweep (see `thr`)
    static_assert(32 % (2 * UR) == 0, "a flag word is filled by whole loop iterations");
    const int U = la.U, maxR = la.maxR;
    const int lane = threadIdx.x & (FRIRL_WAVE - 1), wave = threadIdx.x / FRIRL_WAVE;
    const int il = lane / H, h = lane % H;
    const int tile = blockIdx.x * LR_WPB + wave;
    const int slot = tile * EPW + il;
    const bool exists = tile < la.tiles && slot < la.nlive;
    const int e = exists ? (la.live ? la.live[slot] : slot) : 0;
    for (int i = threadIdx.x; i < NANT * U; i += LR_BLOCK) tab_s[(i / U) * TS + i % U] = la.ve[i];
    if (LU) for (int i = threadIdx.x; i < NANT * U; i += LR_BLOCK) tab_s[NANT * TS + (i / U) * TS + i % U] = la.u[i];
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += LR_BLOCK) grid_s[i] = ag.grid_values[i];
    __syncthreads();
    const double *ves = tab_s, *us;
    if constexpr (LU) us = tab_s + NANT * TS; else us = la.u;
    const int USTR = LU ? TS : U;                                             // row stride of `us`
    if ((int)threadIdx.x < NANT) udiv[threadIdx.x] = universe_div(us + (size_t)threadIdx.x * USTR, U);
    __syncthreads();
    // VE value of x in dimension k and the universe index it snaps to (five_rule_distance.c:75,80)
    auto locate = [&](int k, double x, unsigned &idx) { const double *uni = us + (size_t)k * USTR; idx = snap_index(uni, U, x, udiv[k]); return ves[(size_t)k * TS + idx]; };
    if ((int)threadIdx.x < NA) aidx_s[threadIdx.x] = snap_index(us + (size_t)NS * USTR, U, grid_s[NS * FRIRL_HIP_MAX_GRID + threadIdx.x], udiv[NS]);
    __syncthreads();
    auto pack = [&](uint32_t (&w)[W], int k, unsigned idx) { w[k / FPW] |= (idx & ((1u << BITS) - 1u)) << (BITS * (k % FPW)); };
    const size_t tbase = ((size_t)(tile < la.tiles ? tile : 0) * (la.njmax + LR_PADROWS) + LR_PADROWS) * 64;
    const RecI *Ti_l = reinterpret_cast<const RecI *>(la.Ti) + tbase + lane;      // this lane's rules: element j at [j * 64]
    double *Tq_l = la.Tq + tbase + lane, *Tp_l = la.Tp + tbase + lane;
    RecI *Ti_g = reinterpret_cast<RecI *>(la.Ti) + tbase + il * H;                // the group's rule r: [(r / H) * 64 + r % H]
    double *Tq_g = la.Tq + tbase + il * H;
    const auto pk = pin_pow(PowC<NANT>());

    auto decode = [&](const RecI &x, double (&c)[NANT]) {
        uint32_t w[W];
        if constexpr (W == 1) { w[0] = x; } else if constexpr (W == 2) { w[0] = x.x; w[1] = x.y; } else { w[0] = x.x; w[1] = x.y; w[2] = x.z; if constexpr (W > 3) w[3] = x.w; }
#pragma unroll
        for (int k = 0; k < NANT; k++) {
            constexpr uint32_t FM = ((1u << BITS) - 1u) << 3;
            const int sh = BITS * (k % FPW);
            const uint32_t off = (sh >= 3 ? (w[k / FPW] >> (sh - 3)) : (w[k / FPW] << (3 - sh))) & FM;     // 8 * index
            c[k] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(ves + k * TS) + off);
        }
    };
    // f(r, VE values of rule r, its consequent) for this lane's rules r = j H + h < R, from the highest j down; NEEDQ: load consequents
    // rules jtop, jtop - 1, ..., jtop - UR + 1 of this lane: ONE address per array, the rest are immediate offsets; rows below 0 are
    // the tile's padding (loaded, never consumed)
    auto fetch = [&](RecI(&xi)[UR], double(&xq)[UR], int jtop, bool needq) {
        const RecI *pi = Ti_l + (long)jtop * 64;
        const double *pq = Tq_l + (long)jtop * 64;
#pragma unroll
        for (int t = 0; t < UR; t++) {
            xi[t] = pi[-t * 64];
            xq[t] = needq ? pq[-t * 64] : 0.0;
        }
    };
    // The FIRST batch of the main sweep is requested by slice_prefetch before the environment's own dynamics, so that its global loads are
    // in flight meanwhile (measured neutral at two waves per SIMD: 0.476 vs 0.483 counted; the same for the weighted spread's walk cost
    // four spilled registers and was dropped -- profiles/r03c_learn_prefetch.jsonl).
    struct Pref { RecI i[UR]; double q[UR]; int jt; };
    auto slice_prefetch = [&](int R, bool needq) {
        Pref p;
        p.jt = (R - h + H - 1) / H - 1;                                       // this lane's rules: j = 0 .. jt (jt < 0: none)
        if (p.jt < -1) p.jt = -1;
        fetch(p.i, p.q, p.jt, needq);
        return p;
    };
    // FLAGS: f returns a word whose sign bit says "this rule may carry a significant weight of the pending conclusion"; the bits are
    // collected MSB-first, 32 rules per word, in mask_s[word][thread] (walk position q = jt0 - j -> word q / 32, bit 31 - q % 32)
    auto for_slice_from = [&](Pref &p, bool needq, auto &&f, auto flags) {
        constexpr bool FL = decltype(flags)::value;
        RecI ib[UR];
        double qb[UR];
        uint32_t cmask = 0u;
        auto consume = [&](const RecI(&xi)[UR], const double(&xq)[UR], int jtop) {
#pragma unroll
            for (int t = 0; t < UR; t++) {
                const int j = jtop - t;
                if constexpr (FL) {
                    uint32_t hi = 0u;
                    if (j >= 0) { double c[NANT]; decode(xi[t], c); hi = f(j * H + h, c, xq[t]); }
                    cmask = __builtin_amdgcn_alignbit(cmask, hi, 31);       // (cmask << 1) | (hi >> 31)
                } else {
                    if (j >= 0) { double c[NANT]; decode(xi[t], c); f(j * H + h, c, xq[t]); }
                }
            }
        };
        int jt = p.jt;
        const int jt0 = jt, jtm = jt & 31;
        while (jt >= 0) {
            fetch(ib, qb, jt - UR, needq);
            consume(p.i, p.q, jt);
            fetch(p.i, p.q, jt - 2 * UR, needq);
            consume(ib, qb, jt - UR);
            jt -= 2 * UR;
            if constexpr (FL) {
                if ((jt & 31) == jtm) {                                        // 32 more rules walked: a full word
                    const int wd = ((jt0 - jt) >> 5) - 1;
                    if (wd < MW) mask_s[wd * LR_BLOCK + threadIdx.x] = cmask;
                }
            }
        }
        if constexpr (FL) {
            const int done = jt0 - jt, rem = done & 31;                        // jt0 < 0: nothing walked
            if (jt0 >= 0 && rem != 0 && (done >> 5) < MW) mask_s[(done >> 5) * LR_BLOCK + threadIdx.x] = cmask << (32 - rem);
        }
    };
    auto for_slice = [&](int R, bool needq, auto &&f) {
        Pref p = slice_prefetch(R, needq);
        for_slice_from(p, needq, f, std::false_type());
    };
    auto group_or = [&](bool v) -> bool {
        const unsigned long long b = __ballot(v ? 1 : 0);
        if constexpr (H == 64) return b != 0ull;
        else return ((b >> (il * H)) & ((1ull << H) - 1ull)) != 0ull;
    };

    // ---- per-agent state (replicated in the H lanes of the group) ----------------------------------------------------------
    double states[NS], q_ant[NANT], total = 0.0, prev_reward = 0.0;
    int R = 0, fus = 0, steps = 0, prevR = 0, prev_steps = 0, nep = 0, lsteps = 0;
    bool active = false, begin = false, converged = false, refused = false;
    uint32_t episode = 0;
    long long wmain = 0, wextra = 0;
    // Spread candidates.  The weighted spread (frirl_update_sarsa.c:76-131) moves the rules whose weight w_r / sum(w) for the pending point
    // (s, a) exceeds weight_significant.  sum(w) of this step's pending conclusion is known from BELOW before the sweep starts: it is the
    // sum the previous sweep formed for the action it then chose (same point, same rules in the same order; a rule appended since only adds
    // a non-negative term, and rounding is monotone).  `thr` = weight_significant (less a 4e-9 margin) x that sum: the sweep flags the rules
    // with w_r > thr, one bit each, and the spread visits the flagged rules only -- the same arithmetic on a superset of the rules the full
    // walk would move.  0 = unknown (first step after a launch boundary): the spread walks all rules.
    double thr = 0.0;
    // Work budget.  All waves of a launch are resident together, so the launch lasts as long as its slowest wave; agents are sorted by rule
    // count (a wave's agents walk as many rules as the largest rule base among them), so with the same number of steps for everyone the
    // waves of the large rule bases would finish long after the others.  Each agent therefore stops after the WORK of `budget` steps of an
    // agent with the launch's mean rule count -- rules walked by its fused sweeps plus a per-step constant (the environment's own dynamics,
    // butterfly and update are worth ~24 rules per lane) -- but never after more than 4 x budget steps.  Where a launch cuts an agent's
    // run does not change what it computes (tests/test_hip_learn.py).
    constexpr long long STEP_RULES = 24 * H;
    const long long work_budget = (long long)la.budget * (*la.sum_rules / (la.nlive > 0 ? la.nlive : 1) + STEP_RULES);
    // The pending point (s, a) of a step is the (s', a') of the step before: its VE values, its packed universe indices and "every
    // coordinate is a grid value" (true for everything env_quantize and the action grid produce: frirl_check_possible_states then returns
    // its argument, so the snapped point of frirl_update_sarsa.c:366-370 is the point itself) are carried over instead of recomputed.
    double ve1[NANT];
    uint32_t pk1[W];
    bool ongrid = false;                                                      // unknown after a launch boundary and for a start state
#pragma unroll
    for (int k = 0; k < NS; k++) states[k] = exists ? ev.states[(size_t)e * NS + k] : 0.0;
#pragma unroll
    for (int k = 0; k < NANT; k++) q_ant[k] = exists ? ev.q_ant[(size_t)e * NANT + k] : 0.0;
#pragma unroll
    for (int x = 0; x < W; x++) pk1[x] = 0u;
#pragma unroll
    for (int k = 0; k < NANT; k++) { unsigned i1; ve1[k] = locate(k, q_ant[k], i1); pack(pk1, k, i1); }
    if (exists) {
        R = la.nrules[e]; fus = ev.fus[e]; steps = ev.ep_steps[e]; total = ev.ep_reward[e];
        begin = ev.done[e] != 0;                                              // between two episodes: start the next one
        episode = ev.episode ? (uint32_t)ev.episode[e] : 0u;
        prevR = cv.prev_nrules[e]; prev_steps = cv.prev_steps[e]; prev_reward = cv.prev_reward[e];
        converged = cv.converged[e] != 0; nep = cv.episodes[e];
        active = !converged && nep < la.max_episodes - 1 && la.budget > 0;
    }

#ifdef LEARN_TIMING
    __shared__ unsigned long long tim_s[LR_WPB][16];
    if (lane < 16) tim_s[wave][lane] = 0ull;
    if (lane == 0) tim_s[wave][15] = clock64();
#endif
    while (__any(active ? 1 : 0)) {
        double cur[NS], cur_q[NANT], ve2[NS], reward = 0.0;
        uint32_t pk2[W];
        int success = 0;
#pragma unroll
        for (int x = 0; x < W; x++) pk2[x] = 0u;
        Pref pmain = slice_prefetch(active ? R : 0, true);                    // in flight while the environment steps
        if (active) {
            if (begin) {                                                      // frirl_episode.c:46-48: q_states = states = start state
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const double v = ev.start_states ? ev.start_states[(size_t)e * NS + k] : ag.values_def[k];
                    states[k] = v; cur[k] = v; cur_q[k] = v; q_ant[k] = v;
                }
                steps = 0; total = 0.0;
            } else {
                env_do_action(KIND, q_ant[NS], states, cur);                                     // :97
                env_get_reward(KIND, cur, reward, success);                                      // :106
                env_quantize(KIND, NS, grid_s, ag.grid_len, ag.grid_div, cur, cur_q);            // :112
            }
#pragma unroll
            for (int k = 0; k < NS; k++) { unsigned i2; ve2[k] = locate(k, cur_q[k], i2); pack(pk2, k, i2); }
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) { ve2[k] = 0.0; cur[k] = 0.0; cur_q[k] = 0.0; }
        }

        LT_MARK(0);
        // ---- the agent's cold state leaves the registers for the duration of the sweep: it is the same in all H lanes of the group,
        //      so lane 0 parks it in LDS and every lane reads it back afterwards (the compiler may not forward across the fences);
        //      without this the kernel needs > 256 VGPRs and spills to scratch inside the rule loop
        double *cold = cold_s + (size_t)(wave * EPW + il) * NCOLD;
        if (h == 0) {
#pragma unroll
            for (int k = 0; k < NS; k++) { cold[k] = states[k]; cold[NS + k] = cur[k]; }
#pragma unroll
            for (int k = 0; k < NANT; k++) { cold[2 * NS + k] = q_ant[k]; cold[2 * NS + NANT + k] = cur_q[k]; }
            cold[2 * NS + 2 * NANT] = total; cold[2 * NS + 2 * NANT + 1] = prev_reward; cold[2 * NS + 2 * NANT + 2] = reward;
            int *ci = reinterpret_cast<int *>(cold + 2 * NS + 2 * NANT + 3);
            ci[0] = fus; ci[1] = steps; ci[2] = prevR; ci[3] = prev_steps; ci[4] = nep; ci[5] = lsteps; ci[6] = (int)episode; ci[7] = success;
        }
        asm volatile("" ::: "memory");
        LT_MARK(1);

        // ---- this lane's rules: Q(s', a) for every action (frirl_get_best_action, :148) and Q(s, a) of the pending update
        //      (frirl_update_sarsa.c:357); an exact hit poisons the sums of its own conclusion, which are then not read.  Up to 8 actions:
        //      ONE walk with all A + 1 conclusions in registers.  More (cartpole: 21): the conclusions do not fit the register file at
        //      once, so the rules are walked in PARTS of <= 11 conclusions each (state part and decode repeated per part: ~10 %); every
        //      part reduces its conclusions to "best action so far" before the next one starts.
        double bv = 0.0, swb = 0.0;                                           // greedy: first maximum in action order (max.inl:21), its weight sum
        int ci = 0;
        unsigned hit1 = FRIRL_HIP_NO_HIT;                                     // the pending conclusion
        double vs1 = 0.0, ws1 = 0.0;
        auto part = [&](auto a0c, auto napc, auto withpc, Pref &pf) {
            constexpr int A0 = decltype(a0c)::value, NAP = decltype(napc)::value;
            constexpr bool WITHP = decltype(withpc)::value;
            constexpr int NCP = NAP + (WITHP ? 1 : 0);
            double sv[NCP], sw[NCP], pave[NAP];
            unsigned sh[NCP];
#pragma unroll
            for (int i = 0; i < NCP; i++) { sv[i] = 0.0; sw[i] = 0.0; sh[i] = FRIRL_HIP_NO_HIT; }
#pragma unroll
            for (int a = 0; a < NAP; a++) pave[a] = ag.action_ve[A0 + a];
            if (active) {
                for_slice_from(pf, true, [&](int r, const double (&c)[NANT], double cq) {
                    const double e0 = ve2[0] - c[0];
                    double s2 = e0 * e0, s1 = 0.0;
                    if constexpr (WITHP) { const double g0 = ve1[0] - c[0]; s1 = g0 * g0; }
#pragma unroll
                    for (int k = 1; k < NS; k++) {
                        const double d2 = ve2[k] - c[k];
                        s2 = __fma_rn(d2, d2, s2);
                        if constexpr (WITHP) { const double d1 = ve1[k] - c[k]; s1 = __fma_rn(d1, d1, s1); }
                    }
                    const double va = c[NS];
#pragma unroll
                    for (int a = 0; a < NAP; a++) {
                        const double ea = pave[a] - va;
                        const double d = __fma_rn(ea, ea, s2);
                        sh[a] = (d == 0.0) ? (unsigned)r : sh[a];             // descending walk: the last hit seen is the lowest
                        const double wi = shepard_w(d, pk);
                        sv[a] = __fma_rn(wi, cq, sv[a]);
                        sw[a] = sw[a] + wi;
                    }
                    if constexpr (WITHP) {
                        const double e1 = ve1[NS] - va;
                        const double d = __fma_rn(e1, e1, s1);
                        sh[NAP] = (d == 0.0) ? (unsigned)r : sh[NAP];
                        const double wi = shepard_w(d, pk);
                        sv[NAP] = __fma_rn(wi, cq, sv[NAP]);
                        sw[NAP] = sw[NAP] + wi;
                        return (uint32_t)__double2hiint(thr - wi);            // sign bit: w_r > thr
                    } else {
                        return 0u;
                    }
                }, std::integral_constant<bool, WITHP>());
            }
            if (H > 1) {                                                      // the H rule slices of every conclusion, a few conclusions at a time
                constexpr int CH = 4;
#pragma unroll
                for (int i0 = 0; i0 < NCP; i0 += CH) {
                    for (int off = 1; off < H; off <<= 1) {
                        double tv[CH], tw[CH];
                        unsigned th[CH];
#pragma unroll
                        for (int i = 0; i < CH; i++) if (i0 + i < NCP) { tv[i] = __shfl_xor(sv[i0 + i], off, FRIRL_WAVE); tw[i] = __shfl_xor(sw[i0 + i], off, FRIRL_WAVE); th[i] = (unsigned)__shfl_xor((int)sh[i0 + i], off, FRIRL_WAVE); }
#pragma unroll
                        for (int i = 0; i < CH; i++) if (i0 + i < NCP) { sv[i0 + i] = sv[i0 + i] + tv[i]; sw[i0 + i] = sw[i0 + i] + tw[i]; sh[i0 + i] = th[i] < sh[i0 + i] ? th[i] : sh[i0 + i]; }
                    }
                }
            }
            if (active) {
#pragma unroll
                for (int a = 0; a < NAP; a++) {
                    const double c = (sh[a] != FRIRL_HIP_NO_HIT) ? Tq_g[(size_t)(sh[a] / H) * 64 + (sh[a] % H)] : sv[a] / sw[a];
                    if (A0 + a == 0 || bv < c) { bv = c; ci = A0 + a; swb = sw[a]; }
                }
                if constexpr (WITHP) { hit1 = sh[NAP]; vs1 = sv[NAP]; ws1 = sw[NAP]; }
            }
        };
        if constexpr (NA <= 8) {
            part(std::integral_constant<int, 0>(), std::integral_constant<int, NA>(), std::true_type(), pmain);
        } else {
            constexpr int NA1 = (NA + 1) / 2;
            part(std::integral_constant<int, 0>(), std::integral_constant<int, NA1>(), std::false_type(), pmain);
            Pref p2 = slice_prefetch(active ? R : 0, true);
            part(std::integral_constant<int, NA1>(), std::integral_constant<int, NA - NA1>(), std::true_type(), p2);
        }
        if (active) wmain += R;
        LT_MARK(2);
        // Q(s', ax), its weight sum and exact hit for an action that is not the greedy one (an exploratory choice): its own walk
        auto action_conclusion = [&](int ax, double &v, double &w, unsigned &hh) {
            const double avx = ag.action_ve[ax];
            v = 0.0; w = 0.0; hh = FRIRL_HIP_NO_HIT;
            for_slice(R, true, [&](int r, const double (&c)[NANT], double cq) {
                const double e0 = ve2[0] - c[0];
                double s2 = e0 * e0;
#pragma unroll
                for (int k = 1; k < NS; k++) { const double d2 = ve2[k] - c[k]; s2 = __fma_rn(d2, d2, s2); }
                const double ea = avx - c[NS];
                const double d = __fma_rn(ea, ea, s2);
                hh = (d == 0.0) ? (unsigned)r : hh;
                const double wi = shepard_w(d, pk);
                v = __fma_rn(wi, cq, v);
                w = w + wi;
            });
            for (int off = 1; off < H; off <<= 1) {
                const double tv = __shfl_xor(v, off, FRIRL_WAVE), tw = __shfl_xor(w, off, FRIRL_WAVE);
                const unsigned th = (unsigned)__shfl_xor((int)hh, off, FRIRL_WAVE);
                v = v + tv; w = w + tw; hh = th < hh ? th : hh;
            }
            wextra += R;
        };
        asm volatile("" ::: "memory");
        {
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = cold[k]; cur[k] = cold[NS + k]; }
#pragma unroll
            for (int k = 0; k < NANT; k++) { q_ant[k] = cold[2 * NS + k]; cur_q[k] = cold[2 * NS + NANT + k]; }
            total = cold[2 * NS + 2 * NANT]; prev_reward = cold[2 * NS + 2 * NANT + 1]; reward = cold[2 * NS + 2 * NANT + 2];
            const int *cint = reinterpret_cast<const int *>(cold + 2 * NS + 2 * NANT + 3);
            fus = cint[0]; steps = cint[1]; prevR = cint[2]; prev_steps = cint[3]; nep = cint[4]; lsteps = cint[5]; episode = (uint32_t)cint[6]; success = cint[7];
        }
        LT_MARK(3);
        if (active) {
            if (begin) {
                episode++;
                const int a0 = e_greedy(ag, ci, (uint32_t)e, episode, 0u);                              // :78-82
                q_ant[NS] = grid_s[NS * FRIRL_HIP_MAX_GRID + a0];
                begin = false;
                double w0 = swb;
                if (a0 != ci) { double v0; unsigned h0; action_conclusion(a0, v0, w0, h0); }
                thr = (ag.weight_significant * (1.0 - 4e-9)) * w0;
#pragma unroll
                for (int k = 0; k < NS; k++) ve1[k] = ve2[k];                                           // (start state, a0) is the next pending point
                ve1[NS] = ves[NS * TS + aidx_s[a0]];
#pragma unroll
                for (int x = 0; x < W; x++) pk1[x] = pk2[x];
                pack(pk1, NS, aidx_s[a0]);
                {                                                                                       // a start state may lie anywhere:
                    double sq[NS];                                                                      // a grid value quantizes to itself
                    env_quantize(KIND, NS, grid_s, ag.grid_len, ag.grid_div, q_ant, sq);
                    ongrid = KIND != FRIRL_HIP_ENV_CARTPOLE;
#pragma unroll
                    for (int k = 0; k < NS; k++) ongrid = ongrid && (sq[k] == q_ant[k]);
                }
            } else {
                const int chosen = e_greedy(ag, ci, (uint32_t)e, episode, (uint32_t)steps + 1u);
                const bool flags_known = thr > 0.0 && thr < __builtin_inf();                            // this step's flags were taken against a real bound
                double qp = bv, w0 = swb;                                                                // Q(s',a'), frirl_update_sarsa.c:356
                if (chosen != ci) {                                                                      // an exploratory action: its own conclusion
                    double v;
                    unsigned hh;
                    action_conclusion(chosen, v, w0, hh);
                    qp = (hh != FRIRL_HIP_NO_HIT) ? Tq_g[(size_t)(hh / H) * 64 + (hh % H)] : v / w0;
                }
                thr = (ag.weight_significant * (1.0 - 4e-9)) * w0;                                      // (s', a') is the next pending point
                const double qnow = (hit1 != FRIRL_HIP_NO_HIT) ? Tq_g[(size_t)(hit1 / H) * 64 + (hit1 % H)] : vs1 / ws1;   // Q(s,a), :357
                cur_q[NS] = grid_s[NS * FRIRL_HIP_MAX_GRID + chosen];                                    // frirl_episode.c:151

                // ---- frirl_update_sarsa + update_rules (frirl_update_sarsa.c:348-385, :22-143): every lane of the group follows the
                //      same branch; single stores are issued by lane 0, the weighted spread by every lane for its own rules
                LT_MARK(4);
                if (!ag.evaluate) {                                                                     // frirl_episode.c:155
                    const double qdiff = ag.alpha * (reward + ag.gamma * qp - qnow);                    // :358
                    bool finished = false;
                    if (qdiff > ag.qdiff_pos_boundary || qdiff < ag.qdiff_neg_boundary) {               // :363
                        double rant[NANT], ve3[NANT];
                        uint32_t pk3[W];
#pragma unroll
                        for (int k = 0; k < NANT; k++) { rant[k] = q_ant[k]; ve3[k] = ve1[k]; }
#pragma unroll
                        for (int x = 0; x < W; x++) pk3[x] = pk1[x];
                        double v3 = vs1, w3 = ws1;                                                      // :370 (same VE point => same sums)
                        unsigned hit3 = hit1;
                        if (!ongrid) {                                                                  // :146-170, else the point is its own snap
                            bool same = true;
#pragma unroll
                            for (int x = 0; x < W; x++) pk3[x] = 0u;
#pragma unroll
                            for (int k = 0; k < NANT; k++) {
                                rant[k] = check_possible_states(q_ant[k], grid_s + k * FRIRL_HIP_MAX_GRID, ag.grid_len[k]);
                                unsigned i3;
                                ve3[k] = locate(k, rant[k], i3);
                                pack(pk3, k, i3);
                                same = same && (ve3[k] == ve1[k]);
                            }
                            if (!same) {
                                v3 = 0.0; w3 = 0.0; hit3 = FRIRL_HIP_NO_HIT;
                                for_slice(R, true, [&](int r, const double (&c)[NANT], double cq) {
                                    const double d0 = ve3[0] - c[0];
                                    double s = d0 * d0;
#pragma unroll
                                    for (int k = 1; k < NANT; k++) { const double d = ve3[k] - c[k]; s = __fma_rn(d, d, s); }
                                    hit3 = (s == 0.0) ? (unsigned)r : hit3;
                                    const double wi = shepard_w(s, pk);
                                    v3 = __fma_rn(wi, cq, v3);
                                    w3 = w3 + wi;
                                });
                                for (int off = 1; off < H; off <<= 1) {
                                    const double tv = __shfl_xor(v3, off, FRIRL_WAVE), tw = __shfl_xor(w3, off, FRIRL_WAVE);
                                    const unsigned th = (unsigned)__shfl_xor((int)hit3, off, FRIRL_WAVE);
                                    v3 = v3 + tv; w3 = w3 + tw; hit3 = th < hit3 ? th : hit3;
                                }
                                wextra += R;
                            }
                        }
                        if (hit3 == FRIRL_HIP_NO_HIT) {                                                 // :373-377 append and leave
                            if (R >= maxR) {
                                refused = true;
                            } else {
                                if (h == 0) {
                                    RecI x;
                                    if constexpr (W == 1) { x = pk3[0]; } else if constexpr (W == 2) { x.x = pk3[0]; x.y = pk3[1]; } else { x.x = pk3[0]; x.y = pk3[1]; x.z = pk3[2]; x.w = W > 3 ? pk3[W - 1] : 0u; }
                                    const size_t o = (size_t)(R / H) * 64 + (R % H);
                                    Ti_g[o] = x;
                                    Tq_g[o] = v3 / w3 + qdiff;
#pragma unroll
                                    for (int k = 0; k < NANT; k++) {
                                        la.rb[((size_t)e * (NANT + 1) + k) * maxR + R] = ve3[k];        // five_add_rule.c:80-81 (canonical slab)
                                        la.uidx[((size_t)e * NANT + k) * maxR + R] = (uint16_t)((pk3[k / FPW] >> (BITS * (k % FPW))) & ((1u << BITS) - 1u));  // :76
                                        if (ev.rant) ev.rant[((size_t)e * NANT + k) * maxR + R] = rant[k];
                                    }
                                }
                                R++;
                                fus = 1;
                            }
                            finished = true;
                        } else {
                            fus = 0;                                                                    // :378
                        }
                    }
                    LT_MARK(5);
                    if (!finished) {
                        const int rules = fus ? R - 1 : R;                                              // :30-33
                        if (hit1 != FRIRL_HIP_NO_HIT && (ag.skip_rules == 0 || (ag.skip_rules == 1 && (int)hit1 < rules))) {
                            if (h == 0) Tq_g[(size_t)(hit1 / H) * 64 + (hit1 % H)] = qnow + qdiff;       // :55
                        } else if (ag.skip_rules == 1 && hit1 != FRIRL_HIP_NO_HIT && (int)hit1 == rules) {
                            // :61-63 hit on the just-inserted rule: skipped
                        } else {
                            if (ag.skip_rules == 0) fus = 0;                                            // :70-73
                            const int r_skip = fus ? R - 1 : -1;                                        // :76,124-126
                            if (hit1 == FRIRL_HIP_NO_HIT && h == 0 && ev.spread_ant) {   // this call defines FIVERB.weights from now on (frirl_hip.h)
#pragma unroll
                                for (int k = 0; k < NANT; k++) ev.spread_ant[(size_t)e * NANT + k] = q_ant[k];
                                if (ev.spread_R) ev.spread_R[e] = R;
                            }
                            const double iws = 1.0 / ws1;
                            auto move = [&](int r, const double (&c)[NANT]) {                           // K6 + K7 for one rule of this lane
                                const double d0 = ve1[0] - c[0];
                                double s = d0 * d0;
#pragma unroll
                                for (int k = 1; k < NANT; k++) { const double d = ve1[k] - c[k]; s = __fma_rn(d, d, s); }
                                const double w = shepard_w(s, pk) * iws;
                                if (w > ag.weight_significant && r != r_skip) { const double t = qdiff * w; Tq_l[(size_t)(r / H) * 64] = qnow + t; }
                            };
                            const int jt0 = (R - h + H - 1) / H - 1;                                    // the walk the flags were taken on
                            if (group_or(!flags_known || jt0 >= MW * 32)) {
                                for_slice(R, false, [&](int r, const double (&c)[NANT], double) { move(r, c); });   // every lane its own rules
                                wextra += R;
                            } else {
                                // this lane's flagged rules (rarely more than two or three); their records are then loaded together
                                constexpr int NCD = 8;
                                int js[NCD], nc = 0;
#pragma unroll
                                for (int i = 0; i < NCD; i++) js[i] = 0;
                                const int nw = (jt0 + 32) >> 5;
                                for (int wd = 0; wd < nw; wd++) {
                                    uint32_t m = mask_s[wd * LR_BLOCK + threadIdx.x];
                                    while (m) {
                                        const int b = __clz(m);
                                        m &= ~(0x80000000u >> b);
                                        const int j = jt0 - (wd * 32 + b);
#pragma unroll
                                        for (int i = 0; i < NCD; i++) js[i] = (nc == i) ? j : js[i];
                                        nc++;
                                    }
                                }
                                if (group_or(nc > NCD)) {
                                    for_slice(R, false, [&](int r, const double (&c)[NANT], double) { move(r, c); });
                                    wextra += R;
                                } else {
                                    RecI xs[NCD];
#pragma unroll
                                    for (int i = 0; i < NCD; i++) xs[i] = Ti_l[(long)js[i] * 64];
#pragma unroll
                                    for (int i = 0; i < NCD; i++) if (i < nc) { double c[NANT]; decode(xs[i], c); move(js[i] * H + h, c); }
                                }
                            }
                        }
                    }
                    __threadfence_block();          // the group's stores are visible to its other lanes before the next sweep
                    LT_MARK(6);
                }
#pragma unroll
                for (int k = 0; k < NS; k++) { states[k] = cur[k]; q_ant[k] = cur_q[k]; ve1[k] = ve2[k]; }   // :163-168
                q_ant[NS] = cur_q[NS];
                ve1[NS] = ves[NS * TS + aidx_s[chosen]];
#pragma unroll
                for (int x = 0; x < W; x++) pk1[x] = pk2[x];
                pack(pk1, NS, aidx_s[chosen]);
                ongrid = KIND != FRIRL_HIP_ENV_CARTPOLE;                                               // env_quantize's grid values (envs.h)
                steps++;                                                                                // :174
                lsteps++;
                total = total + reward;                                                                 // :107
                if (success == 1 || steps >= ag.max_steps) {                                            // :183, :86 -- the episode is over
                    // frirl_sequential_run.c:83-148: "RB considered complete" = same #rules, #steps and (good) reward as the previous
                    // episode and no consequent moved by the tolerance or more; then the loop's snapshot (:68-72)
                    const bool same = prevR == R && prev_steps == steps && total > ag.reward_good_above && prev_reward == total;
                    bool moved = false;
                    const int nj = (R - h + H - 1) / H;
                    constexpr int EB = 16;                                    // loads of a batch in flight together
                    for (int j0 = 0; j0 < nj; j0 += EB) {
                        double qn[EB];
#pragma unroll
                        for (int t = 0; t < EB; t++) qn[t] = Tq_l[(size_t)(j0 + t < nj ? j0 + t : 0) * 64];
                        if (same) {
                            double qp[EB];
#pragma unroll
                            for (int t = 0; t < EB; t++) qp[t] = Tp_l[(size_t)(j0 + t < nj ? j0 + t : 0) * 64];
#pragma unroll
                            for (int t = 0; t < EB; t++) if (j0 + t < nj && fabs(qn[t] - qp[t]) >= ag.qdiff_final_tolerance) moved = true;
                        }
#pragma unroll
                        for (int t = 0; t < EB; t++) if (j0 + t < nj) Tp_l[(size_t)(j0 + t) * 64] = qn[t];
                    }
                    moved = group_or(moved);
                    nep++;
                    if (same && !moved) converged = true;
                    prevR = R; prev_steps = steps; prev_reward = total;
                    begin = true;
                    if (converged || nep >= la.max_episodes - 1) active = false;
                }
                LT_MARK(7);
                if (wmain + (long long)lsteps * STEP_RULES >= work_budget || lsteps >= 4 * la.budget) active = false;
            }
        }
        LT_MARK(8);
    }
#ifdef LEARN_TIMING
    if (lane < 15 && la.timing) atomicAdd(&la.timing[lane], tim_s[wave][lane]);
#endif

    if (!exists || h != 0) return;
#pragma unroll
    for (int k = 0; k < NS; k++) ev.states[(size_t)e * NS + k] = states[k];
#pragma unroll
    for (int k = 0; k < NANT; k++) ev.q_ant[(size_t)e * NANT + k] = q_ant[k];
    ev.fus[e] = fus;
    ev.ep_steps[e] = steps;
    ev.ep_reward[e] = total;
    ev.done[e] = begin ? 1 : 0;
    if (ev.episode) ev.episode[e] = (int32_t)episode;
    if (ev.status) ev.status[e] = refused ? FRIRL_HIP_UPD_FULL : FRIRL_HIP_UPD_INACTIVE;
    la.nrules[e] = R;
    cv.prev_nrules[e] = prevR; cv.prev_steps[e] = prev_steps; cv.prev_reward[e] = prev_reward;
    cv.converged[e] = converged ? 1 : 0; cv.episodes[e] = nep;
    if (la.work) { la.work[2 * (size_t)e] += wmain; la.work[2 * (size_t)e + 1] += wextra; }
    if (la.steps_total) la.steps_total[e] += lsteps;
}

}  // namespace frirl


template <int N, int NA, int KIND, int H, int BITS, int WPS>
inline void launch_learn(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                         const frirl_hip_convergence *cv, frirl::LearnArgs la, hipStream_t s)
{
    constexpr int EPW = FRIRL_WAVE / H, W = frirl::Packed<BITS>::words(N);
    la.tiles = (la.nlive + EPW - 1) / EPW;
    la.njmax = (b->maxR + H - 1) / H;
    const size_t n = (size_t)la.tiles * (la.njmax + frirl::LR_PADROWS) * 64;
    char *ws = reinterpret_cast<char *>(la.Ti);
    la.Tq = reinterpret_cast<double *>(ws + ((n * W * sizeof(uint32_t) + 15) / 16) * 16);
    la.Tp = la.Tq + n;
    la.sum_rules = reinterpret_cast<long long *>(la.Tp + n);                  // inside the 256 B the workspace size adds at its end
    (void)hipMemsetAsync(la.sum_rules, 0, sizeof(long long), s);
    constexpr int NCOLD_ = 2 * (N - 1) + 2 * N + 3 + 4;                        // = learn_kernel's NCOLD
    const size_t dyn = sizeof(double) * frirl::LR_WPB * EPW * NCOLD_;
    if (dyn > 40 * 1024) {                                                     // static (~24 KB) + dynamic beyond the 64 KB default
        static bool raised = false;                                            // per instantiation
        if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(frirl::learn_kernel<N, NA, KIND, H, BITS, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); raised = true; }
    }
    hipLaunchKernelGGL((frirl::learn_import_kernel<N, BITS>), dim3(la.tiles), dim3(256), 0, s, la, H, cv->prev_rconc);
    const int blocks = (la.tiles + frirl::LR_WPB - 1) / frirl::LR_WPB;
    hipLaunchKernelGGL((frirl::learn_kernel<N, NA, KIND, H, BITS, WPS>), dim3(blocks), dim3(frirl::LR_BLOCK), dyn, s, la, *ag, *ev, *cv);
    hipLaunchKernelGGL((frirl::learn_export_kernel<N>), dim3(la.tiles), dim3(256), 0, s, la, H, cv->prev_rconc);
}


// one environment's launcher for the lane-group sizes [HLO, HHI]: each instantiation file (learn_i*.hip) compiles a few kernels, so
// that the build runs them in parallel
template <int N, int NA, int KIND, int BITS, int WPS, int HLO, int HHI>
inline void launch_learn_h(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                           const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s)
{
    if constexpr (HLO <= 1 && 1 <= HHI) if (H == 1) return launch_learn<N, NA, KIND, 1, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 2 && 2 <= HHI) if (H == 2) return launch_learn<N, NA, KIND, 2, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 4 && 4 <= HHI) if (H == 4) return launch_learn<N, NA, KIND, 4, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 8 && 8 <= HHI) if (H == 8) return launch_learn<N, NA, KIND, 8, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 16 && 16 <= HHI) if (H == 16) return launch_learn<N, NA, KIND, 16, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 32 && 32 <= HHI) if (H == 32) return launch_learn<N, NA, KIND, 32, BITS, WPS>(t, b, ag, ev, cv, la, s);
    if constexpr (HLO <= 64 && 64 <= HHI) if (H == 64) return launch_learn<N, NA, KIND, 64, BITS, WPS>(t, b, ag, ev, cv, la, s);
}

// defined in learn_i0.hip (mountaincar), learn_i1.hip (acrobot, 1 .. 8 lanes per agent), learn_i2.hip (acrobot, 16 .. 64),
// learn_i3.hip / learn_i4.hip (cartpole, 2 .. 8 / 16 .. 64)
void frirl_learn_launch_cartpole_lo(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                    const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s);
void frirl_learn_launch_cartpole_hi(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                    const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s);
void frirl_learn_launch_mountaincar(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                    const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s);
void frirl_learn_launch_acrobot_lo(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                   const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s);
void frirl_learn_launch_acrobot_hi(int H, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const frirl_hip_envs *ev,
                                   const frirl_hip_convergence *cv, const frirl::LearnArgs &la, hipStream_t s);
