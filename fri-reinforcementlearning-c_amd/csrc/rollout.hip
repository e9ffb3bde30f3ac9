// rollout.hip -- frirl_test_run's greedy episode (reference src/frirl/frirl_test_run.c:66-70 -> frirl_episode.c:28-194 with
// reduction_state == 1) for Q environments on ONE small read-only rule base: the RESIDENT form of frirl_hip_rollout_shared.
//
// What limits a roll-out launch is not the arithmetic but the episode lengths: on the acrobot demo the mean episode is 82
// steps, 2 % take more than 160 and a few per ten thousand never succeed and run to max_steps = 1000.  With one lane per
// environment and the whole batch resident (shared.hip: rollout_shared_kernel) a launch lasts as long as its longest episode
// and most lanes idle most of the time.  Here instead:
//   * the whole rule base lives in LDS (<= ~1200 rules of 5 antecedents), staged once per workgroup: no barrier and no
//     re-staging per step, every wave runs at its own pace;
//   * a GROUP of H consecutive lanes owns one environment: lane h evaluates the rules r = h (mod H) for ALL actions (state
//     part of the squared distance once per rule, A accumulator pairs per lane), the H partial sums are combined by a
//     butterfly inside the group -- the arithmetic of one step is split H ways without recomputing anything but the
//     environment's own dynamics;
//   * groups are fed from an in-order queue (one atomic per wave and refill): a group whose episode ends takes the next
//     environment at once, so the chip works on min(Q, lanes / H) environments at a time and the launch ends when the queue
//     is empty, not when the slowest lane of every wave is done;
//   * an episode that has used its stage's step budget is PARKED (state, reward so far, pending action) and continued by the next
//     stage with more lanes: the survivors of a stage -- counted on the device, no host round trip -- get the lane-group size that
//     fills the chip again (2 -> 4 -> ... -> 64 lanes per environment as their number halves), so the stragglers' serial chain runs at
//     the latency form's speed (~R/64 rules per step) instead of holding throughput waves.
// Sums: each lane adds its rules in index order, the H partials are added in butterfly order; every phase of one call runs a
// different H, so an environment's Shepard sums change their rounding (not their value, <= 1e-13) when it is parked; WHICH
// environments are parked depends only on their own trajectory and the static cap: results are deterministic.
#include "sweeps.h"
#include "envs.h"
#include <mutex>
#include <type_traits>
#include <vector>

namespace frirl {

constexpr int RR_BLOCK = 256;

struct RolloutCtl {            // device control block, zeroed before the first phase
    unsigned next[32];         // queue head of each launch
    unsigned count[8];         // environments parked by stage s (the input of stage s + 1)
    unsigned too_big;          // 1: the rule base does not fit the LDS image -> the tiled kernel (shared.hip) runs instead
    unsigned pad[7];
};

struct RolloutPark {           // environments parked by the throughput phase (SoA, capacity Q)
    int32_t *env;
    double *states;            // [Q][NS]
    double *total;
    int32_t *steps;
    int32_t *act;              // index of the pending action
};

struct RolloutPhase {
    int from_parked;           // 0: items are the environment ids 0..Q-1 (fresh episodes); 1: items are the records parked by stage - 1
    int stage;                 // environments parked here are counted in RolloutCtl::count[stage]
    unsigned lo, hi;           // from_parked: run only if lo < count[stage - 1] <= hi
    int cap;                   // steps per environment in this launch before it is parked (>= max_steps: never)
    int qslot;                 // which RolloutCtl::next
    int rps;                   // rules per column of the LDS image
    int ht;                    // slots of the exact-hit hash table (power of two >= 2 rps)
};

// ---- exact hits by lookup ----------------------------------------------------------------------------------------------
// The rule base of a roll-out never changes, so "the lowest rule whose antecedents equal the observation's VE point" (an exact hit:
// distance 0 <=> every antecedent equal, FIVEVagConcl_FRIRL_BestAct.c:89-93) is found through a hash table over the rules' VE
// tuples, built once per workgroup: one probe per action and step instead of a compare + select per action and RULE in the sweep.
constexpr uint32_t HT_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hash_mix(uint32_t hsh, double x)
{
    const double c = x + 0.0;                             // -0.0 and +0.0 are the same antecedent
    const uint32_t lo = (uint32_t)__double2loint(c), hi = (uint32_t)__double2hiint(c);
    hsh = (hsh ^ lo) * 0x9E3779B1u;
    hsh = (hsh ^ hi) * 0x85EBCA6Bu;
    return hsh ^ (hsh >> 15);
}

// The LDS image of the rule base is rule-major: rule r occupies the NANT + 1 consecutive doubles col[r * (NANT + 1) + k] (antecedent VE
// values, then the consequent) -- one address per rule and 16-byte reads with immediate offsets in the sweep.
template <int NANT>
__device__ __forceinline__ bool rule_equals(const double *col, int, int r, const double (&key)[NANT])
{
    bool eq = true;
#pragma unroll
    for (int k = 0; k < NANT; k++) eq = eq && (col[r * (NANT + 1) + k] == key[k]);
    return eq;
}

template <int NANT>
__device__ __forceinline__ void hash_insert(uint32_t *tab, int ht, const double *col, int RPS, int r)
{
    double key[NANT];
    uint32_t hsh = 0u;
#pragma unroll
    for (int k = 0; k < NANT; k++) { key[k] = col[r * (NANT + 1) + k]; hsh = hash_mix(hsh, key[k]); }
    for (int i = 0; i < ht; i++) {
        const uint32_t slot = (hsh + (uint32_t)i) & (uint32_t)(ht - 1);
        const uint32_t cur = atomicCAS(&tab[slot], HT_EMPTY, (uint32_t)r);
        if (cur == HT_EMPTY) return;
        if (rule_equals<NANT>(col, RPS, (int)cur, key)) { atomicMin(&tab[slot], (uint32_t)r); return; }   // duplicates: the lowest index
    }
}

// lowest rule with these antecedents, or FRIRL_HIP_NO_HIT; `hsh` = hash of key[0 .. NANT-2], the last antecedent is mixed in here
template <int NANT>
__device__ __forceinline__ unsigned hash_lookup(const uint32_t *tab, int ht, const double *col, int RPS, uint32_t hsh, const double (&key)[NANT])
{
    hsh = hash_mix(hsh, key[NANT - 1]);
    for (int i = 0; i < ht; i++) {
        const uint32_t cur = tab[(hsh + (uint32_t)i) & (uint32_t)(ht - 1)];
        if (cur == HT_EMPTY) return FRIRL_HIP_NO_HIT;
        if (rule_equals<NANT>(col, RPS, (int)cur, key)) return cur;
    }
    return FRIRL_HIP_NO_HIT;
}

template <int N>
__device__ __forceinline__ void group_sum_n(double (&v)[N], int H)
{
    for (int off = 1; off < H; off <<= 1) {
        double t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(v[i], off, FRIRL_WAVE);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = v[i] + t[i];
    }
}

template <int NANT, int NA, int KIND, int H, bool EXCL>
__global__ __launch_bounds__(RR_BLOCK, 2) void rollout_resident_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                        const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR,
                                                                        const frirl_hip_agent ag, int Q, const frirl_hip_rollout ro,
                                                                        RolloutCtl *__restrict__ ctl, const RolloutPark park, const RolloutPark dst, const RolloutPhase ph)
{
    constexpr int NS = NANT - 1;
    extern __shared__ __attribute__((aligned(16))) double col[];   // [rps][NANT+1] rule base image (rule-major), [2][NANT][U] tables, hash table, slot bytes
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    const int RPS = ph.rps;
    const int R = nrules[0];
    if (R > RPS) {                                        // does not fit: the tiled kernel takes the whole call (uniform exit)
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->too_big = 1u;
        return;
    }
    const unsigned n_in = ph.from_parked ? ctl->count[ph.stage - 1] : (unsigned)Q;
    if (ph.from_parked && !(n_in > ph.lo && n_in <= ph.hi)) return;
    double *tab_s = col + (size_t)(NANT + 1) * RPS;
    constexpr bool LT = KIND != FRIRL_HIP_ENV_CARTPOLE;   // small tables (41-point universes): universes and VE tables in LDS too
    uint32_t *hash_s = reinterpret_cast<uint32_t *>(tab_s + (LT ? 2 * NANT * U : 0));
    uint8_t *slot_s = reinterpret_cast<uint8_t *>(hash_s + ph.ht);
    const int nj = (R + H - 1) / H;                       // rules per lane; the padding rules of the last round weigh exactly 0
    for (int i = threadIdx.x; i < (NANT + 1) * RPS; i += RR_BLOCK) {
        const int k = i / RPS, r = i - k * RPS;          // coalesced reads of the canonical columns, transposed into the rule-major image
        // padding: first antecedent 1e150 away (squared distance ~1e300, its weight underflows to 0), consequent 0
        col[r * (NANT + 1) + k] = (r < R) ? rb[(size_t)k * maxR + r] : (k == 0 ? 1.0e150 : 0.0);
    }
    if (EXCL) for (int r = threadIdx.x; r < RPS; r += RR_BLOCK) slot_s[r] = (r < R) ? ro.rule_slot[r] : (uint8_t)255;
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += RR_BLOCK) grid_s[i] = ag.grid_values[i];
    if (LT) for (int i = threadIdx.x; i < NANT * U; i += RR_BLOCK) { tab_s[i] = ve[i]; tab_s[NANT * U + i] = u[i]; }
    for (int i = threadIdx.x; i < ph.ht; i += RR_BLOCK) hash_s[i] = HT_EMPTY;
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += RR_BLOCK) hash_insert<NANT>(hash_s, ph.ht, col, RPS, r);
    __syncthreads();                                      // the last barrier: from here on every wave runs on its own
    const double *ves, *us;
    if constexpr (LT) { ves = tab_s; us = tab_s + NANT * U; } else { ves = ve; us = u; }
    double udiv[NS];                                      // FIVEInit.c:244-248, once instead of per observation
#pragma unroll
    for (int k = 0; k < NS; k++) udiv[k] = universe_div(us + (size_t)k * U, U);
    auto observe = [&](int k, double x) { const double *uni = us + (size_t)k * U; return ves[(size_t)k * U + snap_index(uni, U, x, udiv[k])]; };

    const int lane = threadIdx.x & (FRIRL_WAVE - 1), h = lane % H;
    const bool leader = (h == 0);
    const auto pk = pin_pow(PowC<NANT>());
    double ave[NA];                                       // uniform loads: the action VE points stay in scalar registers
#pragma unroll
    for (int a = 0; a < NA; a++) ave[a] = ag.action_ve[a];

    double states[NS], q[NS], cur[NS], total = 0.0, action = 0.0;
    int qi = 0, steps = 0, lsteps = 0, success = 0;
    uint32_t mask = 0u;
    bool active = false, fresh = false, drained = false;
#pragma unroll
    for (int k = 0; k < NS; k++) { states[k] = 0.0; q[k] = 0.0; cur[k] = 0.0; }

    for (;;) {
        // ---- refill: every idle group takes the next item of the in-order queue (one atomic per wave) ----------------
        const bool want = !active && !drained;
        const unsigned long long wb = __ballot(want && leader);
        if (wb) {
            const int first = __ffsll((long long)wb) - 1;
            unsigned base = 0u;
            if (lane == first) base = atomicAdd(&ctl->next[ph.qslot], (unsigned)__popcll(wb));
            base = (unsigned)__shfl((int)base, first);
            unsigned item = base + (unsigned)__popcll(wb & ((1ull << lane) - 1ull));
            item = (unsigned)__shfl((int)item, lane - h);
            if (want) {
                if (item < n_in) {
                    active = true;
                    lsteps = 0;
                    success = 0;
                    if (ph.from_parked) {
                        qi = park.env[item];
                        steps = park.steps[item];
                        total = park.total[item];
                        action = grid_s[NS * FRIRL_HIP_MAX_GRID + park.act[item]];
#pragma unroll
                        for (int k = 0; k < NS; k++) states[k] = park.states[(size_t)item * NS + k];
                        fresh = false;
                    } else {
                        qi = (int)item;
                        steps = 0;
                        total = 0.0;
#pragma unroll
                        for (int k = 0; k < NS; k++) states[k] = ro.start_states ? ro.start_states[(size_t)qi * NS + k] : ag.values_def[k];   // frirl_episode.c:46-48
                        fresh = true;
                    }
                    mask = (EXCL && ro.exclude_mask) ? ro.exclude_mask[qi] : 0u;
                } else {
                    drained = true;
                }
            }
        }
        if (!__any(active ? 1 : 0)) break;                // the queue is empty and every episode of this wave has ended

        // ---- the environment's own step (every lane of the group computes the same values) ---------------------------
        if (active) {
            if (fresh) {
#pragma unroll
                for (int k = 0; k < NS; k++) q[k] = observe(k, states[k]);                  // :78 (un-quantised start state)
            } else {
                double r, qs[NS];
                env_do_action(KIND, action, states, cur);                                         // :97
                env_get_reward(KIND, cur, r, success);                                            // :106
                total = total + r;                                                                       // :107
                env_quantize(KIND, NS, grid_s, ag.grid_len, ag.grid_div, cur, qs);                // :112
#pragma unroll
                for (int k = 0; k < NS; k++) q[k] = observe(k, qs[k]);
            }
        }

        // ---- frirl_get_best_action (:148): this lane's rules, all actions --------------------------------------------
        double sv[NA], sw[NA];
        unsigned sh[NA];
#pragma unroll
        for (int a = 0; a < NA; a++) { sv[a] = 0.0; sw[a] = 0.0; sh[a] = FRIRL_HIP_NO_HIT; }
        if (active) {
            // exact hits: one table probe per action; a hit's own sums come out of the sweep poisoned (rsq(0)) and are not read
            double key[NANT];
            uint32_t hs = 0u;
#pragma unroll
            for (int k = 0; k < NS; k++) { key[k] = q[k]; hs = hash_mix(hs, q[k]); }
#pragma unroll
            for (int a = 0; a < NA; a++) {
                key[NS] = ave[a];
                unsigned f = hash_lookup<NANT>(hash_s, ph.ht, col, RPS, hs, key);
                if (EXCL && f != FRIRL_HIP_NO_HIT) {
                    const unsigned sl = slot_s[f];
                    if (sl < 32u && ((mask >> sl) & 1u)) {      // the hit rule is removed: a later duplicate, if any, takes its place
                        unsigned g = FRIRL_HIP_NO_HIT;
                        for (int r = (int)f + 1; r < R && g == FRIRL_HIP_NO_HIT; r++) {
                            const unsigned s2 = slot_s[r];
                            if (!(s2 < 32u && ((mask >> s2) & 1u)) && rule_equals<NANT>(col, RPS, r, key)) g = (unsigned)r;
                        }
                        f = g;
                    }
                }
                sh[a] = f;
            }
            // Two rules in flight.  A rule's LDS reads are issued right after the first instructions of the PREVIOUS rule's arithmetic -- the ones
            // that consume every value that rule read -- so that the compiler's wait for "everything outstanding" (it does not emit partial
            // LDS waits across this loop's back edge) falls where only reads issued a whole rule ago are outstanding.  The scheduling
            // barriers pin that order; without them the two rules' arithmetic is interleaved and both reads are waited for at once
            // (65 536 acrobot roll-outs 9.1 -> 8.3 ms).
            struct Open { double d[NS], va, cq; };
            auto fetch = [&](double (&x)[NANT + 1], int jj) {
                const double *rule = col + (jj * H + h) * (NANT + 1);
#pragma unroll
                for (int k = 0; k <= NANT; k++) x[k] = rule[k];
            };
            auto open = [&](const double (&rule)[NANT + 1]) {              // touches every value of the row
                Open o;
#pragma unroll
                for (int k = 0; k < NS; k++) o.d[k] = q[k] - rule[k];
                o.va = rule[NS]; o.cq = rule[NANT];
                return o;
            };
            auto close = [&](const Open &o, int jj) {
                double s = o.d[0] * o.d[0];
#pragma unroll
                for (int k = 1; k < NS; k++) s = __fma_rn(o.d[k], o.d[k], s);
                if (EXCL) {                               // a removed rule weighs exactly 0 (same sums as the compacted rule base)
                    const unsigned sl = slot_s[jj * H + h];
                    s = (sl < 32u && ((mask >> sl) & 1u)) ? NO_RULE_STATE_PART : s;
                }
#pragma unroll
                for (int a = 0; a < NA; a++) {
                    const double e = ave[a] - o.va;
                    const double d2 = __fma_rn(e, e, s);
                    const double wi = shepard_w(d2, pk);
                    sv[a] = __fma_rn(wi, o.cq, sv[a]);
                    sw[a] = sw[a] + wi;
                }
            };
            if (nj > 0) {
                double ra[NANT + 1], rb2[NANT + 1];
                fetch(ra, 0);
                int j = 0;
                for (; j + 1 < nj; j += 2) {
                    const Open oa = open(ra);
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(rb2, j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    close(oa, j);
                    __builtin_amdgcn_sched_barrier(0);
                    const Open ob = open(rb2);
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(ra, j + 2 < nj ? j + 2 : nj - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    close(ob, j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j < nj) { const Open oa = open(ra); close(oa, j); }
            }
        }
        if (H > 1) {                                      // combine the H rule slices (all lanes of the wave take part in the moves)
            group_sum_n<NA>(sv, H);
            group_sum_n<NA>(sw, H);
        }
        if (active) {
            // first maximum in action order, `bv < c` as max.inl:21; action 0 seeds it (a NaN there sticks, as in the reference)
            double bv = 0.0;
            int pa = 0;
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const double c = (sh[a] != FRIRL_HIP_NO_HIT) ? col[(int)sh[a] * (NANT + 1) + NANT] : sv[a] / sw[a];
                if (a == 0 || bv < c) { bv = c; pa = a; }
            }
            bool ended = false;
            if (fresh) {
                pa = e_greedy(ag, pa, (uint32_t)qi, 0u, 0u);
                fresh = false;
            } else {
                pa = e_greedy(ag, pa, (uint32_t)qi, 0u, (uint32_t)(steps + 1));
#pragma unroll
                for (int k = 0; k < NS; k++) states[k] = cur[k];                                         // :163-165
                steps++;                                                                                 // :174
                lsteps++;
                if (success == 1) ended = true;                                                          // :183
            }
            action = grid_s[NS * FRIRL_HIP_MAX_GRID + pa];                                               // :82 / :151
            if (!ended && steps >= ag.max_steps) ended = true;                                           // :86
            if (ended) {
                active = false;
                if (leader) {
                    ro.steps[qi] = steps;
                    ro.reward[qi] = total;
                    if (ro.success) ro.success[qi] = success;
                    if (ro.final_states)
                        for (int k = 0; k < NS; k++) ro.final_states[(size_t)qi * NS + k] = states[k];
                }
            } else if (lsteps >= ph.cap) {                // a long episode: a later launch finishes it with a whole wave
                active = false;
                if (leader) {
                    const unsigned w = atomicAdd(&ctl->count[ph.stage], 1u);
                    dst.env[w] = qi;
                    dst.steps[w] = steps;
                    dst.total[w] = total;
                    dst.act[w] = pa;
                    for (int k = 0; k < NS; k++) dst.states[(size_t)w * NS + k] = states[k];
                }
            }
        }
    }
}


// ---- the latency form: TWO waves per environment ---------------------------------------------------------------------------
// An episode is a serial chain: step t + 1 needs the action chosen at step t.  With one wave per environment a step is the sum of two
// halves that do not need each other's RESULT until the very end: the greedy sweep over the rules for the current observation
// (hash probes, ~R / 64 rules per lane, butterfly, arg-max) and the environment's own dynamics for the next state (trig, divisions,
// quantiser, universe snap) -- which depend on the action, but there are only A of them.  So the workgroup has two waves: the SWEEPER
// picks the action for the current state while the STEPPER advances the environment for EVERY action in parallel lanes (lane a: action
// a); after one barrier both take the chosen action's outcome from LDS.  A step costs the longer half plus a barrier instead of the
// sum.  Used for the last stage of the staged roll-outs (the stragglers) and for launches of up to a few thousand environments (the
// try-remove replays of the rule-base reduction).  Same arithmetic per value as the one-wave form: identical results.
constexpr int RP_BLOCK = 2 * FRIRL_WAVE;

template <int NANT>
struct PairSpec {              // outcome of one speculative step (per action), double-buffered
    double state[NANT - 1], q[NANT - 1], reward;
    int success, pad;
};

template <int NANT, int NA, int KIND, bool EXCL>
__global__ __launch_bounds__(RP_BLOCK, 4) void rollout_pair_kernel(const double *__restrict__ u, const double *__restrict__ ve, int U,
                                                                   const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR,
                                                                   const frirl_hip_agent ag, int Q, const frirl_hip_rollout ro,
                                                                   RolloutCtl *__restrict__ ctl, const RolloutPark park, const RolloutPhase ph)
{
    constexpr int NS = NANT - 1, H = FRIRL_WAVE;
    extern __shared__ __attribute__((aligned(16))) double col[];   // image, tables, hash table, slot bytes: as rollout_resident_kernel
    __shared__ double grid_s[NANT * FRIRL_HIP_MAX_GRID];
    __shared__ PairSpec<NANT> spec_s[2][NA];
    __shared__ int act_s[2];
    __shared__ unsigned item_s;
    const int RPS = ph.rps;
    const int R = nrules[0];
    if (R > RPS) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->too_big = 1u;
        return;
    }
    const unsigned n_in = ph.from_parked ? ctl->count[ph.stage - 1] : (unsigned)Q;
    if (ph.from_parked && !(n_in > ph.lo && n_in <= ph.hi)) return;
    double *tab_s = col + (size_t)(NANT + 1) * RPS;
    constexpr bool LT = KIND != FRIRL_HIP_ENV_CARTPOLE;
    uint32_t *hash_s = reinterpret_cast<uint32_t *>(tab_s + (LT ? 2 * NANT * U : 0));
    uint8_t *slot_s = reinterpret_cast<uint8_t *>(hash_s + ph.ht);
    const int nj = (R + H - 1) / H;
    for (int i = threadIdx.x; i < (NANT + 1) * RPS; i += RP_BLOCK) {
        const int k = i / RPS, r = i - k * RPS;
        col[r * (NANT + 1) + k] = (r < R) ? rb[(size_t)k * maxR + r] : (k == 0 ? 1.0e150 : 0.0);
    }
    if (EXCL) for (int r = threadIdx.x; r < RPS; r += RP_BLOCK) slot_s[r] = (r < R) ? ro.rule_slot[r] : (uint8_t)255;
    for (int i = threadIdx.x; i < NANT * FRIRL_HIP_MAX_GRID; i += RP_BLOCK) grid_s[i] = ag.grid_values[i];
    if (LT) for (int i = threadIdx.x; i < NANT * U; i += RP_BLOCK) { tab_s[i] = ve[i]; tab_s[NANT * U + i] = u[i]; }
    for (int i = threadIdx.x; i < ph.ht; i += RP_BLOCK) hash_s[i] = HT_EMPTY;
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += RP_BLOCK) hash_insert<NANT>(hash_s, ph.ht, col, RPS, r);
    __syncthreads();
    const double *ves, *us;
    if constexpr (LT) { ves = tab_s; us = tab_s + NANT * U; } else { ves = ve; us = u; }
    double udiv[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) udiv[k] = universe_div(us + (size_t)k * U, U);
    auto observe = [&](int k, double x) { const double *uni = us + (size_t)k * U; return ves[(size_t)k * U + snap_index(uni, U, x, udiv[k])]; };

    const int lane = threadIdx.x & (FRIRL_WAVE - 1);
    const bool sweeper = threadIdx.x < FRIRL_WAVE;        // wave 0 sweeps, wave 1 steps the environment
    const auto pk = pin_pow(PowC<NANT>());
    double ave[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) ave[a] = ag.action_ve[a];

    // next environment of the in-order queue.  The loop has ONE latch and it holds a barrier: thread 0's "store the results, take the
    // next item" block must not become a second back edge -- the compiler then splits the loop in two and parks lane 0 outside the
    // inner one, where the other lanes wait for an item nobody fetches (seen: a hang).
    if (threadIdx.x == 0) item_s = atomicAdd(&ctl->next[ph.qslot], 1u);
    __syncthreads();
    unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)item_s);
    __syncthreads();
    while (item < n_in) {
        double states[NS], q[NS], total = 0.0;
        int qi, steps = 0, success = 0;
        bool fresh;
        if (ph.from_parked) {
            // a parked episode knows its pending action: that step is taken here (both waves, same values), then the loop selects again
            qi = park.env[item];
            steps = park.steps[item];
            total = park.total[item];
            const double action = grid_s[NS * FRIRL_HIP_MAX_GRID + park.act[item]];
            double s0[NS], cur[NS], qs[NS], r;
#pragma unroll
            for (int k = 0; k < NS; k++) s0[k] = park.states[(size_t)item * NS + k];
            env_do_action(KIND, action, s0, cur);
            env_get_reward(KIND, cur, r, success);
            total = total + r;
            env_quantize(KIND, NS, grid_s, ag.grid_len, ag.grid_div, cur, qs);
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = cur[k]; q[k] = observe(k, qs[k]); }
            steps++;
            fresh = false;
        } else {
            qi = (int)item;
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = ro.start_states ? ro.start_states[(size_t)qi * NS + k] : ag.values_def[k]; q[k] = observe(k, states[k]); }
            fresh = true;
        }
        const uint32_t mask = (EXCL && ro.exclude_mask) ? ro.exclude_mask[qi] : 0u;
        // `steps` transitions have been made; the state they led to needs its action unless the episode is over already
        bool ended = (!fresh && success == 1) || steps >= ag.max_steps;
        for (int it = 0; !ended; it++) {
            const int buf = it & 1;
            if (sweeper) {
                // ---- frirl_get_best_action for the current observation (as rollout_resident_kernel with H = 64) ------------------
                double sv[NA], sw[NA];
                unsigned sh[NA];
                double key[NANT];
                uint32_t hs = 0u;
#pragma unroll
                for (int k = 0; k < NS; k++) { key[k] = q[k]; hs = hash_mix(hs, q[k]); }
#pragma unroll
                for (int a = 0; a < NA; a++) {
                    key[NS] = ave[a];
                    unsigned f = hash_lookup<NANT>(hash_s, ph.ht, col, RPS, hs, key);
                    if (EXCL && f != FRIRL_HIP_NO_HIT) {
                        const unsigned sl = slot_s[f];
                        if (sl < 32u && ((mask >> sl) & 1u)) {
                            unsigned g = FRIRL_HIP_NO_HIT;
                            for (int r = (int)f + 1; r < R && g == FRIRL_HIP_NO_HIT; r++) {
                                const unsigned s2 = slot_s[r];
                                if (!(s2 < 32u && ((mask >> s2) & 1u)) && rule_equals<NANT>(col, RPS, r, key)) g = (unsigned)r;
                            }
                            f = g;
                        }
                    }
                    sh[a] = f;
                    sv[a] = 0.0; sw[a] = 0.0;
                }
#pragma unroll 2
                for (int j = 0; j < nj; j++) {
                    const int r = j * H + lane;
                    const double *rule = col + r * (NANT + 1);
                    const double d0 = q[0] - rule[0];
                    double s = d0 * d0;
#pragma unroll
                    for (int k = 1; k < NS; k++) { const double d = q[k] - rule[k]; s = __fma_rn(d, d, s); }
                    const double va = rule[NS], cq = rule[NANT];
                    if (EXCL) {
                        const unsigned sl = slot_s[r];
                        s = (sl < 32u && ((mask >> sl) & 1u)) ? NO_RULE_STATE_PART : s;
                    }
#pragma unroll
                    for (int a = 0; a < NA; a++) {
                        const double e = ave[a] - va;
                        const double d2 = __fma_rn(e, e, s);
                        const double wi = shepard_w(d2, pk);
                        sv[a] = __fma_rn(wi, cq, sv[a]);
                        sw[a] = sw[a] + wi;
                    }
                }
                group_sum_n<NA>(sv, H);
                group_sum_n<NA>(sw, H);
                double bv = 0.0;
                int pa = 0;
#pragma unroll
                for (int a = 0; a < NA; a++) {
                    const double c = (sh[a] != FRIRL_HIP_NO_HIT) ? col[(int)sh[a] * (NANT + 1) + NANT] : sv[a] / sw[a];
                    if (a == 0 || bv < c) { bv = c; pa = a; }
                }
                pa = e_greedy(ag, pa, (uint32_t)qi, 0u, (uint32_t)steps);          // step index of the state the action is chosen for (:78 / :148)
                if (lane == 0) act_s[buf] = pa;
            } else if (lane < NA) {
                // ---- the environment's step for EVERY action (lane a: action a), before anybody knows which one is taken --------
                double cur[NS], qs[NS], r;
                int succ;
                env_do_action(KIND, grid_s[NS * FRIRL_HIP_MAX_GRID + lane], states, cur);                 // :97
                env_get_reward(KIND, cur, r, succ);                                                      // :106
                env_quantize(KIND, NS, grid_s, ag.grid_len, ag.grid_div, cur, qs);                       // :112
                PairSpec<NANT> &o = spec_s[buf][lane];
#pragma unroll
                for (int k = 0; k < NS; k++) { o.state[k] = cur[k]; o.q[k] = observe(k, qs[k]); }
                o.reward = r;
                o.success = succ;
            }
            __syncthreads();
            const PairSpec<NANT> &o = spec_s[buf][act_s[buf]];
#pragma unroll
            for (int k = 0; k < NS; k++) { states[k] = o.state[k]; q[k] = o.q[k]; }                      // :163-165
            total = total + o.reward;                                                                    // :107
            success = o.success;
            steps++;                                                                                     // :174
            fresh = false;
            ended = success == 1 || steps >= ag.max_steps;                                               // :183, :86
        }
        if (threadIdx.x == 0) {
            ro.steps[qi] = steps;
            ro.reward[qi] = total;
            if (ro.success) ro.success[qi] = success;
            if (ro.final_states)
                for (int k = 0; k < NS; k++) ro.final_states[(size_t)qi * NS + k] = states[k];
            item_s = atomicAdd(&ctl->next[ph.qslot], 1u);
        }
        __syncthreads();
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item_s);
        __syncthreads();
    }
}

}  // namespace frirl

using namespace frirl_host;

namespace {

// Workspace of the resident stages, one per stream and kept: hipMallocAsync / hipFreeAsync per call would let the pool hand the block
// freed on one stream to the next call on ANOTHER stream behind an inserted dependency -- which serialises calls that alternate between
// two streams exactly where they could overlap (one call's straggler tail with the next call's bulk).  A few MB per stream, freed when
// the library is unloaded.
struct StreamWs { hipStream_t s; int dev; char *p; size_t bytes; };
std::mutex g_ws_mu;
std::vector<StreamWs> g_ws;
struct WsCleanup { ~WsCleanup() { for (StreamWs &w : g_ws) if (w.p) (void)hipFree(w.p); } } g_ws_cleanup;
char *stream_workspace(hipStream_t s, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (StreamWs &w : g_ws)
        if (w.s == s && w.dev == dev) {
            if (w.bytes >= bytes) return w.p;
            (void)hipStreamSynchronize(s);                 // growing: the old block may still be in use by queued launches of this stream
            (void)hipFree(w.p);
            w.p = nullptr; w.bytes = 0;
            if (hipMalloc(reinterpret_cast<void **>(&w.p), bytes) != hipSuccess) { (void)hipGetLastError(); w.p = nullptr; return nullptr; }
            w.bytes = bytes;
            return w.p;
        }
    StreamWs w{s, dev, nullptr, bytes};
    if (hipMalloc(reinterpret_cast<void **>(&w.p), bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_ws.push_back(w);
    return w.p;
}

int g_cus = 0;
int device_cus()
{
    if (g_cus) return g_cus;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    g_cus = n;
    return n;
}

template <int N, int NA, int KIND, int H, bool EXCL>
void launch_phase(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                  frirl::RolloutCtl *ctl, const frirl::RolloutPark &src, const frirl::RolloutPark &dst, const frirl::RolloutPhase &ph, unsigned items_max, int wps,
                  hipStream_t s)
{
    const size_t dyn = (size_t)(N + 1) * ph.rps * sizeof(double) + (KIND != FRIRL_HIP_ENV_CARTPOLE ? 2 * sizeof(double) * N * (size_t)t->U : 0) + sizeof(uint32_t) * (size_t)ph.ht +
                       (EXCL ? (size_t)ph.rps : 0);
    const long need = ((long)items_max * H + frirl::RR_BLOCK - 1) / frirl::RR_BLOCK;
    const long cap = (long)wps * device_cus();
    const dim3 grid((unsigned)(need < cap ? (need < 1 ? 1 : need) : cap));
    hipLaunchKernelGGL((frirl::rollout_resident_kernel<N, NA, KIND, H, EXCL>), grid, dim3(frirl::RR_BLOCK), dyn, s, t->u, t->ve, t->U, b->rb, b->nrules, b->maxR, *ag, Q, *ro, ctl,
                       src, dst, ph);
}

template <int N, int NA, int KIND, bool EXCL>
void launch_pair(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                 frirl::RolloutCtl *ctl, const frirl::RolloutPark &src, const frirl::RolloutPhase &ph, unsigned items_max, hipStream_t s)
{
    const size_t dyn = (size_t)(N + 1) * ph.rps * sizeof(double) + (KIND != FRIRL_HIP_ENV_CARTPOLE ? 2 * sizeof(double) * N * (size_t)t->U : 0) + sizeof(uint32_t) * (size_t)ph.ht +
                       (EXCL ? (size_t)ph.rps : 0);
    const long per_cu = dyn > 0 ? (long)((150 * 1024) / (dyn + 4096)) : 8;      // workgroups a CU's LDS holds
    const long cap = (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu)) * device_cus();
    const long need = items_max < 1 ? 1 : (long)items_max;
    hipLaunchKernelGGL((frirl::rollout_pair_kernel<N, NA, KIND, EXCL>), dim3((unsigned)(need < cap ? need : cap)), dim3(frirl::RP_BLOCK), dyn, s, t->u, t->ve, t->U, b->rb,
                       b->nrules, b->maxR, *ag, Q, *ro, ctl, src, ph);
}

// the lane-group sizes compiled per shape: 3 actions 2 ... 64 (try-remove masks: 16 and 64 only, those launches are small), 21 actions 4 and 16
template <int N, int NA, int KIND>
void launch_phase_h(int H, bool excl, const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                    frirl::RolloutCtl *ctl, const frirl::RolloutPark &src, const frirl::RolloutPark &dst, const frirl::RolloutPhase &ph, unsigned items_max, int wps,
                    hipStream_t s)
{
#define GO(HH, EX) launch_phase<N, NA, KIND, HH, EX>(t, b, ag, Q, ro, ctl, src, dst, ph, items_max, wps, s)
    if constexpr (NA <= 8) {
        // a whole wave per environment: the two-wave form (sweeper + speculative stepper) -- unless a FRESH launch has more environments
        // than its workgroups hold at once (each keeps its own LDS image): short replays in two rounds lose more than a step gains
        const size_t pdyn = (size_t)(N + 1) * ph.rps * sizeof(double) + (KIND != FRIRL_HIP_ENV_CARTPOLE ? 2 * sizeof(double) * N * (size_t)t->U : 0) + sizeof(uint32_t) * (size_t)ph.ht + (excl ? (size_t)ph.rps : 0);
        const long pcu = (long)((150 * 1024) / (pdyn + 4096));
        const long pcap = (pcu < 1 ? 1 : (pcu > 8 ? 8 : pcu)) * device_cus();
        // (acrobot's own dynamics are the longer half of a step and its never-succeeding episodes a 1000-step chain: there the two-wave form
        // wins even in several rounds -- 2 000 fresh environments 3.2 vs 5.2 ms; mountaincar's trivial step gains nothing: 3.2 vs 2.3 ms)
        const bool long_step = KIND == FRIRL_HIP_ENV_ACROBOT && !excl;
        if (H >= 64 && opts().rollout_pair != 0 && (ph.from_parked || (long)items_max <= pcap || long_step || opts().rollout_pair == 1)) {
            if (excl) launch_pair<N, NA, KIND, true>(t, b, ag, Q, ro, ctl, src, ph, items_max, s);
            else launch_pair<N, NA, KIND, false>(t, b, ag, Q, ro, ctl, src, ph, items_max, s);
            return;
        }
        if (excl) { if (H >= 64) GO(64, true); else GO(16, true); return; }
        switch (H) {
            case 2: GO(2, false); break;
            case 4: GO(4, false); break;
            case 8: GO(8, false); break;
            case 16: GO(16, false); break;
            case 32: GO(32, false); break;
            default: GO(64, false); break;
        }
    } else {
        if (excl) { if (H >= 16) GO(16, true); else GO(4, true); }
        else { if (H >= 16) GO(16, false); else GO(4, false); }
    }
#undef GO
}

}  // namespace

// rules per LDS column the resident form can hold (0: shape not covered -> the tiled kernel)
extern "C" int frirl_hip_rollout_resident_rules(int32_t nant, int32_t A, int32_t p, int32_t env_kind)
{
    const bool shape = (env_kind == FRIRL_HIP_ENV_MOUNTAINCAR && nant == 3 && A == 3) || (env_kind == FRIRL_HIP_ENV_ACROBOT && nant == 5 && A == 3) ||
                       (env_kind == FRIRL_HIP_ENV_CARTPOLE && nant == 5 && A == 21);       // the demos' shapes; anything else: the tiled kernel
    if (!shape || (p > 0 && p != nant)) return 0;
    if (opts().rollout_resident == 0) return 0;
    return (40 * 1024) / ((nant + 1) * 8 + 1 + 8) / 64 * 64;      // image + slot byte + 2 hash slots per rule in <= 40 KiB
}

// Returns 1 when the resident stages were enqueued (*too_big_flag: set on the device when the rule base does not fit the LDS image --
// the caller then launches the tiled kernel behind them, which runs only in that case), 0 when the shape is not covered.
//
// Stages (3 actions, no try-remove masks): every environment starts in stage 0 with as many lanes as keep ALL of them resident at two
// waves per SIMD (2 for 65 536 environments); an episode that has used the stage's step budget is parked, and the next stage gives the
// survivors -- counted on the device, no host round trip -- the lane-group size that fills the chip again (the launch whose range
// holds the count runs, the others return at once).  Budgets: max_steps / 12, / 8, / 6 cumulative, then to the end: on the acrobot
// demo 45 %, 10 % and 1.5 % of the episodes survive them.  Which environments are parked where depends only on their own trajectory.
int frirl_rollout_resident(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, int Q, const frirl_hip_rollout *ro,
                           hipStream_t s, const unsigned **too_big_flag, void **workspace)
{
    *too_big_flag = nullptr;
    *workspace = nullptr;
    const int cap_rules = frirl_hip_rollout_resident_rules(t->nant, ag->A, ag->p, ag->env_kind);
    if (!cap_rules) return 0;
    if (ag->env_kind != FRIRL_HIP_ENV_CARTPOLE && 2 * sizeof(double) * t->nant * (size_t)t->U > 16 * 1024) return 0;   // their tables go to LDS
    const int NS = t->nant - 1;
    const int rps_full = (b->maxR + 63) / 64 * 64;
    const int rps = rps_full < cap_rules ? rps_full : cap_rules;
    const bool excl = ro->exclude_mask && ro->rule_slot;
    // workspace: control block + two park lists (stages alternate between them), stream-ordered
    const size_t q8 = ((size_t)Q + 1) / 2 * 2;
    const size_t list_bytes = q8 * (sizeof(int32_t) * 3 + sizeof(double) * (NS + 1));
    const size_t ctl_bytes = (sizeof(frirl::RolloutCtl) + 63) / 64 * 64;
    char *ws = stream_workspace(s, ctl_bytes + 2 * list_bytes);      // stream-ordered reuse: the previous call on this stream is done with it
    if (!ws) return 0;
    (void)hipMemsetAsync(ws, 0, ctl_bytes, s);
    frirl::RolloutCtl *ctl = reinterpret_cast<frirl::RolloutCtl *>(ws);
    frirl::RolloutPark list[2];
    for (int i = 0; i < 2; i++) {
        char *base = ws + ctl_bytes + (size_t)i * list_bytes;
        list[i].states = reinterpret_cast<double *>(base);
        list[i].total = list[i].states + q8 * NS;
        list[i].env = reinterpret_cast<int32_t *>(list[i].total + q8);
        list[i].steps = list[i].env + q8;
        list[i].act = list[i].steps + q8;
    }
    const Options &o = opts();
    // persistent waves per SIMD: 2 when that keeps every environment resident (65 536 at 2 lanes each: 8.5 / 8.6 / 8.9 ms with 2 / 3 / 4), 4 when the
    // launch is queue-fed anyway (262 144 environments 26.3 -> 21.8 ms, a million 76.5 -> 70.6 ms: more waves hide what the step outside the rule loop waits for)
    const int wps = (o.rollout_wps >= 1 && o.rollout_wps <= 4) ? o.rollout_wps : ((long)Q * 2 > (long)device_cus() * 4 * 2 * FRIRL_WAVE ? 4 : 2);
    const long lanes = (long)device_cus() * 4 * wps * FRIRL_WAVE;
    frirl::RolloutPhase ph = {};
    ph.rps = rps;
    ph.ht = 64;
    while (ph.ht < 2 * rps) ph.ht *= 2;
    int qslot = 0;
#define PHASE(HH, items, SRC, DST)                                                                                                                 \
    do {                                                                                                                                           \
        ph.qslot = qslot++;                                                                                                                        \
        if (t->nant == 3) launch_phase_h<3, 3, FRIRL_HIP_ENV_MOUNTAINCAR>(HH, excl, t, b, ag, Q, ro, ctl, SRC, DST, ph, items, wps, s);             \
        else if (ag->A == 3) launch_phase_h<5, 3, FRIRL_HIP_ENV_ACROBOT>(HH, excl, t, b, ag, Q, ro, ctl, SRC, DST, ph, items, wps, s);             \
        else launch_phase_h<5, 21, FRIRL_HIP_ENV_CARTPOLE>(HH, excl, t, b, ag, Q, ro, ctl, SRC, DST, ph, items, wps, s);                           \
    } while (0)
    const int never = ag->max_steps > 0 ? ag->max_steps : 1;
    if (ag->A > 8 || excl || (long)Q * 64 <= lanes) {
        // 21 actions / try-remove replays / few environments: one lane-group size; many environments of 21 actions park the long episodes once
        int HA = (long)Q * 64 <= lanes ? 64 : ((long)Q * 16 <= lanes * 2 ? 16 : 4);
        if (o.rollout_slices == 4 || o.rollout_slices == 16 || o.rollout_slices == 64) HA = o.rollout_slices;
        if (ag->A > 8 && HA == 64) HA = 16;                 // the butterfly over 64 lanes would outweigh 3 rules per lane
        int cap = never;
        if (HA == 4 && !excl) {
            cap = ag->max_steps / 6 < 64 ? 64 : ag->max_steps / 6;
            if (o.rollout_cap > 0) cap = o.rollout_cap;
            if (cap > never) cap = never;
        }
        ph.from_parked = 0; ph.stage = 0; ph.cap = cap;
        PHASE(HA, (unsigned)Q, list[0], list[0]);
        if (cap < never) {
            ph.from_parked = 1; ph.stage = 1; ph.cap = never;
            ph.lo = 0u; ph.hi = (unsigned)(lanes / 16);
            PHASE(16, (unsigned)((long)Q < lanes / 16 ? Q : lanes / 16), list[0], list[1]);
            ph.lo = (unsigned)(lanes / 16); ph.hi = 0xffffffffu;
            PHASE(4, (unsigned)Q, list[0], list[1]);
        }
    } else {
        // the staged form
        const int ms = ag->max_steps;
        int budget[4] = {ms / 12, ms / 8 - ms / 12, ms / 6 - ms / 8, never};
        if (o.rollout_cap > 0) { budget[0] = o.rollout_cap; budget[1] = budget[2] = o.rollout_cap / 2 > 0 ? o.rollout_cap / 2 : 1; }
        int nstages = 4;
        if (budget[0] < 16 || budget[1] < 8 || budget[2] < 8) { nstages = 1; budget[0] = never; }      // short episodes: nothing to stage
        int H0 = 2;
        while (H0 < 64 && (long)Q * (2 * H0) <= lanes) H0 *= 2;
        if (o.rollout_slices == 2 || o.rollout_slices == 4 || o.rollout_slices == 8 || o.rollout_slices == 16 || o.rollout_slices == 32 || o.rollout_slices == 64) H0 = o.rollout_slices;
        ph.from_parked = 0; ph.stage = 0; ph.cap = nstages > 1 ? budget[0] : never;
        PHASE(H0, (unsigned)Q, list[0], list[0]);
        unsigned in_max = (unsigned)Q;                       // upper bound of a stage's input
        for (int st = 1; st < nstages; st++) {
            ph.from_parked = 1; ph.stage = st; ph.cap = st == nstages - 1 ? never : budget[st];
            const frirl::RolloutPark &src = list[(st - 1) & 1], &dst = list[st & 1];
            for (int H = 64; H >= 2; H /= 2) {               // the launch whose range holds the survivors' number runs
                const long fit = lanes / H;                  // environments that fill the chip at H lanes each
                ph.lo = H == 64 ? 0u : (unsigned)(fit / 2 < 0xffffffffL ? fit / 2 : 0xffffffffL);
                ph.hi = H == 2 ? 0xffffffffu : (unsigned)fit;
                if (ph.lo >= in_max) continue;               // cannot happen: fewer environments than its range starts at
                PHASE(H, (unsigned)((long)in_max < fit || H == 2 ? in_max : fit), src, dst);
            }
        }
    }
#undef PHASE
    *too_big_flag = rps_full > cap_rules ? &ctl->too_big : nullptr;
    *workspace = nullptr;                                     // owned by the per-stream cache: nothing for the caller to free
    return 1;
}
