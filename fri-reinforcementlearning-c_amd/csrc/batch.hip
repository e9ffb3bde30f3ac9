// batch.hip -- library-owned batch of E agents with host descriptors (frirl_hip_batch_*): what a plain-C host
// needs to run many agents on the GPU without touching the HIP runtime itself.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "batch_internal.h"
#include "device_common.h"

using namespace frirl_host;

struct frirl_hip_batch {
    int32_t nant, U, E, maxR;
    int device;                  // every entry point switches the calling thread to it (batches of several devices in one process)
    hipStream_t s;
    double *d_u, *d_ve, *d_rb, *d_rant, *d_grid, *d_ave, *d_states, *d_q_ant, *d_ep_reward, *d_start, *d_prev_reward, *d_prev_rconc, *d_tmp;
    uint16_t *d_uidx;            // 16-bit universe-index mirror of the antecedents (compressed scans, lane-group index store)
    double *d_weights;           // [E][maxR] FIVERB.weights of every agent (rule-base merge; allocated on first use)
    uint8_t *d_active;           // [E] receiver mask of a merge round
    int32_t *d_full;             // [E] append refused during a merge
    double *d_spread_ant;        // [E][nant] / [E]: what determines FIVERB.weights after learning (frirl_hip_envs.spread_*)
    int32_t *d_spread_R;
    void *d_lanes_ws;            // transposed rule bases of the lane-group kernel (allocated on first use)
    size_t lanes_ws_bytes;
    void *d_learn_ws;            // workspace of the persistent construct loop, frirl_hip_learn_train (allocated on first use)
    size_t learn_ws_bytes;
    int64_t *d_steps_total;      // [E] environment steps of the last frirl_hip_learn_train
    int32_t *d_nrules, *d_fus, *d_done, *d_ep_steps, *d_status, *d_episode, *d_prev_nrules, *d_prev_steps, *d_converged, *d_episodes, *d_epended;
    frirl_hip_tables t;
    frirl_hip_rulebases rb;
    frirl_hip_agent agent;
    frirl_hip_envs envs;
    frirl_hip_convergence conv;
    int64_t total_env_steps;
    std::vector<int32_t> h_i;
    std::vector<double> h_d;
};

#define BCHK(call, what)                                                                                      \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)

template <typename T>
static bool dalloc(T **p, size_t n)
{
    return hipMalloc((void **)p, n * sizeof(T)) == hipSuccess && hipMemset(*p, 0, n * sizeof(T)) == hipSuccess;
}

namespace frirl {
// done[e] |= converged[e]: converged agents sit later episodes out
__global__ void mask_converged_kernel(int32_t *__restrict__ done, const int32_t *__restrict__ converged, int E)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E && converged[e]) done[e] = 1;
}
// end of an exchange round: every agent but the master runs again in the next round whether or not its rule base was complete
// (frirl_sequential_run does not look at is_running on entry, frirl_agent.c:321-328), and `epended` starts every chunk at 0 (:37)
__global__ void next_round_kernel(int32_t *__restrict__ converged, int32_t *__restrict__ epended, int E, int first)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    epended[e] = 0;
    if (e >= first) converged[e] = 0;
}
// number of agents whose episode is still running -> *out
__global__ void count_running_kernel(const int32_t *__restrict__ done, int E, int32_t *__restrict__ out)
{
    int n = 0;
    for (int e = threadIdx.x; e < E; e += blockDim.x) n += done[e] ? 0 : 1;
    __shared__ int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    if (n) atomicAdd(&s, n);
    __syncthreads();
    if (threadIdx.x == 0) *out = s;
}
}  // namespace frirl

extern "C" void frirl_hip_batch_destroy(frirl_hip_batch *b)
{
    if (!b) return;
    DeviceGuard keep_device_;
    (void)hipSetDevice(b->device);
    if (b->s) (void)hipStreamSynchronize(b->s);
    void *ptrs[] = {b->d_u, b->d_ve, b->d_rb, b->d_rant, b->d_grid, b->d_ave, b->d_states, b->d_q_ant, b->d_ep_reward, b->d_start, b->d_prev_reward,
                    b->d_prev_rconc, b->d_tmp, b->d_nrules, b->d_fus, b->d_done, b->d_ep_steps, b->d_status, b->d_episode, b->d_prev_nrules,
                    b->d_prev_steps, b->d_converged, b->d_episodes, b->d_epended, b->d_uidx, b->d_lanes_ws, b->d_learn_ws, b->d_steps_total, b->d_weights, b->d_active, b->d_full, b->d_spread_ant, b->d_spread_R};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (b->s) (void)hipStreamDestroy(b->s);
    delete b;
}

extern "C" frirl_hip_batch *frirl_hip_batch_create(const frirl_hip_batch_desc *d)
{
    if (!d || !d->u || !d->ve || !d->agent.grid_values || !d->agent.action_ve || !d->rant0 || !d->rconc0) { set_error("frirl_hip_batch_create: NULL pointer"); return nullptr; }
    if (d->nant < 2 || d->nant > 9 || d->U < 2 || d->E < 1 || d->maxR < 2 || d->R0 < 1 || d->R0 > d->maxR || d->agent.A < 1 || d->agent.A > FRIRL_HIP_MAX_ACTIONS) { set_error("frirl_hip_batch_create: bad sizes"); return nullptr; }
    DeviceGuard keep_device_;
    if (d->device_select == 1 && hipSetDevice(d->device) != hipSuccess) { set_error("frirl_hip_batch_create: hipSetDevice(%d) failed", d->device); (void)hipGetLastError(); return nullptr; }
    if (check_device()) return nullptr;
    frirl_hip_batch *b = new frirl_hip_batch();
    if (hipGetDevice(&b->device) != hipSuccess) b->device = 0;
    b->nant = d->nant; b->U = d->U; b->E = d->E; b->maxR = d->maxR + (d->maxR & 1);
    const size_t E = (size_t)b->E, n = (size_t)b->nant, M = (size_t)b->maxR, ns = n - 1;
    bool ok = hipStreamCreateWithFlags(&b->s, hipStreamNonBlocking) == hipSuccess;
    ok = ok && dalloc(&b->d_u, n * d->U) && dalloc(&b->d_ve, n * d->U) && dalloc(&b->d_rb, E * (n + 1) * M) && dalloc(&b->d_rant, E * n * M);
    ok = ok && dalloc(&b->d_grid, (size_t)FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID) && dalloc(&b->d_ave, (size_t)FRIRL_HIP_MAX_ACTIONS);
    ok = ok && dalloc(&b->d_states, E * ns) && dalloc(&b->d_q_ant, E * n) && dalloc(&b->d_ep_reward, E) && dalloc(&b->d_prev_reward, E);
    ok = ok && dalloc(&b->d_prev_rconc, E * M) && dalloc(&b->d_tmp, E * (n + 1)) && dalloc(&b->d_nrules, E) && dalloc(&b->d_fus, E) && dalloc(&b->d_done, E);
    ok = ok && dalloc(&b->d_ep_steps, E) && dalloc(&b->d_status, E) && dalloc(&b->d_episode, E) && dalloc(&b->d_prev_nrules, E) && dalloc(&b->d_prev_steps, E);
    ok = ok && dalloc(&b->d_converged, E) && dalloc(&b->d_episodes, E + 1) && dalloc(&b->d_epended, E) && dalloc(&b->d_spread_ant, E * n) && dalloc(&b->d_spread_R, E);
    if (ok && d->U <= 65536) ok = dalloc(&b->d_uidx, E * n * M);
    if (ok && d->start_states) ok = dalloc(&b->d_start, E * ns) && hipMemcpy(b->d_start, d->start_states, sizeof(double) * E * ns, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(b->d_u, d->u, sizeof(double) * n * d->U, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(b->d_ve, d->ve, sizeof(double) * n * d->U, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(b->d_grid, d->agent.grid_values, sizeof(double) * n * FRIRL_HIP_MAX_GRID, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(b->d_ave, d->agent.action_ve, sizeof(double) * d->agent.A, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { set_error("frirl_hip_batch_create: HIP allocation/copy failed: %s", hipGetErrorString(hipGetLastError())); frirl_hip_batch_destroy(b); return nullptr; }
    b->t.nant = b->nant; b->t.U = b->U; b->t.u = b->d_u; b->t.ve = b->d_ve;
    b->rb.E = b->E; b->rb.maxR = b->maxR; b->rb.rb = b->d_rb; b->rb.nrules = b->d_nrules; b->rb.uidx = b->d_uidx;
    b->agent = d->agent;
    b->agent.grid_values = b->d_grid;
    b->agent.action_ve = b->d_ave;
    memset(&b->envs, 0, sizeof b->envs);
    b->envs.states = b->d_states; b->envs.q_ant = b->d_q_ant; b->envs.fus = b->d_fus; b->envs.done = b->d_done; b->envs.ep_steps = b->d_ep_steps;
    b->envs.ep_reward = b->d_ep_reward; b->envs.rant = b->d_rant; b->envs.status = b->d_status; b->envs.start_states = b->d_start; b->envs.episode = b->d_episode;
    b->envs.spread_ant = b->d_spread_ant; b->envs.spread_R = b->d_spread_R;
    b->conv.prev_nrules = b->d_prev_nrules; b->conv.prev_steps = b->d_prev_steps; b->conv.prev_reward = b->d_prev_reward; b->conv.prev_rconc = b->d_prev_rconc;
    b->conv.converged = b->d_converged; b->conv.episodes = b->d_episodes; b->conv.epended = b->d_epended;
    b->total_env_steps = 0;
    // initial rules through FIVE_add_rule, as FIVEInit does (every agent gets the same R0 rules)
    std::vector<double> stage(E * (n + 1));
    for (int r = 0; r < d->R0; r++) {
        for (size_t e = 0; e < E; e++) {
            for (size_t k = 0; k < n; k++) stage[e * n + k] = d->rant0[(size_t)r * n + k];
            stage[E * n + e] = d->rconc0[r];
        }
        if (hipMemcpyAsync(b->d_tmp, stage.data(), sizeof(double) * E * (n + 1), hipMemcpyHostToDevice, b->s) != hipSuccess ||
            five_hip_add_rule(&b->t, &b->rb, b->d_tmp, b->d_tmp + E * n, nullptr, b->d_rant, nullptr, b->s) != 0 ||
            hipStreamSynchronize(b->s) != hipSuccess) {
            frirl_hip_batch_destroy(b);
            return nullptr;
        }
    }
    if (frirl_hip_convergence_init(&b->rb, b->nant, &b->conv, b->s) != 0 || hipStreamSynchronize(b->s) != hipSuccess) { frirl_hip_batch_destroy(b); return nullptr; }
    return b;
}

extern "C" int frirl_hip_batch_episode(frirl_hip_batch *b)
{
    if (!b) { set_error("frirl_hip_batch_episode: NULL batch"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    int rc = frirl_hip_episode_begin(&b->t, &b->rb, &b->agent, &b->envs, b->s);
    if (rc) return rc;
    hipLaunchKernelGGL(frirl::mask_converged_kernel, dim3((b->E + 255) / 256), dim3(256), 0, b->s, b->d_done, b->d_converged, b->E);
    const int chunk = 64;
    int32_t running = 1;
    // many agents / small rule bases: lane-group kernel, the whole episode in one launch (frirl_hip_episode_run_lanes)
    if (frirl_hip_lanes_preferred(b->nant, b->E, b->agent.A)) {
        if (!b->d_lanes_ws) {
            b->lanes_ws_bytes = frirl_hip_lanes_workspace_bytes(b->nant, b->E, b->maxR, b->agent.A);
            BCHK(hipMalloc(&b->d_lanes_ws, b->lanes_ws_bytes), "lane-group workspace");
        }
        if ((rc = frirl_hip_episode_run_lanes(&b->t, &b->rb, &b->agent, &b->envs, b->agent.max_steps, b->d_lanes_ws, b->lanes_ws_bytes, b->s))) return rc;
        running = 0;
    } else
    // small rule bases: whole episode in one launch out of LDS (frirl_hip_episode_run); environments that outgrow the
    // LDS slab come back not-done and finish through the step kernel below
    if (b->agent.A <= 8 && 2 * sizeof(double) * b->nant * (size_t)b->U <= 16 * 1024) {
        b->h_i.resize(b->E);
        BCHK(hipMemcpyAsync(b->h_i.data(), b->d_nrules, sizeof(int32_t) * b->E, hipMemcpyDeviceToHost, b->s), "nrules download");
        BCHK(hipStreamSynchronize(b->s), "nrules sync");
        int need = 0;
        for (int e = 0; e < b->E; e++) if (b->h_i[e] > need) need = b->h_i[e];
        need += 128;
        if (need <= 256) {      // measured: the LDS-resident form pays only while the slab leaves >= ~12 waves per CU
            if ((rc = frirl_hip_episode_run(&b->t, &b->rb, &b->agent, &b->envs, b->agent.max_steps, need <= 256 ? 256 : (need <= 512 ? 512 : 1024), b->s))) return rc;
            hipLaunchKernelGGL(frirl::count_running_kernel, dim3(1), dim3(256), 0, b->s, b->d_done, b->E, b->d_episodes + b->E);
            BCHK(hipMemcpyAsync(&running, b->d_episodes + b->E, sizeof(int32_t), hipMemcpyDeviceToHost, b->s), "running download");
            BCHK(hipStreamSynchronize(b->s), "episode sync");
        }
    }
    for (int done_steps = 0; done_steps < b->agent.max_steps && running > 0; done_steps += chunk) {
        const int nst = (b->agent.max_steps - done_steps < chunk) ? b->agent.max_steps - done_steps : chunk;
        if ((rc = frirl_hip_episode_steps(&b->t, &b->rb, &b->agent, &b->envs, nst, b->s))) return rc;
        hipLaunchKernelGGL(frirl::count_running_kernel, dim3(1), dim3(256), 0, b->s, b->d_done, b->E, b->d_episodes + b->E);
        BCHK(hipMemcpyAsync(&running, b->d_episodes + b->E, sizeof(int32_t), hipMemcpyDeviceToHost, b->s), "running download");
        BCHK(hipStreamSynchronize(b->s), "episode sync");
    }
    // account the steps of the agents that took part in this episode (converged ones keep ep_steps = 0)
    b->h_i.resize(b->E);
    BCHK(hipMemcpyAsync(b->h_i.data(), b->d_ep_steps, sizeof(int32_t) * b->E, hipMemcpyDeviceToHost, b->s), "steps download");
    BCHK(hipStreamSynchronize(b->s), "steps sync");
    for (int e = 0; e < b->E; e++) b->total_env_steps += b->h_i[e];
    if ((rc = frirl_hip_convergence_update(&b->rb, b->nant, &b->agent, &b->envs, &b->conv, b->s))) return rc;
    BCHK(hipStreamSynchronize(b->s), "convergence sync");
    return check_launch("frirl_hip_batch_episode");
}

// The whole construct loop on the device (frirl_hip_learn_train) where the persistent learner covers the shape: every agent runs its
// episodes at its own pace instead of the batch waiting for its longest episode, launch after launch.  Same agents, same results
// (both forms follow the reference's loop per agent); *episodes_run = the largest number of episodes an agent ran in this call.
static int batch_train_persistent(frirl_hip_batch *b, int32_t max_episodes, int32_t *episodes_run)
{
    const size_t E = (size_t)b->E;
    b->h_i.resize(2 * E);
    BCHK(hipMemcpy(b->h_i.data(), b->d_episodes, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "episodes download");
    BCHK(hipMemcpy(b->h_i.data() + E, b->d_converged, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "converged download");
    int before = 0;
    bool all = true;
    for (size_t e = 0; e < E; e++) if (!b->h_i[E + e]) { all = false; before = std::max(before, b->h_i[e]); }
    if (episodes_run) *episodes_run = 0;
    if (all || max_episodes < 2) return FRIRL_HIP_OK;
    if (!b->d_learn_ws) {
        b->learn_ws_bytes = frirl_hip_learn_train_workspace_bytes(b->nant, b->E, b->maxR, b->agent.A);
        BCHK(hipMalloc(&b->d_learn_ws, b->learn_ws_bytes), "learner workspace");
        BCHK(hipMalloc((void **)&b->d_steps_total, sizeof(int64_t) * E), "learner step counters");
    }
    BCHK(hipMemsetAsync(b->d_steps_total, 0, sizeof(int64_t) * E, b->s), "step counters");
    BCHK(hipMemsetD32Async((hipDeviceptr_t)b->d_done, 1, E, b->s), "episode flags");       // every agent is between two episodes here
    // agents that are still learning have all run `before` episodes (the per-episode form keeps them in step): up to max_episodes - 1 more
    int32_t launches = 0;
    int rc = frirl_hip_learn_train(&b->t, &b->rb, &b->agent, &b->envs, &b->conv, 512, max_episodes + before, nullptr, b->d_steps_total, nullptr,
                                   b->d_learn_ws, b->learn_ws_bytes, &launches, nullptr, nullptr, b->s);
    if (rc) return rc;
    std::vector<int64_t> st(E);
    BCHK(hipMemcpy(st.data(), b->d_steps_total, sizeof(int64_t) * E, hipMemcpyDeviceToHost), "steps download");
    BCHK(hipMemcpy(b->h_i.data(), b->d_episodes, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "episodes download");
    int after = before;
    for (size_t e = 0; e < E; e++) { b->total_env_steps += st[e]; after = std::max(after, b->h_i[e]); }
    if (episodes_run) *episodes_run = after - before;
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_batch_train(frirl_hip_batch *b, int32_t max_episodes, int32_t *episodes_run)
{
    if (!b) { set_error("frirl_hip_batch_train: NULL batch"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    if (!b->agent.evaluate && b->d_uidx && frirl_hip_learn_supported(b->nant, b->U, b->agent.A, b->agent.p, b->agent.env_kind))
        return batch_train_persistent(b, max_episodes, episodes_run);
    int ep = 0;
    for (ep = 1; ep < max_episodes; ep++) {             // at most max_episodes-1 episodes (frirl_sequential_run.c:51,59)
        int rc = frirl_hip_batch_episode(b);
        if (rc) return rc;
        b->h_i.resize(b->E);
        BCHK(hipMemcpy(b->h_i.data(), b->d_converged, sizeof(int32_t) * b->E, hipMemcpyDeviceToHost), "converged download");
        bool all = true;
        for (int e = 0; e < b->E && all; e++) all = b->h_i[e] != 0;
        if (all) { ep++; break; }
    }
    if (episodes_run) *episodes_run = ep - 1;
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_batch_stats(frirl_hip_batch *b, frirl_hip_batch_stats_t *out) { return frirl_host::batch_stats_first(b, out, nullptr); }

// the same + "agent 0's rule base is complete" from the same download (the multi-device report's master flag)
int frirl_host::batch_stats_first(frirl_hip_batch *b, frirl_hip_batch_stats_t *out, int32_t *first_converged)
{
    if (!b || !out) { set_error("frirl_hip_batch_stats: NULL"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const int E = b->E;
    std::vector<double> rew(E);
    std::vector<int32_t> steps(E), nr(E), cv(E), eps(E);
    BCHK(hipMemcpy(rew.data(), b->d_prev_reward, sizeof(double) * E, hipMemcpyDeviceToHost), "stats download");   // last finished episode
    BCHK(hipMemcpy(steps.data(), b->d_prev_steps, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "stats download");
    BCHK(hipMemcpy(nr.data(), b->d_nrules, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "stats download");
    BCHK(hipMemcpy(cv.data(), b->d_converged, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "stats download");
    BCHK(hipMemcpy(eps.data(), b->d_episodes, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "stats download");
    memset(out, 0, sizeof *out);
    out->agents = E;
    out->reward_min = rew[0]; out->reward_max = rew[0];
    for (int e = 0; e < E; e++) {
        out->reward_sum += rew[e]; out->steps_sum += steps[e]; out->rules_sum += nr[e];
        if (rew[e] < out->reward_min) out->reward_min = rew[e];
        if (rew[e] > out->reward_max) out->reward_max = rew[e];
        out->converged += cv[e] != 0;
        out->full_agents += nr[e] >= b->maxR;
        if (eps[e] > out->episodes_max) out->episodes_max = eps[e];
    }
    out->total_env_steps = b->total_env_steps;
    if (first_converged) *first_converged = cv[0];
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_batch_get_rulebase(frirl_hip_batch *b, int32_t e, int32_t *R, double *rant, double *rconc)
{
    if (!b || !R || e < 0 || e >= b->E) { set_error("frirl_hip_batch_get_rulebase: bad arguments"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    int32_t r = 0;
    BCHK(hipMemcpy(&r, b->d_nrules + e, sizeof(int32_t), hipMemcpyDeviceToHost), "nrules download");
    *R = r;
    if (!rant || !rconc || r == 0) return FRIRL_HIP_OK;
    const size_t n = b->nant, M = b->maxR;
    std::vector<double> col(r);
    for (size_t k = 0; k < n; k++) {
        BCHK(hipMemcpy(col.data(), b->d_rant + ((size_t)e * n + k) * M, sizeof(double) * r, hipMemcpyDeviceToHost), "rant download");
        for (int i = 0; i < r; i++) rant[(size_t)i * n + k] = col[i];
    }
    BCHK(hipMemcpy(rconc, b->d_rb + ((size_t)e * (n + 1) + n) * M, sizeof(double) * r, hipMemcpyDeviceToHost), "rconc download");
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_batch_reduce(frirl_hip_batch *b, int32_t e, int strategy, double reward_tolerance, int depth, frirl_hip_reduce_result *result)
{
    if (!b || !result || e < 0 || e >= b->E) { set_error("frirl_hip_batch_reduce: bad arguments"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t n = b->nant, M = b->maxR;
    frirl_hip_rulebases one = b->rb;                                 // agent e's slab as a rule-base batch of one
    one.E = 1;
    one.rb = b->d_rb + (size_t)e * (n + 1) * M;
    one.nrules = b->d_nrules + e;
    if (one.uidx) one.uidx = b->rb.uidx + (size_t)e * n * M;
    return frirl_hip_reduce_shared(&b->t, &one, &b->agent, b->d_rant + (size_t)e * n * M, strategy, reward_tolerance, depth, nullptr, result, b->s);
}

// ---- multi-agent rule-base merge: one round of the reference's many-agent loop (frirl_agent.c:426-462) -------------------
// (1) every agent id >= 1 takes over the master's (agent 0's) rules -- all receivers in ONE launch; (2) the master takes over the
// rules of agent 1, 2, ... one after the other (sequential by definition: each merge changes the master).  Agents whose rule base
// passed the cheap completeness test in this chunk (`epended`) do not send, as in the reference (frirl_agent.c:338,352).
// ---- the pieces of a merge round (batch_internal.h; also strung together across devices by multi.hip) --------------------------
namespace frirl_host {

BatchView batch_view(frirl_hip_batch *b)
{
    return BatchView{b->nant, b->maxR, b->E, b->device, b->s, b->d_rant, b->d_rb, b->d_nrules, b->d_converged, b->d_epended};
}

int batch_merge_prepare(frirl_hip_batch *b, std::vector<int32_t> &pended)
{
    std::vector<int32_t> &conv = pended;
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t M = b->maxR, E = b->E;
    if (!b->d_weights) {
        if (!dalloc(&b->d_weights, E * M) || !dalloc(&b->d_active, E) || !dalloc(&b->d_full, E)) { set_error("frirl_hip_batch_merge_round: allocation failed"); return FRIRL_HIP_ELAUNCH; }
    }
    conv.resize(E);
    BCHK(hipMemcpyAsync(conv.data(), b->d_epended, sizeof(int32_t) * E, hipMemcpyDeviceToHost, b->s), "epended download");
    BCHK(hipStreamSynchronize(b->s), "merge sync");
    // the receivers' FIVERB.weights as the learning episodes left them (the reference's merge starts from that array)
    return frirl_hip_weights_from_spread(&b->t, &b->rb, b->agent.p, &b->envs, b->d_weights, b->s);
}

int batch_merge_into_agents(frirl_hip_batch *b, const frirl_hip_sender *snd, bool skip_first)
{
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t E = b->E;
    if (skip_first && E < 2) return FRIRL_HIP_OK;
    std::vector<uint8_t> act(E, 1);
    if (skip_first) act[0] = 0;
    BCHK(hipMemcpyAsync(b->d_active, act.data(), E, hipMemcpyHostToDevice, b->s), "active upload");
    const int rc = frirl_hip_merge_rb(&b->t, &b->rb, &b->agent, b->d_rant, snd, b->d_weights, b->d_active, b->d_full, b->s);
    if (rc) return rc;
    BCHK(hipStreamSynchronize(b->s), "merge sync");              // `act` must outlive the upload
    return FRIRL_HIP_OK;
}

int batch_merge_into_first(frirl_hip_batch *b, const frirl_hip_sender *snd)
{
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    frirl_hip_rulebases master = b->rb;
    master.E = 1;
    return frirl_hip_merge_rb(&b->t, &master, &b->agent, b->d_rant, snd, b->d_weights, nullptr, b->d_full, b->s);
}

frirl_hip_sender batch_sender(frirl_hip_batch *b, int id)
{
    const size_t n = b->nant, M = b->maxR;
    frirl_hip_sender snd;
    memset(&snd, 0, sizeof snd);
    snd.rule_stride = 1; snd.dim_stride = (int64_t)M;
    snd.rant = b->d_rant + (size_t)id * n * M;
    snd.rconc = b->d_rb + ((size_t)id * (n + 1) + n) * M;
    snd.S_dev = b->d_nrules + id;
    return snd;
}

int batch_merge_finish(frirl_hip_batch *b, int32_t *full_agents, bool first_is_master)
{
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t E = b->E;
    b->h_i.resize(E);
    BCHK(hipMemcpyAsync(b->h_i.data(), b->d_nrules, sizeof(int32_t) * E, hipMemcpyDeviceToHost, b->s), "nrules download");
    BCHK(hipStreamSynchronize(b->s), "merge sync");
    if (full_agents) for (size_t e = 0; e < E; e++) *full_agents += b->h_i[e] >= b->maxR;
    hipLaunchKernelGGL(frirl::next_round_kernel, dim3((b->E + 255) / 256), dim3(256), 0, b->s, b->d_converged, b->d_epended, b->E, first_is_master ? 1 : 0);
    // rule count and consequents of the next convergence test are retaken from the merged rule bases (frirl_sequential_run.c:66-72)
    const int rc = frirl_hip_convergence_refresh(&b->rb, b->nant, &b->conv, b->s);
    if (rc) return rc;
    BCHK(hipStreamSynchronize(b->s), "merge sync");
    return FRIRL_HIP_OK;
}

}  // namespace frirl_host

extern "C" int frirl_hip_batch_merge_round(frirl_hip_batch *b, int32_t *full_agents)
{
    if (!b) { set_error("frirl_hip_batch_merge_round: NULL batch"); return FRIRL_HIP_EINVAL; }
    if (full_agents) *full_agents = 0;
    if (b->E < 2) return FRIRL_HIP_OK;
    std::vector<int32_t> conv;
    int rc = batch_merge_prepare(b, conv);
    if (rc) return rc;
    if (!conv[0]) {                                                   // (1) master -> every other agent
        const frirl_hip_sender snd = batch_sender(b, 0);
        if ((rc = batch_merge_into_agents(b, &snd, true))) return rc;
    }
    for (int id = 1; id < b->E; id++) {                               // (2) agent id -> master, id ascending
        if (conv[id]) continue;
        const frirl_hip_sender snd = batch_sender(b, id);
        if ((rc = batch_merge_into_first(b, &snd))) return rc;
    }
    return batch_merge_finish(b, full_agents, true);
}

// frirl_omp_run's loop (frirl_agent.c:319-360) for the agents of this batch: every round each agent runs a chunk of at most
// `chunk - 1` episodes (FRIRL_AGENT_EPCHUNK = 10 in the reference's config.h.in:68; an agent whose rule base is found complete stops
// for the rest of ITS chunk and runs again in the next round), then one exchange round gated by that chunk's `epended` flags -- until
// the master's rule base is complete or the master has started max_episodes episodes, which frirl_sequential_run only looks at when a
// chunk is over (:57-63): the master always finishes its chunk.  *episodes_run = the master's episodes.
extern "C" int frirl_hip_batch_train_merged(frirl_hip_batch *b, int32_t max_episodes, int32_t chunk, int32_t *episodes_run, int32_t *rounds)
{
    if (!b || chunk < 2) { set_error("frirl_hip_batch_train_merged: bad arguments"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    int episode_num = 1, episodes = 0, nrounds = 0;         // frirl_desc.episode_num of the master starts at 1 (frirl_init.c)
    for (;;) {
        int32_t master_done = 0;
        for (int c = 1; c < chunk; c++) {
            const int rc = frirl_hip_batch_episode(b);
            if (rc) return rc;
            episodes++;
            BCHK(hipMemcpy(&master_done, b->d_converged, sizeof(int32_t), hipMemcpyDeviceToHost), "converged download");
            if (master_done) break;
            episode_num++;
        }
        if (master_done || !(episode_num < max_episodes)) break;
        const int rc = frirl_hip_batch_merge_round(b, nullptr);
        if (rc) return rc;
        nrounds++;
    }
    if (episodes_run) *episodes_run = episodes;
    if (rounds) *rounds = nrounds;
    return FRIRL_HIP_OK;
}

// ---- batched rule-base I/O in the reference's .frirlrb.bin record format (frirl_utils.c:151-281) ------------------------
extern "C" int frirl_hip_batch_save_rulebases(frirl_hip_batch *b, const char *path)
{
    if (!b || !path) { set_error("frirl_hip_batch_save_rulebases: bad arguments"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t n = b->nant, M = b->maxR, E = b->E;
    std::vector<int32_t> nr(E);
    std::vector<double> rant(E * n * M), rb(E * (n + 1) * M);
    BCHK(hipStreamSynchronize(b->s), "save sync");
    BCHK(hipMemcpy(nr.data(), b->d_nrules, sizeof(int32_t) * E, hipMemcpyDeviceToHost), "nrules download");
    BCHK(hipMemcpy(rant.data(), b->d_rant, sizeof(double) * rant.size(), hipMemcpyDeviceToHost), "rant download");
    BCHK(hipMemcpy(rb.data(), b->d_rb, sizeof(double) * rb.size(), hipMemcpyDeviceToHost), "rb download");
    FILE *fp = fopen(path, "wb");
    if (!fp) { set_error("frirl_hip_batch_save_rulebases: cannot open %s", path); return FRIRL_HIP_EINVAL; }
    std::vector<double> rec;
    bool ok = true;
    for (size_t e = 0; e < E && ok; e++) {
        const int32_t R = nr[e];
        rec.resize((size_t)R * (n + 1));
        for (int r = 0; r < R; r++) {
            for (size_t k = 0; k < n; k++) rec[(size_t)r * (n + 1) + k] = rant[(e * n + k) * M + r];
            rec[(size_t)r * (n + 1) + n] = rb[(e * (n + 1) + n) * M + r];
        }
        ok = fwrite(&R, sizeof R, 1, fp) == 1 && (R == 0 || fwrite(rec.data(), sizeof(double), rec.size(), fp) == rec.size());
    }
    ok = (fclose(fp) == 0) && ok;
    if (!ok) { set_error("frirl_hip_batch_save_rulebases: write to %s failed", path); return FRIRL_HIP_EINVAL; }
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_batch_load_rulebases(frirl_hip_batch *b, const char *path, int32_t *records_read)
{
    if (!b || !path) { set_error("frirl_hip_batch_load_rulebases: bad arguments"); return FRIRL_HIP_EINVAL; }
    DeviceGuard keep_device_; BCHK(hipSetDevice(b->device), "hipSetDevice");
    const size_t n = b->nant, M = b->maxR, E = b->E;
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error("frirl_hip_batch_load_rulebases: cannot open %s", path); return FRIRL_HIP_EINVAL; }
    // parse and validate the whole file before touching the batch
    std::vector<std::vector<double>> recs;
    for (;;) {
        int32_t R = 0;
        const size_t got = fread(&R, 1, sizeof R, fp);
        if (got == 0) break;                                   // clean end of file
        if (got != sizeof R) { fclose(fp); set_error("frirl_hip_batch_load_rulebases: %s: truncated record header", path); return FRIRL_HIP_EINVAL; }
        if (R < 1 || (size_t)R > M) { fclose(fp); set_error("frirl_hip_batch_load_rulebases: %s: record %zu has %d rules (capacity %zu)", path, recs.size(), R, M); return FRIRL_HIP_EINVAL; }
        std::vector<double> rec((size_t)R * (n + 1));
        if (fread(rec.data(), sizeof(double), rec.size(), fp) != rec.size()) { fclose(fp); set_error("frirl_hip_batch_load_rulebases: %s: record %zu is truncated", path, recs.size()); return FRIRL_HIP_EINVAL; }
        for (double v : rec) if (!(v - v == 0.0)) { fclose(fp); set_error("frirl_hip_batch_load_rulebases: %s: record %zu holds a non-finite value", path, recs.size()); return FRIRL_HIP_EINVAL; }
        recs.push_back(std::move(rec));
        if (recs.size() == E) break;
    }
    fclose(fp);
    if (recs.empty()) { set_error("frirl_hip_batch_load_rulebases: %s holds no rule base", path); return FRIRL_HIP_EINVAL; }
    if (records_read) *records_read = (int32_t)recs.size();
    // rebuild: nrules = 0, then rule r of every agent through FIVE_add_rule (agents whose record is shorter sit out)
    BCHK(hipMemsetAsync(b->d_nrules, 0, sizeof(int32_t) * E, b->s), "nrules reset");
    BCHK(hipMemsetAsync(b->d_fus, 0, sizeof(int32_t) * E, b->s), "fus reset");            // frirl_utils.c:262
    size_t rmax = 0;
    for (size_t e = 0; e < E; e++) { const auto &rec = recs[e < recs.size() ? e : recs.size() - 1]; rmax = std::max(rmax, rec.size() / (n + 1)); }
    std::vector<double> stage(E * (n + 1));
    std::vector<uint8_t> act(E);
    uint8_t *d_act = nullptr;
    BCHK(hipMalloc((void **)&d_act, E), "active mask");
    int rc = FRIRL_HIP_OK;
    for (size_t r = 0; r < rmax && rc == FRIRL_HIP_OK; r++) {
        for (size_t e = 0; e < E; e++) {
            const auto &rec = recs[e < recs.size() ? e : recs.size() - 1];
            const bool has = r < rec.size() / (n + 1);
            act[e] = has ? 1 : 0;
            for (size_t k = 0; k < n; k++) stage[e * n + k] = has ? rec[r * (n + 1) + k] : 0.0;
            stage[E * n + e] = has ? rec[r * (n + 1) + n] : 0.0;
        }
        if (hipMemcpyAsync(b->d_tmp, stage.data(), sizeof(double) * E * (n + 1), hipMemcpyHostToDevice, b->s) != hipSuccess ||
            hipMemcpyAsync(d_act, act.data(), E, hipMemcpyHostToDevice, b->s) != hipSuccess) { set_error("frirl_hip_batch_load_rulebases: upload failed"); rc = FRIRL_HIP_ELAUNCH; break; }
        rc = five_hip_add_rule(&b->t, &b->rb, b->d_tmp, b->d_tmp + E * n, d_act, b->d_rant, nullptr, b->s);
        if (rc == FRIRL_HIP_OK && hipStreamSynchronize(b->s) != hipSuccess) { set_error("frirl_hip_batch_load_rulebases: add_rule failed"); rc = FRIRL_HIP_ELAUNCH; }
    }
    (void)hipFree(d_act);
    if (rc) return rc;
    if ((rc = frirl_hip_convergence_init(&b->rb, b->nant, &b->conv, b->s))) return rc;
    BCHK(hipStreamSynchronize(b->s), "load sync");
    return FRIRL_HIP_OK;
}
