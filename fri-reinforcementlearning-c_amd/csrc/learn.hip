// learn.hip -- entry point of the persistent construct loop (kernels: learn_kernel.h, instantiated in learn_i0/1/2.hip).
#include "learn_kernel.h"

using namespace frirl_host;
int frirl_check_episode(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs, const char *who);

static long learn_lanes()
{
    int cus = 256;
    { int dev = 0, n = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n; }
    return (long)cus * 4 * 2 * FRIRL_WAVE;              // two waves per SIMD (the kernels' register budget)
}

// lanes per agent for a launch of EXACTLY nlive agents: the largest power of two that keeps all of them resident (1 ... 64; one lane per
// agent when more than half the chip's lanes' worth of agents are alive: nothing of a step is then computed twice)
static int learn_slices(int nlive)
{
    { const int v = opts().learn_slices; if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v; }
    const long lanes = learn_lanes();
    int H = 1;
    while (H < 64 && (long)nlive * (2 * H) <= lanes) H *= 2;
    return H;
}

// How many of `nlive` agents that are still learning the next launch should take, and with how many lanes each.  A launch is at its
// best when it fills the chip (two waves per SIMD); with H lanes per agent that takes lanes / H agents.  The more lanes an agent
// has, the larger the share of a step that is not rule work (the environment's own dynamics, ~1000 instructions, and the butterfly),
// so the plan maximises  occupancy(H) x rule-work share(H)  over H = 1 ... 64 -- and when more agents are alive than fill the chip at
// that H, the launch takes the first lanes / H of them and the caller rotates the rest to the front of the next launch (every
// launch full, instead of the half-empty launches a fixed assignment gives between two powers of two).
extern "C" int frirl_hip_learn_plan(int32_t nlive, int32_t mean_rules, int32_t *slices, int32_t *agents_per_launch)
{
    if (nlive < 1 || !slices || !agents_per_launch) { set_error("frirl_hip_learn_plan: bad arguments"); return FRIRL_HIP_EINVAL; }
    const long lanes = learn_lanes();
    const double R = mean_rules > 0 ? mean_rules : 256;
    // time of one step of every live agent, per agent: a wave's step costs a fixed share (the environment's own dynamics, the update
    // logic, the group's dependent loads: ~5200 instruction-equivalents, measured, profiles/r03b_kernels_after_trims.jsonl) plus its
    // rules (~93 instructions each, R / H per lane) plus the butterfly; up to one wave per SIMD the waves run side by side, a lone wave
    // paying `alone` (its stalls are not hidden: measured, a lone wave gets ~half the issue rate -- option learn_alone, in tenths, default 2.0:
    // 0.85 / 0.75 / 0.73 / 0.70 / 0.69 s for 1.0 / 1.3 / 1.6 / 2.0 / 3.0 on the diversified acrobot run); beyond that they share the issue slots.
    const double alone = opts().learn_alone > 0 ? 0.1 * opts().learn_alone : 2.0;
    const double simds = (double)lanes / (2.0 * FRIRL_WAVE);
    int bestH = 2;
    double best = -1.0;
    for (int H = 1; H <= 64; H *= 2) {
        int lg = 0;
        for (int h = H; h > 1; h >>= 1) lg++;
        const double wave_step = 5200.0 + R * 93.0 / H + 40.0 * lg;
        double waves = (double)nlive * H / FRIRL_WAVE;
        double agents = nlive;
        if (waves > 2.0 * simds) { waves = 2.0 * simds; agents = (double)lanes / H; }      // the launch takes what fills the chip
        const double share = waves / simds > alone ? waves / simds : alone;
        const double score = agents / (wave_step * share);                                    // agent-steps per unit of time
        if (score > best) { best = score; bestH = H; }
    }
    { const int v = opts().learn_slices; if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) bestH = v; }
    const long cap = lanes / bestH;
    *slices = bestH;
    *agents_per_launch = (int32_t)((long)nlive < cap ? nlive : cap);
    return FRIRL_HIP_OK;
}

// cartpole's 40 KB of LDS tables leave no room for the cold-state area of 256 one-lane agents per workgroup: at least 2 lanes per agent
static int learn_min_slices(int A) { return A > 8 ? 2 : 1; }

extern "C" int frirl_hip_learn_supported(int32_t nant, int32_t U, int32_t A, int32_t p, int32_t env_kind)
{
    if (p > 0 && p != nant) return 0;
    if (opts().learn_persistent == 0) return 0;
    if (env_kind == FRIRL_HIP_ENV_MOUNTAINCAR && nant == 3 && A == 3 && U <= 64) return 1;      // the demos' shapes
    if (env_kind == FRIRL_HIP_ENV_ACROBOT && nant == 5 && A == 3 && U <= 64) return 1;
    if (env_kind == FRIRL_HIP_ENV_CARTPOLE && nant == 5 && A == 21 && U <= 1024) return 1;      // 22 conclusions: two walks of 11 per step
    return 0;
}

static size_t learn_entries(int nlive, int maxR, int H)
{
    const int EPW = FRIRL_WAVE / H;
    const size_t tiles = ((size_t)nlive + EPW - 1) / EPW, njmax = ((size_t)maxR + H - 1) / H;
    return tiles * (njmax + frirl::LR_PADROWS) * 64;
}

extern "C" size_t frirl_hip_learn_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A)
{
    if (nant < 1 || E < 1 || maxR < 1 || A < 1) return 0;
    size_t m = 0;
    for (int H : {1, 2, 4, 8, 16, 32, 64}) { const size_t n = learn_entries(E, maxR, H); m = n > m ? n : m; }
    return m * (16 + 8 + 8) + 256;
}

#ifdef LEARN_TIMING
static unsigned long long *learn_timing_buffer()
{
    static unsigned long long *p = nullptr;
    if (!p) { if (hipMalloc(&p, 16 * 8) != hipSuccess) return nullptr; (void)hipMemset(p, 0, 16 * 8); }
    return p;
}
extern "C" int frirl_hip_learn_timing(unsigned long long *out16)       // read and reset (experimental builds, tools/exp)
{
    unsigned long long *p = learn_timing_buffer();
    if (!p || hipMemcpy(out16, p, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return FRIRL_HIP_EINVAL;
    (void)hipMemset(p, 0, 16 * 8);
    return FRIRL_HIP_OK;
}
#endif

// One chunk of the construct loop for the agents in `live` (device array of nlive agent ids, or NULL = agents 0..nlive-1):
// see include/frirl_hip.h.
extern "C" int frirl_hip_learn_run(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                                   const frirl_hip_convergence *conv, const int32_t *live, int32_t nlive, int32_t budget_steps, int32_t max_episodes,
                                   int64_t *work, int64_t *steps_total, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = frirl_check_episode(t, b, agent, envs, "frirl_hip_learn_run");
    if (rc) return rc;
    if (!conv || !conv->prev_nrules || !conv->prev_steps || !conv->prev_reward || !conv->prev_rconc || !conv->converged || !conv->episodes) { set_error("frirl_hip_learn_run: NULL convergence state"); return FRIRL_HIP_EINVAL; }
    if (!frirl_hip_learn_supported(t->nant, t->U, agent->A, agent->p, agent->env_kind)) { set_error("frirl_hip_learn_run: shape nant=%d U=%d A=%d p=%d not covered (use frirl_hip_episode_run_lanes)", t->nant, t->U, agent->A, agent->p); return FRIRL_HIP_EINVAL; }
    if (!b->uidx) { set_error("frirl_hip_learn_run: needs the 16-bit index mirror (frirl_hip_rulebases.uidx)"); return FRIRL_HIP_EINVAL; }
    if (nlive < 1 || nlive > b->E || budget_steps < 0 || max_episodes < 1) { set_error("frirl_hip_learn_run: nlive=%d / budget=%d / max_episodes=%d", nlive, budget_steps, max_episodes); return FRIRL_HIP_EINVAL; }
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) { set_error("frirl_hip_learn_run: workspace must be a 16-byte aligned device buffer"); return FRIRL_HIP_EINVAL; }
    const size_t need = frirl_hip_learn_workspace_bytes(t->nant, nlive, b->maxR, agent->A);
    if (workspace_bytes < need) { set_error("frirl_hip_learn_run: workspace %zu B < %zu B (frirl_hip_learn_workspace_bytes)", workspace_bytes, need); return FRIRL_HIP_EINVAL; }
    hipStream_t s = as_stream(stream);
    frirl::LearnArgs la = {};
    la.u = t->u; la.ve = t->ve; la.U = t->U;
    la.Ti = static_cast<uint32_t *>(workspace);
    la.live = live; la.nlive = nlive;
    la.rb = b->rb; la.uidx = b->uidx; la.nrules = b->nrules; la.maxR = b->maxR;
    la.work = work; la.steps_total = steps_total; la.budget = budget_steps; la.max_episodes = max_episodes;
#ifdef LEARN_TIMING
    la.timing = learn_timing_buffer();
#endif
    int H = learn_slices(nlive);
    if (H < learn_min_slices(agent->A)) H = learn_min_slices(agent->A);
    if (t->nant == 3) frirl_learn_launch_mountaincar(H, t, b, agent, envs, conv, la, s);
    else if (agent->A == 3) { if (H <= 8) frirl_learn_launch_acrobot_lo(H, t, b, agent, envs, conv, la, s); else frirl_learn_launch_acrobot_hi(H, t, b, agent, envs, conv, la, s); }
    else { if (H <= 8) frirl_learn_launch_cartpole_lo(H, t, b, agent, envs, conv, la, s); else frirl_learn_launch_cartpole_hi(H, t, b, agent, envs, conv, la, s); }
    return check_launch("frirl_hip_learn_run");
}


// ---- the whole construct loop for E agents: launch plan, the agents' queue, and the compaction between launches -------------------
// (host side of the reference's loop, frirl_sequential_run.c:55-165 run for every agent: nothing here touches a rule)
namespace {
constexpr int LQ_BLOCK = 1024, LQ_BINS = 4096;

__global__ void learn_queue_init_kernel(int32_t *q, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) q[i] = i;
}

// live[0 .. take) = the first `take` agents of the queue, ordered by rule count (counting sort in LDS; agents with the same count in
// any order): agents of one wave walk as many rules as the largest rule base among them, so neighbours should have similar counts
__global__ __launch_bounds__(LQ_BLOCK) void learn_queue_sort_kernel(const int32_t *__restrict__ q, int take, const int32_t *__restrict__ nrules, int shift,
                                                                    int32_t *__restrict__ live)
{
    __shared__ int bins[LQ_BINS];
    __shared__ int part[LQ_BLOCK];
    for (int i = threadIdx.x; i < LQ_BINS; i += LQ_BLOCK) bins[i] = 0;
    __syncthreads();
    auto key = [&](int e) { const int k = nrules[e] >> shift; return k < 0 ? 0 : (k >= LQ_BINS ? LQ_BINS - 1 : k); };
    for (int i = threadIdx.x; i < take; i += LQ_BLOCK) atomicAdd(&bins[key(q[i])], 1);
    __syncthreads();
    constexpr int PER = LQ_BINS / LQ_BLOCK;
    int loc[PER], sum = 0;
#pragma unroll
    for (int x = 0; x < PER; x++) { loc[x] = sum; sum += bins[threadIdx.x * PER + x]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < LQ_BLOCK; off <<= 1) {                 // inclusive scan of the per-thread sums
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const int base = part[threadIdx.x] - sum;
#pragma unroll
    for (int x = 0; x < PER; x++) bins[threadIdx.x * PER + x] = base + loc[x];
    __syncthreads();
    for (int i = threadIdx.x; i < take; i += LQ_BLOCK) { const int e = q[i]; live[atomicAdd(&bins[key(e)], 1)] = e; }
}

// next queue = the agents that had to wait (q[take .. n)), then the agents of this launch that are still learning, in launch order;
// counters[0] = its length, counters[1] = the sum of its agents' rule counts; refused[e] |= "an append was refused" (status)
__global__ __launch_bounds__(LQ_BLOCK) void learn_queue_next_kernel(const int32_t *__restrict__ q, int n, int take, const int32_t *__restrict__ live,
                                                                    const int32_t *__restrict__ converged, const int32_t *__restrict__ episodes, int max_episodes,
                                                                    const int32_t *__restrict__ nrules, const int32_t *__restrict__ status, uint8_t *refused,
                                                                    int32_t *__restrict__ nq, long long *counters)
{
    __shared__ int wsum[LQ_BLOCK / 64];
    __shared__ int base_s;
    __shared__ unsigned long long rsum_s;
    if (threadIdx.x == 0) { base_s = n - take; rsum_s = 0ull; }
    unsigned long long rsum = 0ull;
    for (int i = threadIdx.x; i < n - take; i += LQ_BLOCK) { const int e = q[take + i]; nq[i] = e; rsum += (unsigned long long)nrules[e]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < take; i0 += LQ_BLOCK) {
        const int i = i0 + threadIdx.x;
        int e = 0;
        bool keep = false;
        if (i < take) {
            e = live[i];
            keep = converged[e] == 0 && episodes[e] < max_episodes - 1;
            if (refused && status && status[e] == FRIRL_HIP_UPD_FULL) refused[e] = 1;
        }
        const unsigned long long b = __ballot(keep ? 1 : 0);
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < LQ_BLOCK / 64; w++) { const int c = wsum[w]; if (w < wave) before += c; total += c; }
        if (keep) { nq[base_s + before + __popcll(b & ((1ull << lane) - 1ull))] = e; rsum += (unsigned long long)nrules[e]; }
        __syncthreads();
        if (threadIdx.x == 0) base_s += total;
        __syncthreads();
    }
    atomicAdd(&rsum_s, rsum);
    __syncthreads();
    if (threadIdx.x == 0) { counters[0] = base_s; counters[1] = (long long)rsum_s; }
}

size_t learn_train_tail_bytes(int E) { return ((size_t)3 * E * sizeof(int32_t) + 15) / 16 * 16 + 64; }
}  // namespace

extern "C" size_t frirl_hip_learn_train_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A)
{
    const size_t a = frirl_hip_learn_workspace_bytes(nant, E, maxR, A);
    return a ? (a + 15) / 16 * 16 + learn_train_tail_bytes(E) : 0;
}

extern "C" int frirl_hip_learn_train(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                                     const frirl_hip_convergence *conv, int32_t budget_steps, int32_t max_episodes, int64_t *work, int64_t *steps_total,
                                     uint8_t *refused, void *workspace, size_t workspace_bytes, int32_t *launches_out,
                                     frirl_hip_learn_chunk_fn on_chunk, void *user, void *stream)
{
    int rc = frirl_check_episode(t, b, agent, envs, "frirl_hip_learn_train");
    if (rc) return rc;
    if (!conv || !conv->converged || !conv->episodes) { set_error("frirl_hip_learn_train: NULL convergence state"); return FRIRL_HIP_EINVAL; }
    const int E = b->E;
    const size_t run_bytes = (frirl_hip_learn_workspace_bytes(t->nant, E, b->maxR, agent->A) + 15) / 16 * 16;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < run_bytes + learn_train_tail_bytes(E)) {
        set_error("frirl_hip_learn_train: workspace %zu B < %zu B (frirl_hip_learn_train_workspace_bytes) or not 16-byte aligned", workspace_bytes, run_bytes + learn_train_tail_bytes(E));
        return FRIRL_HIP_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    char *tail = static_cast<char *>(workspace) + run_bytes;
    int32_t *qa = reinterpret_cast<int32_t *>(tail), *qb = qa + E, *live = qb + E;
    long long *counters = reinterpret_cast<long long *>(tail + ((size_t)3 * E * sizeof(int32_t) + 15) / 16 * 16);
    int shift = 0;
    while ((b->maxR >> shift) >= LQ_BINS) shift++;
    hipLaunchKernelGGL(learn_queue_init_kernel, dim3((E + 255) / 256 < 1024 ? (E + 255) / 256 : 1024), dim3(256), 0, s, qa, E);
    int n = E, launches = 0;
    long long mean_rules = 0;
    while (n > 0) {
        int32_t H = 0, take = 0;
        // the plan's cost model counts a rule as 4 conclusions (the 3-action demos): 22 conclusions weigh 5.5 times as much
        if ((rc = frirl_hip_learn_plan(n, (int32_t)(mean_rules * (agent->A + 1) / 4), &H, &take)) != 0) return rc;
        if (H < learn_min_slices(agent->A)) { H = learn_min_slices(agent->A); const long fit = learn_lanes() / H; take = (int32_t)(n < fit ? n : fit); }
        hipLaunchKernelGGL(learn_queue_sort_kernel, dim3(1), dim3(LQ_BLOCK), 0, s, qa, take, b->nrules, shift, live);
        if ((rc = frirl_hip_learn_run(t, b, agent, envs, conv, live, take, budget_steps, max_episodes, work, steps_total, workspace, run_bytes, stream)) != 0) return rc;
        launches++;
        hipLaunchKernelGGL(learn_queue_next_kernel, dim3(1), dim3(LQ_BLOCK), 0, s, qa, n, take, live, conv->converged, conv->episodes, max_episodes, b->nrules,
                           envs->status, refused, qb, counters);
        long long h[2] = {0, 0};
        if (hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return check_launch("frirl_hip_learn_train");
        if (on_chunk) on_chunk(user, launches, live, take);
        int32_t *x = qa; qa = qb; qb = x;
        n = (int)h[0];
        mean_rules = n > 0 ? h[1] / n : 0;
    }
    if (launches_out) *launches_out = launches;
    return check_launch("frirl_hip_learn_train");
}
