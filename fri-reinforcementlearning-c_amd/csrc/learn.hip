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

// lanes per agent for a launch of EXACTLY nlive agents: the largest power of two that keeps all of them resident (2 ... 64)
static int learn_slices(int nlive)
{
    { const int v = opts().learn_slices; if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v; }
    const long lanes = learn_lanes();
    int H = 2;
    while (H < 64 && (long)nlive * (2 * H) <= lanes) H *= 2;
    return H;
}

// How many of `nlive` agents that are still learning the next launch should take, and with how many lanes each.  A launch is at its
// best when it fills the chip (two waves per SIMD); with H lanes per agent that takes lanes / H agents.  The more lanes an agent
// has, the larger the share of a step that is not rule work (the environment's own dynamics, ~1000 instructions, and the butterfly),
// so the plan maximises  occupancy(H) x rule-work share(H)  over H = 2 ... 64 -- and when more agents are alive than fill the chip at
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
    for (int H = 2; H <= 64; H *= 2) {
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
    { const int v = opts().learn_slices; if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) bestH = v; }
    const long cap = lanes / bestH;
    *slices = bestH;
    *agents_per_launch = (int32_t)((long)nlive < cap ? nlive : cap);
    return FRIRL_HIP_OK;
}

extern "C" int frirl_hip_learn_supported(int32_t nant, int32_t U, int32_t A, int32_t p, int32_t env_kind)
{
    if (p > 0 && p != nant) return 0;
    if (opts().learn_persistent == 0) return 0;
    if (env_kind == FRIRL_HIP_ENV_MOUNTAINCAR && nant == 3 && A == 3 && U <= 64) return 1;      // the demos' shapes
    if (env_kind == FRIRL_HIP_ENV_ACROBOT && nant == 5 && A == 3 && U <= 64) return 1;
    // cartpole (21 actions): 22 conclusions per lane do not fit the register file; it stays with the lane groups of lanes.hip
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
    for (int H : {2, 4, 8, 16, 32, 64}) { const size_t n = learn_entries(E, maxR, H); m = n > m ? n : m; }
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
    const int H = learn_slices(nlive);
    if (t->nant == 3) frirl_learn_launch_mountaincar(H, t, b, agent, envs, conv, la, s);
    else if (agent->A == 3) { if (H <= 8) frirl_learn_launch_acrobot_lo(H, t, b, agent, envs, conv, la, s); else frirl_learn_launch_acrobot_hi(H, t, b, agent, envs, conv, la, s); }
    else { set_error("frirl_hip_learn_run: cartpole's 21 actions are not covered"); return FRIRL_HIP_EINVAL; }
    return check_launch("frirl_hip_learn_run");
}
