// mirror.hip -- device mirror of ONE struct FIVERB with host-pointer entry points (E = 1).
//
// This is the thin layer the ANSI-C drop-in host library calls: each function stages its small
// inputs through a pinned buffer, launches the same kernels as the batched API on a private stream,
// copies the result back and synchronises.  The legacy single-agent path is latency-bound (a few
// hundred rules per sweep), not bandwidth-bound; throughput comes from the batched API.
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "device_common.h"

using namespace frirl_host;

struct five_hip_mirror {
    int32_t nant, U, maxR, p;
    double *d_u, *d_ve;          // tables
    double *d_rb;                // [nant+1][maxR]
    int32_t *d_nrules;
    double *d_buf;               // scratch: observations, outputs, weights/dists rows
    double *d_row;               // [maxR] distances / weights
    double *d_grid;              // [nant][MAX_GRID] agent grid
    double *d_rant;              // [nant][maxR] raw antecedents of appended rules
    int32_t *d_ibuf;             // fus, status, done...
    double *h_pin;               // pinned staging
    int32_t *h_ipin;
    void *h_out, *d_out;         // pinned + device-mapped result block of the fused step (written by the kernel)
    uint32_t step_seq;           // number of the last fused step (completion flag of the result block)
    double *h_rconc, *d_rconc;   // pinned + device-mapped copy of the consequents after the step
    double grid_sig;             // checksum of the agent grid last uploaded to d_grid
    int grid_valid;
    int32_t R;                   // host copy of numofrules
    hipStream_t s;
    frirl_hip_tables t;
    frirl_hip_rulebases b;
};

#define HIPCHK(call, what)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e_)); return FRIRL_HIP_ELAUNCH; } \
    } while (0)

static const int PIN_DOUBLES = 4096;

int frirl_launch_mirror_step(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *ag, const double *q_ant, double reward,
                             const double *cur_q_states, const double *action_ve, const double *action_values, int A, int fus, double *rant_store,
                             void *out_dev, double *rconc_out_dev, uint32_t seq, hipStream_t s);
size_t frirl_mirror_step_out_bytes();
bool frirl_mirror_step_done(const void *out_host, uint32_t seq);
void frirl_mirror_step_unpack(const void *out_host, int nant, int A, uint32_t *best, double *actconc, double *cur_q_ant, int32_t *fus, int32_t *status,
                              int32_t *nrules, double *new_rant, double *new_rconc);

namespace frirl {
__global__ void bestact_kernel(const double *__restrict__ rb, const int32_t *__restrict__ nrules, int maxR, int nant, int p,
                               const double *__restrict__ dists, double *__restrict__ conc);
__global__ void remove_rule_kernel(double *__restrict__ rb, int maxR, int cols, int R, int r);
}

extern "C" five_hip_mirror *five_hip_mirror_create(int32_t nant, int32_t U, const double *u, const double *ve, int32_t maxR, int32_t p)
{
    if (nant < 1 || nant > FRIRL_HIP_MAX_NANT || U < 2 || !u || !ve || maxR < 2) { set_error("five_hip_mirror_create: bad arguments"); return nullptr; }
    if (check_device()) return nullptr;
    five_hip_mirror *m = (five_hip_mirror *)calloc(1, sizeof(*m));
    if (!m) return nullptr;
    m->nant = nant; m->U = U; m->maxR = maxR + (maxR & 1); m->p = p > 0 ? p : nant;
    const size_t tb = sizeof(double) * nant * U, sb = sizeof(double) * (size_t)(nant + 1) * m->maxR;
    bool ok = hipStreamCreateWithFlags(&m->s, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&m->d_u, tb) == hipSuccess && hipMalloc(&m->d_ve, tb) == hipSuccess;
    ok = ok && hipMalloc(&m->d_rb, sb) == hipSuccess && hipMalloc(&m->d_nrules, 64) == hipSuccess;
    ok = ok && hipMalloc(&m->d_buf, sizeof(double) * PIN_DOUBLES) == hipSuccess && hipMalloc(&m->d_row, sizeof(double) * m->maxR) == hipSuccess;
    ok = ok && hipMalloc(&m->d_grid, sizeof(double) * FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID) == hipSuccess;
    ok = ok && hipMalloc(&m->d_rant, sizeof(double) * (size_t)nant * m->maxR) == hipSuccess;
    ok = ok && hipMalloc(&m->d_ibuf, 256) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&m->h_pin, sizeof(double) * PIN_DOUBLES, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&m->h_ipin, 256, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc(&m->h_out, frirl_mirror_step_out_bytes(), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && hipHostGetDevicePointer(&m->d_out, m->h_out, 0) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&m->h_rconc, sizeof(double) * m->maxR, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
         hipHostGetDevicePointer((void **)&m->d_rconc, m->h_rconc, 0) == hipSuccess;
    if (ok) {
        memset(m->h_out, 0, frirl_mirror_step_out_bytes());          // completion flag starts at step 0
        ok = hipMemcpyAsync(m->d_u, u, tb, hipMemcpyHostToDevice, m->s) == hipSuccess &&
             hipMemcpyAsync(m->d_ve, ve, tb, hipMemcpyHostToDevice, m->s) == hipSuccess &&
             hipMemsetAsync(m->d_rb, 0, sb, m->s) == hipSuccess && hipMemsetAsync(m->d_nrules, 0, 64, m->s) == hipSuccess &&
             hipMemsetAsync(m->d_ibuf, 0, 256, m->s) == hipSuccess &&
             hipMemsetAsync(m->d_rant, 0, sizeof(double) * (size_t)nant * m->maxR, m->s) == hipSuccess &&
             hipStreamSynchronize(m->s) == hipSuccess;
    }
    if (!ok) {
        set_error("five_hip_mirror_create: HIP allocation/copy failed: %s", hipGetErrorString(hipGetLastError()));
        five_hip_mirror_destroy(m);
        return nullptr;
    }
    m->t.nant = nant; m->t.U = U; m->t.u = m->d_u; m->t.ve = m->d_ve;
    m->b.E = 1; m->b.maxR = m->maxR; m->b.rb = m->d_rb; m->b.nrules = m->d_nrules; m->b.uidx = nullptr;
    m->R = 0;
    return m;
}

extern "C" void five_hip_mirror_destroy(five_hip_mirror *m)
{
    if (!m) return;
    if (m->s) (void)hipStreamSynchronize(m->s);
    (void)hipFree(m->d_u); (void)hipFree(m->d_ve); (void)hipFree(m->d_rb); (void)hipFree(m->d_nrules); (void)hipFree(m->d_buf);
    (void)hipFree(m->d_row); (void)hipFree(m->d_grid); (void)hipFree(m->d_rant); (void)hipFree(m->d_ibuf);
    if (m->h_pin) (void)hipHostFree(m->h_pin);
    if (m->h_ipin) (void)hipHostFree(m->h_ipin);
    if (m->h_out) (void)hipHostFree(m->h_out);
    if (m->h_rconc) (void)hipHostFree(m->h_rconc);
    if (m->s) (void)hipStreamDestroy(m->s);
    free(m);
}

extern "C" int32_t five_hip_mirror_numofrules(const five_hip_mirror *m) { return m ? m->R : -1; }

static int set_nrules(five_hip_mirror *m, int32_t R)
{
    m->R = R;
    m->h_ipin[0] = R;
    HIPCHK(hipMemcpyAsync(m->d_nrules, m->h_ipin, sizeof(int32_t), hipMemcpyHostToDevice, m->s), "nrules upload");
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_upload(five_hip_mirror *m, int32_t R, const double *const *veval_rows, const double *rconc)
{
    if (!m || R < 0 || R > m->maxR || (R && (!veval_rows || !rconc))) { set_error("five_hip_mirror_upload: bad arguments"); return FRIRL_HIP_EINVAL; }
    HIPCHK(hipMemsetAsync(m->d_rb, 0, sizeof(double) * (size_t)(m->nant + 1) * m->maxR, m->s), "slab clear");
    for (int k = 0; k < m->nant && R; k++)
        HIPCHK(hipMemcpyAsync(m->d_rb + (size_t)k * m->maxR, veval_rows[k], sizeof(double) * R, hipMemcpyHostToDevice, m->s), "veval upload");
    if (R) HIPCHK(hipMemcpyAsync(m->d_rb + (size_t)m->nant * m->maxR, rconc, sizeof(double) * R, hipMemcpyHostToDevice, m->s), "rconc upload");
    int rc = set_nrules(m, R);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(m->s), "upload sync");
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_add_rule(five_hip_mirror *m, const double *rant, double rconc)
{
    if (!m || !rant) { set_error("five_hip_mirror_add_rule: NULL"); return FRIRL_HIP_EINVAL; }
    if (m->R >= m->maxR) { set_error("five_hip_mirror_add_rule: rule base full (%d)", m->maxR); return FRIRL_HIP_EINVAL; }
    memcpy(m->h_pin, rant, sizeof(double) * m->nant);
    m->h_pin[m->nant] = rconc;
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_pin, sizeof(double) * (m->nant + 1), hipMemcpyHostToDevice, m->s), "add_rule upload");
    int rc = five_hip_add_rule(&m->t, &m->b, m->d_buf, m->d_buf + m->nant, nullptr, m->d_rant, nullptr, m->s);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(m->s), "add_rule sync");
    m->R++;
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_remove_rule(five_hip_mirror *m, uint32_t r)
{
    if (!m || (int32_t)r >= m->R) { set_error("five_hip_mirror_remove_rule: rule %u out of range", r); return FRIRL_HIP_EINVAL; }
    hipLaunchKernelGGL(frirl::remove_rule_kernel, dim3(m->nant + 1), dim3(256), 0, m->s, m->d_rb, m->maxR, m->nant + 1, m->R, (int)r);
    hipLaunchKernelGGL(frirl::remove_rule_kernel, dim3(m->nant), dim3(256), 0, m->s, m->d_rant, m->maxR, m->nant, m->R, (int)r);
    int rc = check_launch("five_hip_mirror_remove_rule");
    if (rc) return rc;
    if ((rc = set_nrules(m, m->R - 1))) return rc;
    HIPCHK(hipStreamSynchronize(m->s), "remove_rule sync");
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_set_rconc(five_hip_mirror *m, const double *rconc, int32_t R)
{
    if (!m || !rconc || R < 0 || R > m->maxR) { set_error("five_hip_mirror_set_rconc: bad arguments"); return FRIRL_HIP_EINVAL; }
    if (R) HIPCHK(hipMemcpyAsync(m->d_rb + (size_t)m->nant * m->maxR, rconc, sizeof(double) * R, hipMemcpyHostToDevice, m->s), "rconc upload");
    HIPCHK(hipStreamSynchronize(m->s), "set_rconc sync");
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_get_rconc(five_hip_mirror *m, double *rconc, int32_t R)
{
    if (!m || !rconc || R < 0 || R > m->maxR) { set_error("five_hip_mirror_get_rconc: bad arguments"); return FRIRL_HIP_EINVAL; }
    if (R) HIPCHK(hipMemcpyAsync(rconc, m->d_rb + (size_t)m->nant * m->maxR, sizeof(double) * R, hipMemcpyDeviceToHost, m->s), "rconc download");
    HIPCHK(hipStreamSynchronize(m->s), "get_rconc sync");
    return FRIRL_HIP_OK;
}

static int upload_obs(five_hip_mirror *m, const double *x, int n, int at)
{
    memcpy(m->h_pin + at, x, sizeof(double) * n);
    HIPCHK(hipMemcpyAsync(m->d_buf + at, m->h_pin + at, sizeof(double) * n, hipMemcpyHostToDevice, m->s), "observation upload");
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_rule_distance(five_hip_mirror *m, const double *x, double *ruledists, uint32_t *hit)
{
    if (!m || !x || !hit) { set_error("five_hip_mirror_rule_distance: NULL"); return FRIRL_HIP_EINVAL; }
    int rc = upload_obs(m, x, m->nant, 0);
    if (rc) return rc;
    uint32_t *d_hit = reinterpret_cast<uint32_t *>(m->d_ibuf);
    if ((rc = five_hip_rule_distance(&m->t, &m->b, m->d_buf, ruledists ? m->d_row : nullptr, d_hit, m->s))) return rc;
    HIPCHK(hipMemcpyAsync(m->h_ipin, d_hit, sizeof(uint32_t), hipMemcpyDeviceToHost, m->s), "hit download");
    if (ruledists && m->R) HIPCHK(hipMemcpyAsync(ruledists, m->d_row, sizeof(double) * m->R, hipMemcpyDeviceToHost, m->s), "ruledists download");
    HIPCHK(hipStreamSynchronize(m->s), "rule_distance sync");
    *hit = (uint32_t)m->h_ipin[0];
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_vag_concl(five_hip_mirror *m, const double *x, double *conc, uint32_t *hit)
{
    if (!m || !x || !conc || !hit) { set_error("five_hip_mirror_vag_concl: NULL"); return FRIRL_HIP_EINVAL; }
    int rc = upload_obs(m, x, m->nant, 0);
    if (rc) return rc;
    uint32_t *d_hit = reinterpret_cast<uint32_t *>(m->d_ibuf);
    if ((rc = five_hip_vag_concl(&m->t, &m->b, m->p, m->d_buf, m->d_buf + 64, d_hit, m->s))) return rc;
    HIPCHK(hipMemcpyAsync(m->h_pin + 64, m->d_buf + 64, sizeof(double), hipMemcpyDeviceToHost, m->s), "conc download");
    HIPCHK(hipMemcpyAsync(m->h_ipin, d_hit, sizeof(uint32_t), hipMemcpyDeviceToHost, m->s), "hit download");
    HIPCHK(hipStreamSynchronize(m->s), "vag_concl sync");
    *conc = m->h_pin[64];
    *hit = (uint32_t)m->h_ipin[0];
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_vag_concl_weight(five_hip_mirror *m, const double *x, double *weights, uint32_t *hit)
{
    if (!m || !x || !weights || !hit) { set_error("five_hip_mirror_vag_concl_weight: NULL"); return FRIRL_HIP_EINVAL; }
    int rc = upload_obs(m, x, m->nant, 0);
    if (rc) return rc;
    uint32_t *d_hit = reinterpret_cast<uint32_t *>(m->d_ibuf);
    if ((rc = five_hip_vag_concl_weight(&m->t, &m->b, m->p, m->d_buf, m->d_row, d_hit, m->s))) return rc;
    HIPCHK(hipMemcpyAsync(m->h_ipin, d_hit, sizeof(uint32_t), hipMemcpyDeviceToHost, m->s), "hit download");
    HIPCHK(hipStreamSynchronize(m->s), "vag_concl_weight sync");
    *hit = (uint32_t)m->h_ipin[0];
    if (*hit == FRIRL_HIP_NO_HIT && m->R) {     // exact hit: weights stay untouched, as in the reference
        HIPCHK(hipMemcpyAsync(weights, m->d_row, sizeof(double) * m->R, hipMemcpyDeviceToHost, m->s), "weights download");
        HIPCHK(hipStreamSynchronize(m->s), "weights sync");
    }
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_bestact(five_hip_mirror *m, const double *ruledists, double *conc)
{
    if (!m || !ruledists || !conc) { set_error("five_hip_mirror_bestact: NULL"); return FRIRL_HIP_EINVAL; }
    if (m->R) HIPCHK(hipMemcpyAsync(m->d_row, ruledists, sizeof(double) * m->R, hipMemcpyHostToDevice, m->s), "ruledists upload");
    int rc = five_hip_bestact(&m->b, m->nant, m->p, m->d_row, m->d_buf + 64, m->s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(m->h_pin + 64, m->d_buf + 64, sizeof(double), hipMemcpyDeviceToHost, m->s), "conc download");
    HIPCHK(hipStreamSynchronize(m->s), "bestact sync");
    *conc = m->h_pin[64];
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_get_best_action(five_hip_mirror *m, const double *states, const double *action_ve, int32_t A,
                                               double *actconc, uint32_t *best)
{
    if (!m || !states || !action_ve || !actconc || !best || A < 1 || A > FRIRL_HIP_MAX_ACTIONS) { set_error("five_hip_mirror_get_best_action: bad arguments"); return FRIRL_HIP_EINVAL; }
    memcpy(m->h_pin, states, sizeof(double) * (m->nant - 1));
    memcpy(m->h_pin + 32, action_ve, sizeof(double) * A);
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_pin, sizeof(double) * (32 + A), hipMemcpyHostToDevice, m->s), "gba upload");
    int32_t *d_best = m->d_ibuf;
    int rc = frirl_hip_get_best_action(&m->t, &m->b, m->p, m->d_buf, m->d_buf + 32, A, m->d_buf + 128, d_best, m->s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(m->h_pin + 128, m->d_buf + 128, sizeof(double) * A, hipMemcpyDeviceToHost, m->s), "actconc download");
    HIPCHK(hipMemcpyAsync(m->h_ipin, d_best, sizeof(int32_t), hipMemcpyDeviceToHost, m->s), "best download");
    HIPCHK(hipStreamSynchronize(m->s), "gba sync");
    memcpy(actconc, m->h_pin + 128, sizeof(double) * A);
    *best = (uint32_t)m->h_ipin[0];
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_update_sarsa(five_hip_mirror *m, const frirl_hip_agent *agent, const double *q_ant, double reward,
                                            const double *cur_q_ant, int32_t *fus, int32_t *status, double *new_rant, double *new_rconc,
                                            double *rconc)
{
    if (!m || !agent || !agent->grid_values || !q_ant || !cur_q_ant || !fus || !status) { set_error("five_hip_mirror_update_sarsa: NULL"); return FRIRL_HIP_EINVAL; }
    const int n = m->nant;
    // staging: [0,n) q_ant, [16,16+n) cur_q_ant, [32] reward, [64, 64 + n*MAX_GRID) grid
    memcpy(m->h_pin, q_ant, sizeof(double) * n);
    memcpy(m->h_pin + 16, cur_q_ant, sizeof(double) * n);
    m->h_pin[32] = reward;
    memcpy(m->h_pin + 64, agent->grid_values, sizeof(double) * n * FRIRL_HIP_MAX_GRID);
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_pin, sizeof(double) * 64, hipMemcpyHostToDevice, m->s), "sarsa upload");
    HIPCHK(hipMemcpyAsync(m->d_grid, m->h_pin + 64, sizeof(double) * n * FRIRL_HIP_MAX_GRID, hipMemcpyHostToDevice, m->s), "grid upload");
    m->h_ipin[0] = *fus;
    HIPCHK(hipMemcpyAsync(m->d_ibuf, m->h_ipin, sizeof(int32_t), hipMemcpyHostToDevice, m->s), "fus upload");
    frirl_hip_agent ag = *agent;
    ag.grid_values = m->d_grid;
    ag.action_ve = nullptr;
    ag.p = m->p;
    frirl_hip_envs ev;
    memset(&ev, 0, sizeof ev);
    ev.fus = m->d_ibuf;
    ev.status = m->d_ibuf + 1;
    ev.rant = m->d_rant;
    int rc = frirl_hip_update_sarsa(&m->t, &m->b, &ag, &ev, m->d_buf, m->d_buf + 32, m->d_buf + 16, nullptr, m->s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(m->h_ipin, m->d_ibuf, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, m->s), "status download");
    HIPCHK(hipMemcpyAsync(m->h_ipin + 2, m->d_nrules, sizeof(int32_t), hipMemcpyDeviceToHost, m->s), "nrules download");
    HIPCHK(hipStreamSynchronize(m->s), "sarsa sync");
    *fus = m->h_ipin[0];
    *status = m->h_ipin[1];
    const int Rold = m->R;
    m->R = m->h_ipin[2];
    if (*status == FRIRL_HIP_UPD_INSERTED && new_rant && new_rconc) {
        for (int k = 0; k < n; k++)
            HIPCHK(hipMemcpyAsync(m->h_pin + k, m->d_rant + (size_t)k * m->maxR + Rold, sizeof(double), hipMemcpyDeviceToHost, m->s), "rant download");
        HIPCHK(hipMemcpyAsync(m->h_pin + 16, m->d_rb + (size_t)n * m->maxR + Rold, sizeof(double), hipMemcpyDeviceToHost, m->s), "rconc download");
        HIPCHK(hipStreamSynchronize(m->s), "sarsa sync 2");
        memcpy(new_rant, m->h_pin, sizeof(double) * n);
        *new_rconc = m->h_pin[16];
    }
    if (rconc && m->R) {
        HIPCHK(hipMemcpyAsync(rconc, m->d_rb + (size_t)n * m->maxR, sizeof(double) * m->R, hipMemcpyDeviceToHost, m->s), "rconc download");
        HIPCHK(hipStreamSynchronize(m->s), "sarsa sync 3");
    }
    return FRIRL_HIP_OK;
}

extern "C" int five_hip_mirror_greedy_step(five_hip_mirror *m, const frirl_hip_agent *agent, const double *q_ant, double reward,
                                           const double *cur_q_states, const double *action_ve, const double *action_values, int32_t A,
                                           uint32_t *best, double *actconc, double *cur_q_ant, int32_t *fus, int32_t *status, double *new_rant,
                                           double *new_rconc, double *rconc)
{
    if (!m || !agent || !agent->grid_values || !q_ant || !cur_q_states || !action_ve || !action_values || !best || !actconc || !cur_q_ant || !fus || !status ||
        A < 1 || A > FRIRL_HIP_MAX_ACTIONS) { set_error("five_hip_mirror_greedy_step: bad arguments"); return FRIRL_HIP_EINVAL; }
    const int n = m->nant;
    double sig = 0.0;              // the agent's grids change rarely: re-upload only when their checksum does
    for (int i = 0; i < n * FRIRL_HIP_MAX_GRID; i++) sig += agent->grid_values[i] * (double)(i + 1);
    for (int k = 0; k < n; k++) sig += 1e6 * agent->grid_len[k] * (k + 1);
    if (!m->grid_valid || sig != m->grid_sig) {
        memcpy(m->h_pin + 64, agent->grid_values, sizeof(double) * n * FRIRL_HIP_MAX_GRID);
        HIPCHK(hipMemcpyAsync(m->d_grid, m->h_pin + 64, sizeof(double) * n * FRIRL_HIP_MAX_GRID, hipMemcpyHostToDevice, m->s), "grid upload");
        m->grid_sig = sig;
        m->grid_valid = 1;
    }
    frirl_hip_agent ag = *agent;
    ag.grid_values = m->d_grid;
    ag.action_ve = nullptr;
    ag.p = m->p;
    const uint32_t seq = ++m->step_seq;
    int rc = frirl_launch_mirror_step(&m->t, &m->b, &ag, q_ant, reward, cur_q_states, action_ve, action_values, A, *fus, m->d_rant, m->d_out, m->d_rconc, seq, m->s);
    if (rc) return rc;
    // wait for the kernel's completion flag in the mapped result block (a few microseconds); hipStreamSynchronize only as the fallback
    // (a kernel that runs long -- thousands of rules -- or fails never sets the flag in time; the stream call reports the error)
    {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        const bool sync_now = opts().mirror_sync != 0;
        while (!frirl_mirror_step_done(m->h_out, seq)) {
            __builtin_ia32_pause();
            if (sync_now || ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400))) {
                HIPCHK(hipStreamSynchronize(m->s), "greedy_step sync");
                if (!frirl_mirror_step_done(m->h_out, seq)) { set_error("five_hip_mirror_greedy_step: the step kernel did not complete"); return FRIRL_HIP_ELAUNCH; }
                break;
            }
        }
    }
    int32_t nr = 0;
    frirl_mirror_step_unpack(m->h_out, n, A, best, actconc, cur_q_ant, fus, status, &nr, new_rant, new_rconc);
    m->R = nr;
    if (rconc && nr) memcpy(rconc, m->h_rconc, sizeof(double) * nr);
    return FRIRL_HIP_OK;
}
