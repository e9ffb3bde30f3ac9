/*
 * demo_envs.c -- the three demo environments as host callbacks + their descriptors (part of libfrirl_dropin).
 *
 * Own driver for machines without the reference tree (the GPU box): it fills struct frirl_desc with the
 * hyper-parameters of the reference's examples (data from examples/<env>/<env>.c main()), installs host
 * callbacks implementing the same dynamics with libm trig, calls frirl_init / frirl_run (construct mode)
 * and dumps <env>.frirlrb.txt/.bin exactly like the reference's examples.  The reference's own,
 * unchanged example sources also compile and link against these headers and this library.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "frirl.h"
#include "frirl_types_def.h"
#include "frirl_app_helpers.h"
#include "frirl_demo.h"
#include "dropin_internal.h"
#include "frirl_hip.h"

#define PI 3.14159265358979323846264338327

static void generic_quantize(struct frirl_desc *fr, fri_float *s, int n, fri_float *q)
{
    int i;
    for (i = 0; i < n; i++) {
        int where = (int)round((s[i] + fabs(fr->statedims[i].values[0])) / fr->statedims[i].values_div);
        if (where < 0) where = 0;
        else if (where > fr->statedims[i].values_len - 1) where = fr->statedims[i].values_len - 1;
        q[i] = fr->statedims[i].values[where];
    }
}

/* mountain car */
static void mc_do_action(struct frirl_desc *fr, fri_float a, fri_float *s, int n, fri_float *ns)
{
    double v = (s[1] + (0.001 * a) + (-0.0025 * cos(3.0 * s[0]))) * 0.999, p;
    (void)fr; (void)n;
    if (v < -0.07) v = -0.07;
    if (v > 0.07) v = 0.07;
    p = s[0] + v;
    if (p <= -1.5) { p = -1.5; v = 0.0; }
    ns[0] = p; ns[1] = v;
}
static void mc_reward(struct frirl_desc *fr, fri_float *s, int n, struct frirl_reward_desc *rw)
{
    (void)fr; (void)n;
    rw->value = -10; rw->success = 0;
    if (s[0] >= 0.45) { rw->value = 1000; rw->success = 1; }
}

/* cart pole (sin and cos of the same angle through one sincos(), as gcc -O2 compiles the reference) */
static void cp_do_action(struct frirl_desc *fr, fri_float a, fri_float *s, int n, fri_float *ns)
{
    const double g = 9.8, mt = 1.0 + 0.1, mp = 0.1, len = 0.5, pml = mp * len, tau = 0.02, fourthirds = 4.0 / 3.0;
    const double force = a * 10.0;
    double sn, cs, temp, thacc, xacc;
    (void)fr; (void)n;
    sincos(s[2], &sn, &cs);
    temp = (force + pml * s[3] * s[3] * sn) / mt;
    thacc = (g * sn - cs * temp) / (len * (fourthirds - mp * cs * cs / mt));
    xacc = temp - pml * thacc * cs / mt;
    ns[0] = s[0] + tau * s[1];
    ns[1] = s[1] + tau * xacc;
    ns[2] = s[2] + tau * s[3];
    ns[3] = s[3] + tau * thacc;
}
static void cp_reward(struct frirl_desc *fr, fri_float *s, int n, struct frirl_reward_desc *rw)
{
    const double lim = PI / 4;
    (void)fr; (void)n;
    if (s[0] < -4.0 || s[0] > 4.0 || s[2] < (-1 * lim) || s[2] > lim) { rw->value = -10000 - 50 * fabs(s[0]) - 100 * fabs(s[2]); rw->success = 1; }
    else { rw->value = 10 - 1000 * s[2] * s[2] - 5 * fabs(s[0]) - 10 * s[3]; rw->success = 0; }
}
static void cp_quantize(struct frirl_desc *fr, fri_float *s, int n, fri_float *q)
{
    const double d12 = PI / 15, d3 = PI / 60;
    double q0 = s[0], q1 = round(s[1]), q2 = floor(s[2] / d3) * d3, q3 = s[3];
    (void)fr; (void)n;
    if (q0 < 0) q0 = -1;
    if (q0 > 0) q0 = 1;
    if (q1 < -1) q1 = -1;
    if (q1 > 1) q1 = 1;
    if (q2 > d12) q2 = d12;
    if (q2 < (-1 * d12)) q2 = -1 * d12;
    if (q3 < 0) q3 = -1;
    if (q3 > 0) q3 = 1;
    q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
}

/* acrobot */
static void ab_do_action(struct frirl_desc *fr, fri_float torque, fri_float *s, int n, fri_float *ns)
{
    const double vmax1 = 4 * PI, vmax2 = 9 * PI, m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8, dt = 0.05;
    const double l1sq = l1 * l1, lc1sq = lc1 * lc1, lc2sq = lc2 * lc2;
    double t1 = s[0], t2 = s[1], t1d = s[2], t2d = s[3];
    double c2, s2, d1, d2, phi1, phi2, acc1, acc2;
    int i;
    (void)fr; (void)n;
    sincos(t2, &s2, &c2);
    d1 = m1 * lc1sq + m2 * (l1sq + lc2sq + 2 * l1 * lc2 * c2) + I1 + I2;
    d2 = m2 * (lc2sq + l1 * lc2 * c2) + I2;
    phi2 = m2 * lc2 * g * cos(t1 + t2 - PI / 2);
    phi1 = -m2 * l1 * lc2 * t2d * s2 * (t2d - 2 * t1d) + (m1 * lc1 + m2 * l1) * g * cos(t1 - (PI / 2)) + phi2;
    acc2 = (torque + phi1 * (d2 / d1) - m2 * l1 * lc2 * t1d * t1d * s2 - phi2);
    acc2 = acc2 / (m2 * lc2sq + I2 - (d2 * d2 / d1));
    acc1 = -(d2 * acc2 + phi1) / d1;
    for (i = 0; i < 4; i++) {
        t1d = t1d + acc1 * dt;
        if (t1d < -vmax1) t1d = -vmax1;
        if (t1d > vmax1) t1d = vmax1;
        t1 = t1 + t1d * dt;
        t2d = t2d + acc2 * dt;
        if (t2d < -vmax2) t2d = -vmax2;
        if (t2d > vmax2) t2d = vmax2;
        t2 = t2 + t2d * dt;
    }
    if (t1 < -PI) t1 = -PI;
    if (t1 > PI) t1 = PI;
    if (t2 < -PI) t2 = -PI;
    if (t2 > PI) t2 = PI;
    ns[0] = t1; ns[1] = t2; ns[2] = t1d; ns[3] = t2d;
}
static void ab_reward(struct frirl_desc *fr, fri_float *s, int n, struct frirl_reward_desc *rw)
{
    const double y1 = 0.0 - cos(s[0]);
    const double y2 = y1 - cos(s[1]);
    (void)fr; (void)n;
    rw->value = -10; rw->success = 0;
    if (y2 >= 0.0 + 1.0) { rw->value = 1000; rw->success = 1; }
}

static void set_dim(struct frirl_dimension_desc *d, int n, const double *vals, double *store, double vdiv, double steep, double def, int U, double udiv)
{
    memset(d, 0, sizeof *d);
    d->values_len = n; d->values = store; d->values_div = vdiv; d->values_steep = steep; d->values_def = def;
    d->universe_len = U; d->universe_div = udiv;
    d->universe = malloc(sizeof(double) * U);
    if (vals) memcpy(store, vals, sizeof(double) * n); else frirl_gen_fixres_arr(store, n, vdiv);
    frirl_gen_fixres_arr(d->universe, U, udiv);
}

/* fills `fr` (already holding frirl_desc_default or parsed options) for one of the three demos; the
 * dimension descriptors and value arrays are heap-allocated and released by frirl_demo_release(). */
int frirl_demo_setup(struct frirl_desc *fr, const char *env)
{
    struct frirl_dimension_desc *sd = calloc(4, sizeof *sd);
    double (*v)[32] = calloc(5, sizeof *v);
    if (!sd || !v) return -1;
    fr->skip_rules = 1;
    if (!strcmp(env, "mountaincar")) {
        static const double v0[] = {-1.5, -1.295, -1.09, -0.885, -0.68, -0.475, -0.27, -0.065, 0.14, 0.345};
        static const double v1[] = {-0.07, -0.042, -0.014, 0.014, 0.042, 0.07};
        fr->reward_good_above = -5000.0; fr->alpha = 0.5; fr->gamma = 1.0; fr->epsilon = 0.01;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -4.0; fr->qdiff_final_tolerance = 500.0;
        fr->do_action_func = mc_do_action; fr->get_reward_func = mc_reward; fr->quant_obs_func = generic_quantize;
        set_dim(&sd[0], 10, v0, v[0], 0.205, 1.0, -0.5, 41, 0.1);
        set_dim(&sd[1], 6, v1, v[1], 0.028, 1.0, 0.0, 41, 0.005);
        fr->statedims_len = 2;
        set_dim(&fr->actiondim, 3, NULL, v[4], 1.0, 0.0, 0.0, 41, 0.1);
    } else if (!strcmp(env, "cartpole")) {
        static const double v2[] = {-0.2094, -0.1571, -0.1047, -0.0524, 0.0, 0.0524, 0.1047, 0.1571, 0.2094};
        static const double va[] = {-1.0, -0.9, -0.8, -0.7, -0.6, -0.5, -0.3999999999999999, -0.29999999999999992, -0.19999999999999995,
                                    -0.09999999999999998, 0.0, +0.09999999999999998, +0.19999999999999995, +0.29999999999999992,
                                    +0.3999999999999999, +0.5, +0.6, +0.7, +0.8, +0.9, +1.0};
        fr->reward_good_above = 0.0; fr->alpha = 0.3; fr->gamma = 1.0; fr->epsilon = 0.001;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -200.0; fr->qdiff_final_tolerance = 250.0;
        fr->do_action_func = cp_do_action; fr->get_reward_func = cp_reward; fr->quant_obs_func = cp_quantize;
        set_dim(&sd[0], 2, NULL, v[0], 2.0, 1.0, 1.0, 1001, 0.016);
        set_dim(&sd[1], 3, NULL, v[1], 1.0, 1.0, 0.0, 1001, 0.032);
        set_dim(&sd[2], 9, v2, v[2], 0.0524, 21.485917317405871, 0.0, 1001, 0.0031415926535897933);
        set_dim(&sd[3], 2, NULL, v[3], 2.0, 1.0, 0.0, 1001, 0.016);
        fr->statedims_len = 4;
        set_dim(&fr->actiondim, 21, va, v[4], 0.1, 0.0, 0.0, 1001, 0.008);
    } else if (!strcmp(env, "acrobot")) {
        static const double v0[] = {-1.570796326794897, -0.785398163397448, 0, 0.785398163397448, 1.570796326794897};
        fr->reward_good_above = 0.0; fr->alpha = 0.5; fr->gamma = 1.0; fr->epsilon = 0.001;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -200.0; fr->qdiff_final_tolerance = 50.0;
        fr->do_action_func = ab_do_action; fr->get_reward_func = ab_reward; fr->quant_obs_func = generic_quantize;
        set_dim(&sd[0], 5, v0, v[0], 0.785398163397448, 1.0, 0.0, 41, 0.1);
        set_dim(&sd[1], 5, v0, v[1], 0.785398163397448, 1.0, 0.0, 41, 0.1);
        set_dim(&sd[2], 3, NULL, v[2], 0.785398163397448, 1.0, 0.0, 41, 0.05);
        set_dim(&sd[3], 3, NULL, v[3], 0.785398163397448, 1.0, 0.0, 41, 0.05);
        fr->statedims_len = 4;
        set_dim(&fr->actiondim, 3, NULL, v[4], 1.0, 0.0, 0.0, 41, 0.1);
    } else {
        free(sd); free(v); fprintf(stderr, "unknown env %s (mountaincar|cartpole|acrobot)\n", env);
        return -1;
    }
    fr->statedims = sd;
    fr->construct_rb = 1;                 /* construct mode (the reference's mountaincar ships reduce-only) */
    fr->reduce_rb = 0;
    return 0;
}

void frirl_demo_release(struct frirl_desc *fr)
{
    int i;
    double *store = fr->statedims ? fr->statedims[0].values : NULL;
    for (i = 0; i < fr->statedims_len; i++) free(fr->statedims[i].universe);
    free(fr->actiondim.universe);
    free(store);
    free(fr->statedims);
    fr->statedims = NULL;
}

/* Flat description of a demo for bindings (bench.py, tests): tables built by the same host functions
 * frirl_init() uses (frirl_init_ve; per-action VE values as frirl_init.c:156-158).  Needs no GPU. */
int frirl_demo_describe(const char *env, int *nstates, int *U, int *A, double *u, double *ve, double *grid, int *grid_len,
                        double *grid_div, double *values_def, double *action_ve, double *hparams, int *max_steps)
{
    struct frirl_desc fr = frirl_desc_default;
    int k, j, n, usize;
    if (frirl_demo_setup(&fr, env) != 0) return -1;
    n = fr.statedims_len + 1;
    usize = fr.statedims[0].universe_len;
    *nstates = fr.statedims_len; *U = usize; *A = fr.actiondim.values_len; *max_steps = fr.max_steps;
    if (u && ve) {
        for (k = 0; k < fr.statedims_len; k++) memcpy(u + k * usize, fr.statedims[k].universe, sizeof(double) * usize);
        memcpy(u + fr.statedims_len * usize, fr.actiondim.universe, sizeof(double) * usize);
        fr.numofantecedents = n;
        if (frirl_init_ve(&fr, ve, u, usize) != 0) return -1;
        for (k = 0; k < n; k++) {
            const struct frirl_dimension_desc *d = (k < fr.statedims_len) ? &fr.statedims[k] : &fr.actiondim;
            grid_len[k] = d->values_len; grid_div[k] = d->values_div; values_def[k] = d->values_def;
            for (j = 0; j < d->values_len; j++) grid[k * 64 + j] = d->values[j];
        }
        {
            const double *ua = u + fr.statedims_len * usize, *vea = ve + fr.statedims_len * usize;
            const double udiv = (ua[usize - 1] - ua[0]) / (usize - 1);
            for (j = 0; j < fr.actiondim.values_len; j++) action_ve[j] = vea[five_dropin_snap(ua, usize - 1, fr.actiondim.values[j], udiv)];
        }
        hparams[0] = fr.alpha; hparams[1] = fr.gamma; hparams[2] = fr.qdiff_pos_boundary; hparams[3] = fr.qdiff_neg_boundary;
        hparams[4] = fr.rule_weight_considered_significant_for_update; hparams[5] = fr.skip_rules; hparams[6] = fr.reward_good_above;
        hparams[7] = fr.qdiff_final_tolerance;
    }
    frirl_demo_release(&fr);
    return 0;
}

/* E agents of a demo, batched on the device: the C-level counterpart of the reference's many-agent run modes
 * (frirl_agent.c:294-467) without rule-base merging.  The environment's step() runs on the device with the portable
 * trig (include/frirl_hip.h), so the learned rule bases equal the host demo's up to the trig's last-bit differences. */
int frirl_demo_batch_run(const char *env, int agents, int max_episodes, const char *out_txt, int verbose)
{
    return frirl_demo_batch_run_reduce(env, agents, max_episodes, 0, out_txt, verbose);
}

int frirl_demo_batch_run_reduce(const char *env, int agents, int max_episodes, int reduce_strategy, const char *out_txt, int verbose)
{
    return frirl_demo_batch_run_ex(env, agents, max_episodes, reduce_strategy, NULL, NULL, out_txt, verbose);
}

/* host tables + agent descriptor of a demo for frirl_hip_batch_create / frirl_hip_multi_create; release with demo_desc_free */
struct demo_desc_mem { double *u, *ve, *grid, *action_ve, *rant0, *rconc0; };

static int demo_desc_build(const char *env, int agents, frirl_hip_batch_desc *d, struct demo_desc_mem *mm, int *nant_out)
{
    int ns, U, A, max_steps, nant, k, j, R0;
    int grid_len[FRIRL_HIP_MAX_NANT];
    double grid_div[FRIRL_HIP_MAX_NANT], values_def[FRIRL_HIP_MAX_NANT], hp[8];
    memset(mm, 0, sizeof *mm);
    if (frirl_demo_describe(env, &ns, &U, &A, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, &max_steps) != 0) return -1;
    nant = ns + 1;
    mm->u = malloc(sizeof(double) * nant * U); mm->ve = malloc(sizeof(double) * nant * U);
    mm->grid = calloc((size_t)FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID, sizeof(double)); mm->action_ve = calloc(FRIRL_HIP_MAX_ACTIONS, sizeof(double));
    if (!mm->u || !mm->ve || !mm->grid || !mm->action_ve) return -1;
    if (frirl_demo_describe(env, &ns, &U, &A, mm->u, mm->ve, mm->grid, grid_len, grid_div, values_def, mm->action_ve, hp, &max_steps) != 0) return -1;
    R0 = 1 << nant;
    mm->rant0 = malloc(sizeof(double) * R0 * nant); mm->rconc0 = calloc(R0, sizeof(double));
    if (!mm->rant0 || !mm->rconc0) return -1;
    for (k = 0; k < nant; k++) {                               /* frirl_init_rb.c:99-126 */
        double mn = mm->grid[k * FRIRL_HIP_MAX_GRID], mx = mn;
        const unsigned int divider = (unsigned int)R0 >> (k + 1);
        for (j = 0; j < grid_len[k]; j++) { const double v = mm->grid[k * FRIRL_HIP_MAX_GRID + j]; if (v > mx) mx = v; if (v < mn) mn = v; }
        for (j = 0; j < R0; j++) mm->rant0[j * nant + k] = (((j / divider) % 2) == 0) ? mn : mx;
    }
    memset(d, 0, sizeof *d);
    d->nant = nant; d->U = U; d->E = agents; d->maxR = 1024; d->u = mm->u; d->ve = mm->ve; d->R0 = R0; d->rant0 = mm->rant0; d->rconc0 = mm->rconc0;
    d->agent.alpha = hp[0]; d->agent.gamma = hp[1]; d->agent.qdiff_pos_boundary = hp[2]; d->agent.qdiff_neg_boundary = hp[3];
    d->agent.weight_significant = hp[4]; d->agent.skip_rules = (int32_t)hp[5]; d->agent.reward_good_above = hp[6]; d->agent.qdiff_final_tolerance = hp[7];
    d->agent.p = 0; d->agent.A = A; d->agent.max_steps = max_steps; d->agent.no_random = 1;
    d->agent.env_kind = !strcmp(env, "mountaincar") ? FRIRL_HIP_ENV_MOUNTAINCAR : (!strcmp(env, "cartpole") ? FRIRL_HIP_ENV_CARTPOLE : FRIRL_HIP_ENV_ACROBOT);
    for (k = 0; k < nant; k++) { d->agent.grid_len[k] = grid_len[k]; d->agent.grid_div[k] = grid_div[k]; d->agent.values_def[k] = values_def[k]; }
    d->agent.grid_values = mm->grid; d->agent.action_ve = mm->action_ve;
    *nant_out = nant;
    return 0;
}

static void demo_desc_free(struct demo_desc_mem *mm)
{
    free(mm->u); free(mm->ve); free(mm->grid); free(mm->action_ve); free(mm->rant0); free(mm->rconc0);
}

static int dump_rule_base_txt(const char *out_txt, int nant, int R, const double *rant, const double *rconc)
{
    int j, k;
    FILE *fp = fopen(out_txt, "w");
    if (!fp) return -1;
    for (j = 0; j < R; j++) {
        for (k = 0; k < nant; k++) fprintf(fp, "%.18f ", rant[j * nant + k]);
        fprintf(fp, "%.18f \n", rconc[j]);
    }
    fclose(fp);
    return 0;
}

/* `agents` agents of a demo sharded over `gpus` visible devices (0 = all): frirl_hip_multi_* -- one batch and one host thread
 * per device, the per-episode report all-reduced with RCCL.  Prints the job's report; out_txt (or NULL) receives the rule base of
 * the agent with global id 0.  Returns the number of converged agents, or -1. */
int frirl_demo_multi_run(const char *env, int agents, int gpus, int max_episodes, const char *out_txt, int verbose)
{
    frirl_hip_batch_desc d;
    struct demo_desc_mem mm;
    frirl_hip_multi *m;
    frirl_hip_batch_stats_t st;
    int nant = 0, episodes = 0, rc, g;
    int32_t ng = 0, ver = 0;
    int64_t start[64], count[64];
    if (demo_desc_build(env, agents, &d, &mm, &nant) != 0) return -1;
    m = frirl_hip_multi_create(&d, agents, gpus);
    if (!m) five_dropin_fatal("frirl_demo_multi_run(create)", FRIRL_HIP_ENODEV);
    rc = frirl_hip_multi_train(m, max_episodes, &episodes);
    if (rc) five_dropin_fatal("frirl_demo_multi_run(train)", rc);
    rc = frirl_hip_multi_stats(m, &st);
    if (rc) five_dropin_fatal("frirl_demo_multi_run(stats)", rc);
    frirl_hip_multi_info(m, &ng, &ver, NULL, NULL);
    if (ng <= 64) frirl_hip_multi_info(m, &ng, &ver, start, count);
    if (verbose) {
        printf("multi %s: gpus %d (RCCL %d) agents %lld episodes %d converged %lld env-steps %lld mean-rules %.3f mean-reward %.6f (min %.6f max %.6f)\n", env,
               (int)ng, (int)ver, (long long)st.agents, episodes, (long long)st.converged, (long long)st.total_env_steps, st.rules_sum / st.agents,
               st.reward_sum / st.agents, st.reward_min, st.reward_max);
        for (g = 0; g < ng && ng <= 64; g++) printf("multi %s: device %d runs agents [%lld, %lld)\n", env, g, (long long)start[g], (long long)(start[g] + count[g]));
    }
    if (st.full_agents > 0)
        fprintf(stderr, "Warning: %lld of %lld rule bases are at their capacity of %d rules: further rule insertions were refused\n",
                (long long)st.full_agents, (long long)st.agents, (int)d.maxR);
    if (out_txt) {
        int32_t R = 0;
        double *rant = malloc(sizeof(double) * 1024 * nant), *rconc = malloc(sizeof(double) * 1024);
        if (!rant || !rconc || frirl_hip_multi_get_rulebase(m, 0, &R, rant, rconc) != 0 || dump_rule_base_txt(out_txt, nant, R, rant, rconc) != 0) {
            fprintf(stderr, "frirl_demo_multi_run: cannot dump rule base\n");
            return -1;
        }
        free(rant); free(rconc);
    }
    frirl_hip_multi_destroy(m);
    demo_desc_free(&mm);
    return (int)st.converged;
}

/* The reference's many-agent mode WITH the rule-base exchange (frirl_omp_run, frirl_agent.c:294-385 + :424-462) on one GPU: `agents`
 * agents start from different states (gen_def_states), learn in chunks of FRIRL_AGENT_EPCHUNK - 1 = 9 episodes and merge their rule
 * bases after every chunk (frirl_hip_batch_train_merged).  out_txt = the master's (agent 0's) rule base.  Returns the number of merge
 * rounds, or -1. */
int frirl_demo_merged_run(const char *env, int agents, int max_episodes, const char *out_txt, int verbose)
{
    int nant = 0, episodes = 0, rounds = 0, rc;
    frirl_hip_batch_desc d;
    struct demo_desc_mem mm;
    frirl_hip_batch *b;
    frirl_hip_batch_stats_t st;
    double *start;
    if (demo_desc_build(env, agents, &d, &mm, &nant) != 0) return -1;
    start = malloc(sizeof(double) * (size_t)agents * (nant - 1));
    if (!start || frirl_hip_gen_def_states(d.rant0, d.R0, nant, agents, d.agent.values_def, start) != 0) return -1;
    d.start_states = start;
    b = frirl_hip_batch_create(&d);
    if (!b) five_dropin_fatal("frirl_demo_merged_run(create)", FRIRL_HIP_ENODEV);
    rc = frirl_hip_batch_train_merged(b, max_episodes, 10, &episodes, &rounds);
    if (rc) five_dropin_fatal("frirl_demo_merged_run(train)", rc);
    rc = frirl_hip_batch_stats(b, &st);
    if (rc) five_dropin_fatal("frirl_demo_merged_run(stats)", rc);
    if (verbose)
        printf("merged %s: agents %lld episodes %d merge-rounds %d converged %lld env-steps %lld mean-rules %.3f mean-reward %.6f\n", env, (long long)st.agents,
               episodes, rounds, (long long)st.converged, (long long)st.total_env_steps, st.rules_sum / st.agents, st.reward_sum / st.agents);
    if (st.full_agents > 0)
        fprintf(stderr, "Warning: %lld of %lld rule bases are at their capacity of %d rules: further rule insertions were refused\n",
                (long long)st.full_agents, (long long)st.agents, (int)d.maxR);
    if (out_txt) {
        int32_t R = 0;
        double *rant = malloc(sizeof(double) * 1024 * nant), *rconc = malloc(sizeof(double) * 1024);
        if (!rant || !rconc || frirl_hip_batch_get_rulebase(b, 0, &R, rant, rconc) != 0 || dump_rule_base_txt(out_txt, nant, R, rant, rconc) != 0) return -1;
        free(rant); free(rconc);
    }
    frirl_hip_batch_destroy(b);
    demo_desc_free(&mm);
    free(start);
    return rounds;
}

/* The same over `gpus` devices (0 = all): frirl_hip_multi_train_merged -- the master's rule list is broadcast to every device, the devices
 * send their agents' rule lists to device 0 (RCCL), the master takes them over in global agent order.  Returns the merge rounds, or -1. */
int frirl_demo_multi_merged_run(const char *env, int agents, int gpus, int max_episodes, const char *out_txt, int verbose)
{
    int nant = 0, episodes = 0, rounds = 0, rc;
    int32_t ng = 0, ver = 0;
    frirl_hip_batch_desc d;
    struct demo_desc_mem mm;
    frirl_hip_multi *m;
    frirl_hip_batch_stats_t st;
    double *start;
    if (demo_desc_build(env, agents, &d, &mm, &nant) != 0) return -1;
    start = malloc(sizeof(double) * (size_t)agents * (nant - 1));
    if (!start || frirl_hip_gen_def_states(d.rant0, d.R0, nant, agents, d.agent.values_def, start) != 0) return -1;
    d.start_states = start;
    m = frirl_hip_multi_create(&d, agents, gpus);
    if (!m) five_dropin_fatal("frirl_demo_multi_merged_run(create)", FRIRL_HIP_ENODEV);
    rc = frirl_hip_multi_train_merged(m, max_episodes, 10, &episodes, &rounds);
    if (rc) five_dropin_fatal("frirl_demo_multi_merged_run(train)", rc);
    rc = frirl_hip_multi_stats(m, &st);
    if (rc) five_dropin_fatal("frirl_demo_multi_merged_run(stats)", rc);
    frirl_hip_multi_info(m, &ng, &ver, NULL, NULL);
    if (verbose)
        printf("merged %s: gpus %d (RCCL %d) agents %lld episodes %d merge-rounds %d converged %lld env-steps %lld mean-rules %.3f mean-reward %.6f\n", env,
               (int)ng, (int)ver, (long long)st.agents, episodes, rounds, (long long)st.converged, (long long)st.total_env_steps, st.rules_sum / st.agents,
               st.reward_sum / st.agents);
    if (st.full_agents > 0)
        fprintf(stderr, "Warning: %lld of %lld rule bases are at their capacity of %d rules: further rule insertions were refused\n",
                (long long)st.full_agents, (long long)st.agents, (int)d.maxR);
    if (out_txt) {
        int32_t R = 0;
        double *rant = malloc(sizeof(double) * 1024 * nant), *rconc = malloc(sizeof(double) * 1024);
        if (!rant || !rconc || frirl_hip_multi_get_rulebase(m, 0, &R, rant, rconc) != 0 || dump_rule_base_txt(out_txt, nant, R, rant, rconc) != 0) return -1;
        free(rant); free(rconc);
    }
    frirl_hip_multi_destroy(m);
    demo_desc_free(&mm);
    free(start);
    return rounds;
}

int frirl_demo_batch_run_ex(const char *env, int agents, int max_episodes, int reduce_strategy, const char *load_bin, const char *save_bin,
                            const char *out_txt, int verbose)
{
    int nant = 0, episodes = 0, rc;
    frirl_hip_batch_desc d;
    struct demo_desc_mem mm;
    frirl_hip_batch *b;
    frirl_hip_batch_stats_t st;
    if (demo_desc_build(env, agents, &d, &mm, &nant) != 0) return -1;
    d.agent.evaluate = load_bin ? 1 : 0;          /* loaded rule bases are replayed greedily, not trained (frirl_test_run.c:66-70) */
    b = frirl_hip_batch_create(&d);
    if (!b) five_dropin_fatal("frirl_demo_batch_run(create)", FRIRL_HIP_ENODEV);
    if (load_bin) {
        int32_t nrec = 0;
        rc = frirl_hip_batch_load_rulebases(b, load_bin, &nrec);
        if (rc) five_dropin_fatal("frirl_demo_batch_run(load)", rc);
        if (verbose) printf("batch %s: loaded %d rule base record(s) from %s\n", env, (int)nrec, load_bin);
        rc = frirl_hip_batch_episode(b);          /* one greedy episode per agent */
        episodes = 1;
    } else {
        rc = frirl_hip_batch_train(b, max_episodes, &episodes);
    }
    if (rc) five_dropin_fatal("frirl_demo_batch_run(train)", rc);
    rc = frirl_hip_batch_stats(b, &st);
    if (rc) five_dropin_fatal("frirl_demo_batch_run(stats)", rc);
    if (verbose)
        printf("batch %s: agents %lld episodes %d converged %lld env-steps %lld mean-rules %.3f mean-reward %.6f (min %.6f max %.6f)\n", env,
               (long long)st.agents, episodes, (long long)st.converged, (long long)st.total_env_steps, st.rules_sum / st.agents,
               st.reward_sum / st.agents, st.reward_min, st.reward_max);
    if (st.full_agents > 0)      /* the reference never checks the capacity (FIVE_add_rule writes past it); here the append is refused and reported */
        fprintf(stderr, "Warning: %lld of %lld rule bases are at their capacity of %d rules: further rule insertions were refused\n",
                (long long)st.full_agents, (long long)st.agents, (int)d.maxR);
    if (save_bin) {                               /* all agents' rule bases, before any reduction */
        rc = frirl_hip_batch_save_rulebases(b, save_bin);
        if (rc) five_dropin_fatal("frirl_demo_batch_run(save)", rc);
    }
    if (reduce_strategy == 1 || reduce_strategy == 2) {
        frirl_hip_reduce_result rr;
        rc = frirl_hip_batch_reduce(b, 0, reduce_strategy, 0.0, 0, &rr);
        if (rc) five_dropin_fatal("frirl_demo_batch_run(reduce)", rc);
        if (verbose)
            printf("batch %s: agent 0 reduced %d -> %d rules (strategy %d, %d launches, %d replays, episode of %d steps, reward %.6f)\n", env,
                   rr.rules_before, rr.rules_after, reduce_strategy, rr.rounds, rr.rollouts, rr.steps_incremental, rr.reward);
    }
    if (out_txt) {
        int32_t R = 0;
        double *rant = malloc(sizeof(double) * 1024 * nant), *rconc = malloc(sizeof(double) * 1024);
        if (!rant || !rconc || frirl_hip_batch_get_rulebase(b, 0, &R, rant, rconc) != 0 || dump_rule_base_txt(out_txt, nant, R, rant, rconc) != 0) {
            fprintf(stderr, "frirl_demo_batch_run: cannot dump rule base\n");
            return -1;
        }
        free(rant); free(rconc);
    }
    frirl_hip_batch_destroy(b);
    demo_desc_free(&mm);
    return (int)st.converged;
}
