/*
 * ref_harness.c -- drives the GENUINE reference (compiled by oracle/Makefile from the sources
 * under /root/reference into oracle/_ref/) to produce golden vectors and a CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY (build container + cpu_baseline timing); never part of the product.
 * This file contains no reference code: it calls the reference's exported C API
 * (src/five/FIVE.h:79-103, src/frirl/frirl.h:42-64) and each example's own main(), which the
 * Makefile renames to ref_main_<env>; main()'s call to frirl_run() is redirected to
 * harness_run() below so the harness receives the reference-built frirl_desc, including the
 * reference's own do_action / get_reward / quantize_observations callbacks.
 *
 * Modes (all output is JSON-lines with C hex-floats):
 *   demo    <env> <outdir>            whole construct run; <env>.frirlrb.txt (written by the example's
 *                                     own main) + <env>.trace.jsonl (first steps, per-episode records,
 *                                     running FNV-1a hash over every step of the run)
 *   vectors <env> <outfile> <nep>     run <nep> episodes, then function-level vectors on that rule base
 *   synth   <nant> <U> <R> <A> <seed> <nq> <outfile>   large synthetic bases (inputs from orc_synth_*)
 *   bench   <nant> <U> <R> <nq>       times five_rule_distance / FIVE_vag_concl (cpu_baseline "reference")
 *   reduce  <env> <outdir> <strategy> construct run, then the reference's rule-base reduction
 *                                     (frirl_sequential_run.c:170-350) on the result; <env>.reduced<strategy>.frirlrb.txt
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "frirl_types.h"
#include "frirl.h"
#include "frirl_app_helpers.h"
#include "FIVE.h"
#include "frirl_oracle.h"

int ref_main_mountaincar(int, char **);
int ref_main_cartpole(int, char **);
int ref_main_acrobot(int, char **);

/* ------------------------------------------------------------------ state */
static const char *g_mode, *g_env, *g_out;
static int g_nep = 0;
static FILE *g_fp;
static uint64_t g_hash;
static long g_step;
static int g_ep_steps, g_ep;
static double g_ep_reward;
static int g_nstates;
#ifndef TRACE_STEPS
#define TRACE_STEPS 400
#endif

static void (*o_do_action)(struct frirl_desc *, fri_float, fri_float *, int, fri_float *);
static void (*o_get_reward)(struct frirl_desc *, fri_float *, int, struct frirl_reward_desc *);
static void (*o_quant)(struct frirl_desc *, fri_float *, int, fri_float *);

static void jd(FILE *fp, double v) { fprintf(fp, "\"%a\"", v); }
static void jarr(FILE *fp, const char *key, const double *v, int n)
{
    fprintf(fp, "\"%s\":[", key);
    for (int i = 0; i < n; i++) { if (i) fputc(',', fp); jd(fp, v[i]); }
    fputc(']', fp);
}

static void flush_episode(struct frirl_desc *fr)
{
    if (g_ep_steps == 0) return;
    fprintf(g_fp, "{\"k\":\"ep\",\"ep\":%d,\"steps\":%d,\"reward\":", g_ep, g_ep_steps); jd(g_fp, g_ep_reward);
    fprintf(g_fp, ",\"R\":%d,\"hash\":\"%016llx\"}\n", fr->fiverb->numofrules, (unsigned long long)g_hash);
    g_ep_steps = 0; g_ep_reward = 0;
}

/* wrappers: hash exactly what orc_episode() hashes, in the same order */
static void w_do_action(struct frirl_desc *fr, fri_float a, fri_float *s, int n, fri_float *ns)
{
    if (fr->reward.ep_total_steps == 0 && g_ep_steps > 0) flush_episode(fr);
    if (g_ep_steps == 0) g_ep++;
    o_do_action(fr, a, s, n, ns);
    g_hash = orc_hash_doubles(g_hash, &a, 1);
    g_hash = orc_hash_doubles(g_hash, ns, n);
    if (g_step < TRACE_STEPS) { fprintf(g_fp, "{\"k\":\"step\",\"t\":%ld,\"ep\":%d,\"a\":", g_step, g_ep); jd(g_fp, a); fputc(',', g_fp); jarr(g_fp, "s", ns, n); }
}

static void w_get_reward(struct frirl_desc *fr, fri_float *s, int n, struct frirl_reward_desc *rw)
{
    o_get_reward(fr, s, n, rw);
    double rs[2] = { rw->value, (double)rw->success };
    g_hash = orc_hash_doubles(g_hash, rs, 2);
    g_ep_reward += rw->value;
    if (g_step < TRACE_STEPS) { fprintf(g_fp, ",\"r\":"); jd(g_fp, rw->value); fprintf(g_fp, ",\"f\":%d", rw->success); }
}

static void w_quant(struct frirl_desc *fr, fri_float *s, int n, fri_float *ns)
{
    o_quant(fr, s, n, ns);
    double nr = (double)fr->fiverb->numofrules;
    g_hash = orc_hash_doubles(g_hash, ns, n);
    g_hash = orc_hash_doubles(g_hash, &nr, 1);
    if (g_step < TRACE_STEPS) { fputc(',', g_fp); jarr(g_fp, "q", ns, n); fprintf(g_fp, ",\"R\":%d}\n", fr->fiverb->numofrules); }
    g_step++; g_ep_steps++;
}

/* ------------------------------------------------------------------ vectors */
static uint64_t g_rng = 0xF1F0ULL;
static double rnd(void) { return orc_rand_unit(&g_rng); }

static void dim_range(struct frirl_desc *fr, int k, double *lo, double *hi)
{
    struct FIVERB *f = fr->fiverb;
    *lo = f->uk[k][0]; *hi = f->uk[k][f->univlength - 1];
}

static double grid_value(struct frirl_desc *fr, int k)
{
    struct frirl_dimension_desc *d = (k < fr->statedims_len) ? &fr->statedims[k] : &fr->actiondim;
    return d->values[(int)(rnd() * d->values_len) % d->values_len];
}

static void emit_tables(struct frirl_desc *fr)
{
    struct FIVERB *f = fr->fiverb;
    int n = f->numofunivs, U = f->univlength;
    fprintf(g_fp, "{\"k\":\"tables\",\"nant\":%d,\"U\":%d,\"A\":%d,\"p\":%d,\"R\":%d,", n, U, fr->actiondim.values_len, f->p, f->numofrules);
    fprintf(g_fp, "\"u_hash\":\"%016llx\",\"ve_hash\":\"%016llx\",", (unsigned long long)orc_hash_doubles(0, f->u, n * U),
            (unsigned long long)orc_hash_doubles(0, f->ve, n * U));
    jarr(g_fp, "udivs", f->udivs, n); fputc(',', g_fp);
    jarr(g_fp, "vevalues", fr->possible_actions->vevalues, fr->actiondim.values_len);
    if (U <= 64) { fputc(',', g_fp); jarr(g_fp, "u", f->u, n * U); fputc(',', g_fp); jarr(g_fp, "ve", f->ve, n * U); }
    else { /* sampled */
        double su[5 * 21], sv[5 * 21]; int c = 0;
        for (int k = 0; k < n; k++) for (int j = 0; j < U; j += 50) { su[c] = f->uk[k][j]; sv[c] = f->vek[k][j]; c++; }
        fputc(',', g_fp); jarr(g_fp, "u_s50", su, c); fputc(',', g_fp); jarr(g_fp, "ve_s50", sv, c);
    }
    fprintf(g_fp, "}\n");
    /* the rule base the vectors below are evaluated on */
    fprintf(g_fp, "{\"k\":\"rb\","); jarr(g_fp, "rant", f->rant, f->numofrules * n); fputc(',', g_fp);
    jarr(g_fp, "rconc", f->rconc, f->numofrules); fprintf(g_fp, "}\n");
}

static void emit_snap(struct frirl_desc *fr)
{
    /* get_vag_abs_min_i_fixres (src/inl/min.inl:71-92) is static inline; observe it through
     * FIVE_add_rule's rseqant_uindex (src/five/five_add_rule.c:76) on a scratch rule base. */
    struct FIVERB *f = fr->fiverb;
    int n = f->numofunivs, U = f->univlength, NQ = 600;
    double *rant = aligned_alloc(32, sizeof(double) * (NQ + 8) * n), *rconc = aligned_alloc(32, sizeof(double) * (NQ + 8));
    memset(rant, 0, sizeof(double) * (NQ + 8) * n); memset(rconc, 0, sizeof(double) * (NQ + 8));
    struct FIVERB *s = FIVEInit(f->u, f->ve, 0, n, U, 0, NQ + 8, n + 1, rant, rconc);
    double *pts = malloc(sizeof(double) * NQ * n);
    for (int q = 0; q < NQ; q++) {
        double x[FIVE_MAX_NUM_OF_UNIVERSES];
        for (int k = 0; k < n; k++) {
            double lo, hi; dim_range(fr, k, &lo, &hi);
            int j = (int)(rnd() * (U - 1));            /* never the last cell: the reference reads one past the row there */
            switch (q % 6) {
                case 0: x[k] = f->uk[k][j]; break;                                   /* on a node */
                case 1: x[k] = (f->uk[k][j] + f->uk[k][j + 1]) * 0.5; break;          /* mid-point: tie -> low */
                case 2: x[k] = lo - rnd(); break;                                    /* below */
                case 3: x[k] = hi + f->udivs[k] * (1.5 + rnd()); break;              /* above (low >= len) */
                default: x[k] = lo + (f->uk[k][U - 2] - lo) * rnd(); break;          /* anywhere but the last cell */
            }
            pts[q * n + k] = x[k];
        }
        FIVE_add_rule(s, x, 0.0);
    }
    fprintf(g_fp, "{\"k\":\"snap\",\"n\":%d,", NQ); jarr(g_fp, "pts", pts, NQ * n);
    fprintf(g_fp, ",\"idx\":[");
    for (int q = 0; q < NQ; q++) for (int k = 0; k < n; k++) fprintf(g_fp, "%s%u", (q || k) ? "," : "", s->rseqant_uindex[k][q]);
    fprintf(g_fp, "]}\n");
    free(pts);
}

static void make_query(struct frirl_desc *fr, int kind, double *x)
{
    struct FIVERB *f = fr->fiverb;
    int n = f->numofunivs;
    if (kind == 0) {          /* continuous: no hit */
        for (int k = 0; k < n; k++) { double lo, hi; dim_range(fr, k, &lo, &hi); x[k] = lo + (f->uk[k][f->univlength - 2] - lo) * rnd(); }
    } else if (kind == 1) {   /* an existing rule: exact hit */
        int r = (int)(rnd() * f->numofrules) % f->numofrules;
        for (int k = 0; k < n; k++) x[k] = f->rant[r * n + k];
    } else {                  /* allowed grid point: hit or miss */
        for (int k = 0; k < n; k++) x[k] = grid_value(fr, k);
    }
}

static void emit_five(struct frirl_desc *fr)
{
    struct FIVERB *f = fr->fiverb;
    int n = f->numofunivs, R = f->numofrules;
    for (int q = 0; q < 240; q++) {
        double x[FIVE_MAX_NUM_OF_UNIVERSES], conc = 0;
        make_query(fr, q % 3, x);
        int ret = five_rule_distance(f, x);
        fprintf(g_fp, "{\"k\":\"five\","); jarr(g_fp, "x", x, n); fprintf(g_fp, ",\"ret\":%d", ret);
        if (ret == -1) {
            fprintf(g_fp, ",\"d_hash\":\"%016llx\",", (unsigned long long)orc_hash_doubles(0, f->ruledists, R));
            if (q < 12) jarr(g_fp, "d", f->ruledists, R); else jarr(g_fp, "d", f->ruledists, R < 8 ? R : 8);
        }
        unsigned h = FIVE_vag_concl(f, x, &conc);
        fprintf(g_fp, ",\"vc_ret\":%d,\"conc\":", (int)h); jd(g_fp, conc);
        unsigned hw = FIVE_vag_concl_weight(f, x, f->weights);
        fprintf(g_fp, ",\"w_ret\":%d", (int)hw);
        if (hw == ~0u) {
            fprintf(g_fp, ",\"w_hash\":\"%016llx\",", (unsigned long long)orc_hash_doubles(0, f->weights, R));
            if (q < 12) jarr(g_fp, "w", f->weights, R); else jarr(g_fp, "w", f->weights, R < 8 ? R : 8);
        }
        fprintf(g_fp, "}\n");
    }
}

static void emit_gba(struct frirl_desc *fr)
{
    struct FIVERB *f = fr->fiverb;
    int ns = fr->statedims_len, A = fr->actiondim.values_len;
    for (int q = 0; q < 120; q++) {
        double x[FIVE_MAX_NUM_OF_UNIVERSES];
        make_query(fr, q % 3, x);
        unsigned a = frirl_get_best_action(fr, x);
        fprintf(g_fp, "{\"k\":\"gba\","); jarr(g_fp, "s", x, ns); fprintf(g_fp, ",\"best\":%u,", a);
        jarr(g_fp, "actconc", fr->fgba_actconc, A); fprintf(g_fp, "}\n");
    }
    (void)f;
}

static void emit_cps(struct frirl_desc *fr)
{
    int n = fr->statedims_len + 1;
    for (int q = 0; q < 200; q++) {
        double obs[FIVE_MAX_NUM_OF_UNIVERSES], out[FIVE_MAX_NUM_OF_UNIVERSES];
        for (int k = 0; k < n; k++) {
            struct frirl_values_desc *pv = (k < fr->statedims_len) ? &fr->possible_states[k] : fr->possible_actions;
            double lo = pv->values[0], hi = pv->values[pv->values_len - 1], span = hi - lo;
            obs[k] = (q % 4 == 0) ? pv->values[(int)(rnd() * pv->values_len) % pv->values_len] : lo - 0.2 * span + 1.4 * span * rnd();
            if (q % 4 == 1 && pv->values_len > 1) { int j = (int)(rnd() * (pv->values_len - 1)); obs[k] = (pv->values[j] + pv->values[j + 1]) * 0.5; }
            out[k] = frirl_check_possible_states(fr, obs[k], pv);
        }
        fprintf(g_fp, "{\"k\":\"cps\","); jarr(g_fp, "obs", obs, n); fputc(',', g_fp); jarr(g_fp, "out", out, n); fprintf(g_fp, "}\n");
    }
}

static void emit_env(struct frirl_desc *fr)
{
    int ns = fr->statedims_len;
    for (int q = 0; q < 300; q++) {
        double s[8], nsv[8], qv[8]; struct frirl_reward_desc rw; memset(&rw, 0, sizeof rw);
        for (int k = 0; k < ns; k++) {
            double lo = fr->statedims[k].values[0], hi = fr->statedims[k].values[fr->statedims[k].values_len - 1];
            s[k] = lo - 0.3 * (hi - lo) + 1.6 * (hi - lo) * rnd();
        }
        double a = fr->actiondim.values[(int)(rnd() * fr->actiondim.values_len) % fr->actiondim.values_len];
        o_do_action(fr, a, s, ns, nsv);
        o_get_reward(fr, nsv, ns, &rw);
        o_quant(fr, nsv, ns, qv);
        fprintf(g_fp, "{\"k\":\"env\",\"a\":"); jd(g_fp, a); fputc(',', g_fp); jarr(g_fp, "s", s, ns); fputc(',', g_fp);
        jarr(g_fp, "ns", nsv, ns); fprintf(g_fp, ",\"r\":"); jd(g_fp, rw.value); fprintf(g_fp, ",\"f\":%d,", rw.success);
        jarr(g_fp, "q", qv, ns); fprintf(g_fp, "}\n");
    }
}

static void emit_sarsa(struct frirl_desc *fr)
{
    struct FIVERB *f = fr->fiverb;
    int n = f->numofunivs;
    for (int q = 0; q < 400; q++) {
        double qa[FIVE_MAX_NUM_OF_UNIVERSES], cq[FIVE_MAX_NUM_OF_UNIVERSES], reward;
        for (int k = 0; k < n; k++) { qa[k] = grid_value(fr, k); cq[k] = grid_value(fr, k); }
        if (q % 5 == 3) for (int k = 0; k < n - 1; k++) {          /* off-grid state: Shepard spread */
            double lo, hi; dim_range(fr, k, &lo, &hi); lo *= 0.5; hi *= 0.5; qa[k] = lo + (hi - lo) * rnd(); }
        switch (q % 4) { case 0: reward = -10; break; case 1: reward = 1000; break; case 2: reward = -3000 * rnd(); break; default: reward = 20 * rnd() - 10; }
        double fus_before = fr->fus_is_rule_inserted;
        frirl_update_sarsa(fr, qa, reward, cq);
        fprintf(g_fp, "{\"k\":\"sarsa\","); jarr(g_fp, "q_ant", qa, n); fputc(',', g_fp); jarr(g_fp, "cur_q_ant", cq, n);
        fprintf(g_fp, ",\"reward\":"); jd(g_fp, reward);
        fprintf(g_fp, ",\"fus_before\":%d,\"fus_after\":%d,\"R\":%d,\"rconc_hash\":\"%016llx\",\"rant_hash\":\"%016llx\"", (int)fus_before,
                (int)fr->fus_is_rule_inserted, f->numofrules, (unsigned long long)orc_hash_doubles(0, f->rconc, f->numofrules),
                (unsigned long long)orc_hash_doubles(0, f->rant, f->numofrules * n));
        if (q % 40 == 39) { fputc(',', g_fp); jarr(g_fp, "rconc", f->rconc, f->numofrules); }
        fprintf(g_fp, "}\n");
    }
}

/* ------------------------------------------------------------------ redirected frirl_run */
void harness_run(struct frirl_desc *fr, int verbose)
{
    fr->construct_rb = 1; fr->reduce_rb = 0;      /* mountaincar ships reduce-only (mountaincar.c:240-242); SURVEY 4 */
    fr->original_learning = 1;
    g_nstates = fr->statedims_len;
    o_do_action = fr->do_action_func; o_get_reward = fr->get_reward_func; o_quant = fr->quant_obs_func;
    if (!strcmp(g_mode, "reduce")) {
        char name[256];
        frirl_run(fr, verbose);                                  /* construct */
        fr->construct_rb = 0; fr->reduce_rb = 1; fr->reduction_strategy = g_nep;
        frirl_sequential_run(fr);                                /* reduction phase only */
        snprintf(name, sizeof name, "%s.reduced%d.frirlrb.txt", g_env, g_nep);
        frirl_save_rb_to_text_file(fr, name);
        fprintf(g_fp, "{\"k\":\"reduced\",\"env\":\"%s\",\"strategy\":%d,\"R\":%d,\"steps\":%d,\"reward\":", g_env, g_nep, fr->fiverb->numofrules,
                fr->reward.ep_total_steps); jd(g_fp, fr->reward.ep_total_value); fprintf(g_fp, "}\n");
        fr->reduce_rb = 0;
        return;
    }
    if (!strcmp(g_mode, "demo")) {
        fr->do_action_func = w_do_action; fr->get_reward_func = w_get_reward; fr->quant_obs_func = w_quant;
        fprintf(g_fp, "{\"k\":\"hdr\",\"env\":\"%s\",\"nstates\":%d,\"A\":%d,\"U\":%d}\n", g_env, fr->statedims_len,
                fr->actiondim.values_len, fr->statedims[0].universe_len);
        frirl_run(fr, verbose);
        flush_episode(fr);
        fprintf(g_fp, "{\"k\":\"end\",\"total_steps\":%ld,\"episodes\":%d,\"R\":%d,\"hash\":\"%016llx\",\"epended\":%d}\n", g_step, g_ep,
                fr->fiverb->numofrules, (unsigned long long)g_hash, fr->epended);
        fr->do_action_func = o_do_action; fr->get_reward_func = o_get_reward; fr->quant_obs_func = o_quant;
    } else {
        fr->max_episodes = g_nep + 1;             /* at most max_episodes-1 episodes run (frirl_sequential_run.c:51,59) */
        frirl_run(fr, verbose);
        emit_tables(fr);
        emit_snap(fr);
        emit_cps(fr);
        emit_env(fr);
        emit_five(fr);
        emit_gba(fr);
        emit_sarsa(fr);
    }
}

/* ------------------------------------------------------------------ synthetic bases */
static struct FIVERB *synth_base(int nant, int U, int R, int A, uint64_t seed, double **pu, double **pve)
{
    double *u = aligned_alloc(32, sizeof(double) * nant * U + 64), *ve = aligned_alloc(32, sizeof(double) * nant * U + 64);
    orc_synth_tables(nant, U, seed, u, ve);
    uint32_t *uidx = malloc(sizeof(uint32_t) * (size_t)nant * R);
    double *rc = malloc(sizeof(double) * R);
    orc_synth_rules(nant, U, R, A, seed, uidx, rc);
    size_t cap = (size_t)R + 8;
    double *rant = aligned_alloc(32, ((sizeof(double) * cap * nant + 31) / 32) * 32), *rconc = aligned_alloc(32, ((sizeof(double) * cap + 31) / 32) * 32);
    memset(rant, 0, sizeof(double) * cap * nant); memset(rconc, 0, sizeof(double) * cap);
    struct FIVERB *f = FIVEInit(u, ve, 0, nant, U, 0, (int)cap, nant + 1, rant, rconc);
    double x[FIVE_MAX_NUM_OF_UNIVERSES];
    for (int r = 0; r < R; r++) {
        for (int k = 0; k < nant; k++) x[k] = u[k * U + uidx[(size_t)k * R + r]];
        FIVE_add_rule(f, x, rc[r]);
    }
    free(uidx); free(rc);
    *pu = u; *pve = ve;
    return f;
}

static void synth_query(struct FIVERB *f, uint64_t *rng, int q, double *x)
{
    int n = f->numofunivs, U = f->univlength;
    if (q % 8 == 7) { int r = (int)(orc_splitmix64(rng) % (uint64_t)f->numofrules); for (int k = 0; k < n; k++) x[k] = f->rant[(size_t)r * n + k]; }
    else for (int k = 0; k < n; k++) { double lo = f->uk[k][0], hi = f->uk[k][U - 2]; x[k] = lo + (hi - lo) * orc_rand_unit(rng); }
}

static int run_synth(int argc, char **argv)
{
    int nant = atoi(argv[2]), U = atoi(argv[3]), R = atoi(argv[4]), A = atoi(argv[5]);
    uint64_t seed = strtoull(argv[6], NULL, 0); int nq = atoi(argv[7]);
    FILE *fp = fopen(argv[8], "w");
    double *u, *ve; struct FIVERB *f = synth_base(nant, U, R, A, seed, &u, &ve);
    fprintf(fp, "{\"k\":\"synth\",\"nant\":%d,\"U\":%d,\"R\":%d,\"A\":%d,\"seed\":%llu,\"nq\":%d,\"veval_hash\":\"%016llx\"}\n", nant, U, R, A,
            (unsigned long long)seed, nq, (unsigned long long)orc_hash_doubles(0, f->rseqant_veval[nant - 1], R));
    uint64_t rng = seed * 77 + 5;
    for (int q = 0; q < nq; q++) {
        double x[FIVE_MAX_NUM_OF_UNIVERSES], conc = 0;
        synth_query(f, &rng, q, x);
        int ret = five_rule_distance(f, x);
        fprintf(fp, "{\"k\":\"sq\",\"q\":%d,\"ret\":%d", q, ret);
        if (ret == -1) fprintf(fp, ",\"d_hash\":\"%016llx\"", (unsigned long long)orc_hash_doubles(0, f->ruledists, R));
        unsigned h = FIVE_vag_concl(f, x, &conc);
        fprintf(fp, ",\"vc_ret\":%d,\"conc\":", (int)h); jd(fp, conc);
        unsigned hw = FIVE_vag_concl_weight(f, x, f->weights);
        if (hw == ~0u) fprintf(fp, ",\"w_hash\":\"%016llx\"", (unsigned long long)orc_hash_doubles(0, f->weights, R));
        fprintf(fp, "}\n");
    }
    fclose(fp);
    return 0;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int run_bench(int argc, char **argv)
{
    int nant = atoi(argv[2]), U = atoi(argv[3]), R = atoi(argv[4]), nq = atoi(argv[5]);
    double *u, *ve; struct FIVERB *f = synth_base(nant, U, R, 0, 12345, &u, &ve);
    uint64_t rng = 99; double x[FIVE_MAX_NUM_OF_UNIVERSES], acc = 0, conc;
    double *qs = malloc(sizeof(double) * nq * nant);
    for (int q = 0; q < nq; q++) { synth_query(f, &rng, 0, x); memcpy(qs + q * nant, x, sizeof(double) * nant); }
    for (int q = 0; q < (nq < 8 ? nq : 8); q++) five_rule_distance(f, qs + q * nant);     /* warm-up */
    double t0 = now_s();
    for (int q = 0; q < nq; q++) { five_rule_distance(f, qs + q * nant); acc += f->ruledists[q % R]; }
    double t1 = now_s();
    int nq2 = nq / 8 > 8 ? nq / 8 : nq;                 /* Shepard leg: 1/8 of the queries keeps the sample bounded */
    for (int q = 0; q < nq2; q++) { FIVE_vag_concl(f, qs + q * nant, &conc); acc += conc; }
    double t2 = now_s();
    printf("{\"kind\":\"reference\",\"nant\":%d,\"U\":%d,\"R\":%d,\"nq\":%d,\"rule_distance_s\":%.6f,\"rule_distance_evals_per_s\":%.6e,"
           "\"vag_concl_s\":%.6f,\"vag_concl_evals_per_s\":%.6e,\"check\":%.3e}\n", nant, U, R, nq, t1 - t0, (double)R * nq / (t1 - t0),
           t2 - t1, (double)R * nq2 / (t2 - t1), acc);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness demo|vectors|synth|bench ...\n"); return 2; }
    g_mode = argv[1];
    if (!strcmp(g_mode, "synth")) return (argc >= 9) ? run_synth(argc, argv) : 2;
    if (!strcmp(g_mode, "bench")) return (argc >= 6) ? run_bench(argc, argv) : 2;
    if (argc < 4) return 2;
    g_env = argv[2]; g_out = argv[3];
    char *av[] = { "ref", "-q", NULL };
    if (!strcmp(g_mode, "demo")) {
        if (chdir(g_out) != 0) { perror("chdir"); return 1; }
        char p[256]; snprintf(p, sizeof p, "%s.trace.jsonl", g_env);
        g_fp = fopen(p, "w");
    } else if (!strcmp(g_mode, "reduce")) {
        if (argc < 5) return 2;
        g_nep = atoi(argv[4]);
        if (chdir(g_out) != 0) { perror("chdir"); return 1; }
        char p[256]; snprintf(p, sizeof p, "%s.reduce%d.jsonl", g_env, g_nep);
        g_fp = fopen(p, "w");
    } else if (!strcmp(g_mode, "vectors")) {
        if (argc < 5) return 2;
        g_nep = atoi(argv[4]);
        g_fp = fopen(g_out, "w");
        if (chdir("/tmp") != 0) return 1;     /* the example's main() drops its .frirlrb files in cwd */
    } else return 2;
    if (!g_fp) { perror("open"); return 1; }
    if (!strcmp(g_env, "mountaincar")) ref_main_mountaincar(2, av);
    else if (!strcmp(g_env, "cartpole")) ref_main_cartpole(2, av);
    else if (!strcmp(g_env, "acrobot")) ref_main_acrobot(2, av);
    else return 2;
    fclose(g_fp);
    return 0;
}
