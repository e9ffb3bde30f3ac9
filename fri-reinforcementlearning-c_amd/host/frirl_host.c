/*
 * frirl_host.c -- host side of the FRIRL agent (ANSI C), MI355X drop-in for the reference's libfrirl.
 *
 * Exports the reference's agent API (include/frirl.h == reference src/frirl/frirl.h:42-64).  The RL
 * driver (init, episode loop with the application's host callbacks, convergence test) is host code as
 * in the reference; the two hot calls of every environment step -- frirl_get_best_action and
 * frirl_update_sarsa -- are each ONE fused GPU call on the rule base's device mirror.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "dropin_internal.h"

void five_dropin_note_appended(struct FIVERB *frb, const double *rant, double rconc);

/* ---- initialisation ----------------------------------------------------------------------------- */

/* reference src/frirl/frirl_init_ve.c:25-121: scaling points (value, steep, steep) per state grid value;
 * action scaling points at i*2/(A-1)-1 with steepness (A-1)/2 in INTEGER arithmetic (:91-92). */
int frirl_init_ve(struct frirl_desc *frirl, fri_float *ve, fri_float *u, int univlength)
{
    int st, i, c;
    double *scf = MALLOC(sizeof(double) * univlength), *row;
    if (!scf) return -1;
    for (st = 0; st <= frirl->statedims_len; st++) {
        const int is_action = (st == frirl->statedims_len);
        const int n = is_action ? frirl->actiondim.values_len : frirl->statedims[st].values_len;
        double *sp = MALLOC(sizeof(double) * 3 * n);
        if (!sp) return -1;
        if (is_action) {
            const double divratio = 1.0 / (n - 1) * 2.0;
            for (i = 0; i < n; i++) { sp[3 * i] = i * divratio - 1.0; sp[3 * i + 1] = sp[3 * i + 2] = (n - 1) / 2; }
        } else {
            for (c = 0; c < n; c++) { sp[3 * c] = frirl->statedims[st].values[c]; sp[3 * c + 1] = sp[3 * c + 2] = frirl->statedims[st].values_steep; }
        }
        FIVE_GSc_func(u + st * univlength, 1, univlength, sp, n, 3, NAN, scf);
        free(sp);
        row = FIVEGVagEnv(u + st * univlength, 1, univlength, scf);
        if (!row) return -1;
        memcpy(ve + st * univlength, row, sizeof(double) * univlength);
        free(row);
    }
    free(scf);
    return 0;
}

/* reference src/frirl/frirl_init_rb.c:86-147: the 2^nant corner rules (min/max of every grid), Q = 0 */
int frirl_init_rb(struct frirl_desc *frirl, fri_float *rant, fri_float *rconc, int *numofrules)
{
    const int n = frirl->numofantecedents;
    int i, j, c;
    *numofrules = (int)pow(2, n);
    for (j = 0; j < *numofrules; j++) rconc[j] = 0.0;
    for (i = 0; i < n; i++) {
        const struct frirl_dimension_desc *d = (i < frirl->statedims_len) ? &frirl->statedims[i] : &frirl->actiondim;
        double mn = d->values[0], mx = d->values[0];
        const unsigned int divider = (unsigned int)*numofrules >> (i + 1);
        for (c = 0; c < d->values_len; c++) { if (d->values[c] > mx) mx = d->values[c]; if (d->values[c] < mn) mn = d->values[c]; }
        for (j = 0; j < *numofrules; j++) rant[j * n + i] = (((j / divider) % 2) == 0) ? mn : mx;
    }
    return 0;
}

/* reference src/frirl/frirl_init.c:29-341 */
int frirl_init(struct frirl_desc *frirl)
{
    struct timeval t1;
    struct FIVERB *frb;
    const int usize = frirl->statedims[0].universe_len, ns = frirl->statedims_len;
    int i, j, k, numofrules, nant;
    fri_float *u, *ve, *rant, *rconc;

    frirl->numofantecedents = nant = ns + 1;
    u = MALLOC(sizeof(fri_float) * usize * nant);
    ve = MALLOC(sizeof(fri_float) * usize * nant);
    rant = MALLOC(sizeof(fri_float) * (size_t)frirl->five_maxnumofrules * nant);
    rconc = MALLOC(sizeof(fri_float) * frirl->five_maxnumofrules);
    if (!u || !ve || !rant || !rconc) return -1;
    memset(rant, 0, sizeof(fri_float) * (size_t)frirl->five_maxnumofrules * nant);
    memset(rconc, 0, sizeof(fri_float) * frirl->five_maxnumofrules);
    for (i = 0; i < ns; i++) memcpy(u + i * usize, frirl->statedims[i].universe, sizeof(fri_float) * usize);
    memcpy(u + ns * usize, frirl->actiondim.universe, sizeof(fri_float) * usize);

    frirl->possible_states = MALLOC(sizeof(struct frirl_values_desc) * ns);
    frirl->possible_actions = MALLOC(sizeof(struct frirl_values_desc));
    if (!frirl->possible_states || !frirl->possible_actions) return -1;
    for (i = 0; i < ns; i++) {
        struct frirl_values_desc *pv = &frirl->possible_states[i];
        pv->values_len = frirl->statedims[i].values_len;
        pv->values = MALLOC(sizeof(double) * pv->values_len);
        pv->vevalues = NULL;
        if (!pv->values) return -1;
        memcpy(pv->values, frirl->statedims[i].values, sizeof(double) * pv->values_len);
        pv->epsilon = frirl->statedims[i].values_div;
    }
    frirl->possible_actions->values_len = frirl->actiondim.values_len;
    frirl->possible_actions->values = MALLOC(sizeof(fri_float) * frirl->actiondim.values_len);
    frirl->possible_actions->vevalues = MALLOC(sizeof(fri_float) * frirl->actiondim.values_len);
    if (!frirl->possible_actions->values || !frirl->possible_actions->vevalues) return -1;
    memcpy(frirl->possible_actions->values, frirl->actiondim.values, sizeof(fri_float) * frirl->actiondim.values_len);
    frirl->possible_actions->epsilon = frirl->actiondim.values_div;

    frirl->reduction_state = 0;
    if (frirl_init_ve(frirl, ve, u, usize) != 0) return -1;
    frirl_init_rb(frirl, rant, rconc, &numofrules);
    frirl->fiverb = frb = FIVEInit(u, ve, 0, nant, usize, numofrules, frirl->five_maxnumofrules, nant + 1, rant, rconc);
    if (!frb) return -1;

    frirl->reward.ep_total_value = -1;
    frirl->reward.ep_total_steps = -1;
    frirl->fus_is_rule_inserted = 0;
    frirl->fiverb_vea = ve + usize * ns;
    frirl->fiverb_ua = u + usize * ns;
    /* VE value of every action; the reference passes len = uksize = U-1 here (frirl_init.c:156-158) */
    for (j = 0; j < frirl->actiondim.values_len; j++)
        frirl->possible_actions->vevalues[j] = frirl->fiverb_vea[five_dropin_snap(frirl->fiverb_ua, (int)frb->uksize, frirl->possible_actions->values[j], frb->udivs[nant - 1])];

    /* parameter sanity checks (frirl_init.c:160-208) */
    for (k = 0; k < nant; k++) {
        const struct frirl_values_desc *pv = (k < ns) ? &frirl->possible_states[k] : frirl->possible_actions;
        if ((frb->uk[k][usize - 1] - frb->uk[k][0]) == 0.0) { printf("frirl_init: FATAL ERROR: U dimension %d max-min=0.0!\n", k); exit(1); }
        if (pv->values[0] < frb->uk[k][0] || pv->values[pv->values_len - 1] > frb->uk[k][usize - 1]) {
            printf("frirl_init: FATAL ERROR: the grid of dimension %d [%f, %f] leaves its universe [%f, %f]\n", k, pv->values[0],
                   pv->values[pv->values_len - 1], frb->uk[k][0], frb->uk[k][usize - 1]);
            exit(1);
        }
        if (pv->values_len > FRIRL_HIP_MAX_GRID) { printf("frirl_init: FATAL ERROR: dimension %d has %d grid values (max %d)\n", k, pv->values_len, FRIRL_HIP_MAX_GRID); exit(1); }
    }

    /* scratch the reference hangs off the descriptor; kept because applications may read fgba_actconc */
    frirl->fgba_vagdist_states = calloc((size_t)(frirl->five_maxnumofrules + 4) * ns, sizeof(fri_float));
    frirl->fgba_ruledist = calloc(frirl->five_maxnumofrules, sizeof(fri_float));
    frirl->fgba_actconc = calloc(frirl->five_maxnumofrules, sizeof(fri_float));
    frirl->fgba_dists = calloc(4 * FIVE_MAX_NUM_OF_UNIVERSES, sizeof(fri_float));
    frirl->fgba_statedistsum = calloc(frirl->five_maxnumofrules, sizeof(fri_float));
    frirl->fus_values = calloc(frb->rulelength, sizeof(fri_float));
    frirl->fus_proposed_values = calloc(frb->rulelength, sizeof(fri_float));
    frirl->fus_check_states = calloc(nant, sizeof(fri_float));
    frirl->fep_ant = calloc(nant, sizeof(fri_float));
    frirl->fep_cur_ant = calloc(nant, sizeof(fri_float));
    frirl->fep_q_ant = calloc(nant, sizeof(fri_float));
    frirl->fep_cur_q_ant = calloc(nant, sizeof(fri_float));
    if (!frirl->fgba_vagdist_states || !frirl->fgba_ruledist || !frirl->fgba_actconc || !frirl->fgba_dists || !frirl->fgba_statedistsum ||
        !frirl->fus_values || !frirl->fus_proposed_values || !frirl->fus_check_states || !frirl->fep_ant || !frirl->fep_cur_ant ||
        !frirl->fep_q_ant || !frirl->fep_cur_q_ant)
        return -1;

    frirl->is_running = 1;
    frirl->episode_num = 1;
    gettimeofday(&t1, NULL);
    srand((unsigned int)(t1.tv_usec * t1.tv_sec));
    return 0;
}

/* reference src/frirl/frirl_deinit.c:18-60 */
void frirl_deinit(struct frirl_desc *frirl)
{
    int i;
    double *u = frirl->fiverb->u, *ve = frirl->fiverb->ve, *rant = frirl->fiverb->rant, *rconc = frirl->fiverb->rconc;
    five_deinit(frirl->fiverb);
    free(u); free(ve); free(rant); free(rconc);
    free(frirl->fus_proposed_values); free(frirl->fus_values); free(frirl->fus_check_states);
    free(frirl->fgba_statedistsum); free(frirl->fgba_vagdist_states); free(frirl->fgba_ruledist); free(frirl->fgba_actconc); free(frirl->fgba_dists);
    free(frirl->fep_ant); free(frirl->fep_cur_ant); free(frirl->fep_q_ant); free(frirl->fep_cur_q_ant);
    for (i = 0; i < frirl->statedims_len; i++) free(frirl->possible_states[i].values);
    free(frirl->possible_states);
    free(frirl->possible_actions->values); free(frirl->possible_actions->vevalues); free(frirl->possible_actions);
    if (frirl->rbfile) free(frirl->rbfile);
}

/* ---- action selection --------------------------------------------------------------------------- */

/* reference src/frirl/frirl_get_best_action.c:31-341: one fused GPU sweep evaluates all actions */
unsigned int frirl_get_best_action(struct frirl_desc *frirl, fri_float *states)
{
    uint32_t best;
    int rc = five_hip_mirror_get_best_action(five_dropin_mirror(frirl->fiverb), states, frirl->possible_actions->vevalues,
                                             frirl->actiondim.values_len, frirl->fgba_actconc, &best);
    if (rc) five_dropin_fatal("frirl_get_best_action", rc);
    return (unsigned int)best;
}

/* reference src/frirl/frirl_e_greedy_selection.c:21-37; the random index is clamped to A-1 (the
 * reference can return A, one past the last action -- SURVEY Appendix C "fix") */
unsigned int frirl_e_greedy_selection(struct frirl_desc *frirl, fri_float *states)
{
    double r;
    unsigned int a;
    if (frirl->no_random == 1 || frirl->epsilon == 0.0) return frirl_get_best_action(frirl, states);
    r = (double)rand() / (double)RAND_MAX;
    if (r > frirl->epsilon) return frirl_get_best_action(frirl, states);
    a = (unsigned int)round((double)rand() / (double)RAND_MAX * frirl->actiondim.values_len);
    if (a >= (unsigned int)frirl->actiondim.values_len) a = (unsigned int)frirl->actiondim.values_len - 1;
    return a;
}

/* reference src/frirl/frirl_check_possible_states.c:96-122 (+ :53-88): nearest allowed grid value, ties
 * to the upper neighbour; the grid-refinement branch (:68-75) cannot trigger with epsilon = values_div */
fri_float frirl_check_possible_states(struct frirl_desc *frirl, fri_float observation, struct frirl_values_desc *pv)
{
    int i;
    (void)frirl;
    for (i = 0; i < pv->values_len; i++) if (observation < pv->values[i]) break;
    if (i == pv->values_len) return pv->values[pv->values_len - 1];
    if (i == 0) return pv->values[0];
    i--;
    return ((observation - pv->values[i]) < (pv->values[i + 1] - observation)) ? pv->values[i] : pv->values[i + 1];
}

/* ---- SARSA update -------------------------------------------------------------------------------- */

/* the agent's hyper-parameters and grids in the HIP layer's form (host pointers) */
static void dropin_agent(struct frirl_desc *frirl, frirl_hip_agent *ag, double *grid)
{
    const int n = frirl->numofantecedents;
    int k;
    memset(ag, 0, sizeof *ag);
    memset(grid, 0, sizeof(double) * FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID);
    ag->alpha = frirl->alpha; ag->gamma = frirl->gamma;
    ag->qdiff_pos_boundary = frirl->qdiff_pos_boundary; ag->qdiff_neg_boundary = frirl->qdiff_neg_boundary;
    ag->weight_significant = frirl->rule_weight_considered_significant_for_update;
    ag->skip_rules = frirl->skip_rules; ag->p = frirl->fiverb->p; ag->A = frirl->actiondim.values_len;
    for (k = 0; k < n; k++) {
        const struct frirl_values_desc *pv = (k < frirl->statedims_len) ? &frirl->possible_states[k] : frirl->possible_actions;
        ag->grid_len[k] = pv->values_len;
        memcpy(grid + k * FRIRL_HIP_MAX_GRID, pv->values, sizeof(double) * pv->values_len);
    }
    ag->grid_values = grid;
}

/* reference src/frirl/frirl_update_sarsa.c:348-385: the whole TD step (Q(s',a'), Q(s,a), qdiff, grid snap,
 * append or exact / weighted write-back) is one fused GPU call; the host mirrors its bookkeeping. */
void frirl_update_sarsa(struct frirl_desc *frirl, fri_float *q_ant, fri_float reward, fri_float *cur_q_ant)
{
    struct FIVERB *frb = frirl->fiverb;
    frirl_hip_agent ag;
    double grid[FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID];      /* per call, on the stack: distinct frirl_desc instances stay independent (threads) */
    double new_rant[FRIRL_HIP_MAX_NANT], new_rconc = 0.0;
    int32_t fus = (frirl->fus_is_rule_inserted != 0.0), status = 0;
    int rc;

    dropin_agent(frirl, &ag, grid);
    rc = five_hip_mirror_update_sarsa(five_dropin_mirror(frb), &ag, q_ant, reward, cur_q_ant, &fus, &status, new_rant, &new_rconc, frb->rconc);
    if (rc) five_dropin_fatal("frirl_update_sarsa", rc);
    if (status == FRIRL_HIP_UPD_INSERTED) five_dropin_note_appended(frb, new_rant, new_rconc);
    else if (status == FRIRL_HIP_UPD_FULL) fprintf(stderr, "frirl_update_sarsa: rule base full (%d rules), new rule dropped\n", frb->numofrules);
    frirl->fus_is_rule_inserted = (fri_float)fus;
}

/* ---- episode ------------------------------------------------------------------------------------- */

void getActionFromTerminal(struct frirl_desc *frirl);

/* reference src/frirl/frirl_episode.c:28-194 */
void frirl_episode(struct frirl_desc *frirl)
{
    const int ns = frirl->statedims_len, n = frirl->numofantecedents;
    fri_float *q_ant = frirl->fep_q_ant, *cur_q_ant = frirl->fep_cur_q_ant;
    fri_float *states = frirl->fep_ant, *cur_states = frirl->fep_cur_ant;
    unsigned int a, ap;
    int i, step;

    for (i = 0; i < ns; i++) q_ant[i] = states[i] = frirl->statedims[i].values_def;
    frirl->reward.ep_total_value = 0;
    frirl->reward.ep_total_steps = 0;
    if (frirl->original_learning == 0) {               /* imitation: a key, or space for the greedy action */
        getActionFromTerminal(frirl);
        a = (frirl->keyaction == 32 || frirl->keyaction >= (unsigned int)frirl->actiondim.values_len) ? frirl_e_greedy_selection(frirl, states) : frirl->keyaction;
    } else {
        a = frirl_e_greedy_selection(frirl, states);   /* on the un-quantised default state (:78) */
    }
    q_ant[ns] = frirl->actiondim.values[a];

    for (step = 1; step <= frirl->max_steps; step++) {
        frirl->do_action_func(frirl, q_ant[ns], states, ns, cur_states);
        frirl->get_reward_func(frirl, cur_states, ns, &frirl->reward);
        frirl->reward.ep_total_value += frirl->reward.value;
        frirl->quant_obs_func(frirl, cur_states, ns, cur_q_ant);
        if (frirl->original_learning != 0 && frirl->reduction_state == 0 && (frirl->no_random == 1 || frirl->epsilon == 0.0)) {
            /* greedy learning step (every shipped demo): frirl_get_best_action (:148) and frirl_update_sarsa (:159) as ONE
             * GPU launch and one synchronisation; same kernels' arithmetic as the two separate calls */
            double grid[FRIRL_HIP_MAX_NANT * FRIRL_HIP_MAX_GRID];
            frirl_hip_agent ag;
            double new_rant[FRIRL_HIP_MAX_NANT], new_rconc = 0.0;
            int32_t fus = (frirl->fus_is_rule_inserted != 0.0), status = 0;
            uint32_t best = 0;
            int rc;
            dropin_agent(frirl, &ag, grid);
            rc = five_hip_mirror_greedy_step(five_dropin_mirror(frirl->fiverb), &ag, q_ant, frirl->reward.value, cur_q_ant,
                                             frirl->possible_actions->vevalues, frirl->actiondim.values, frirl->actiondim.values_len, &best,
                                             frirl->fgba_actconc, cur_q_ant, &fus, &status, new_rant, &new_rconc, frirl->fiverb->rconc);
            if (rc) five_dropin_fatal("frirl_episode(greedy step)", rc);
            if (status == FRIRL_HIP_UPD_INSERTED) five_dropin_note_appended(frirl->fiverb, new_rant, new_rconc);
            else if (status == FRIRL_HIP_UPD_FULL) fprintf(stderr, "frirl_episode: rule base full (%d rules), new rule dropped\n", frirl->fiverb->numofrules);
            frirl->fus_is_rule_inserted = (fri_float)fus;
            ap = best;
        } else {
            if (frirl->original_learning == 0) {
                getActionFromTerminal(frirl);
                ap = (frirl->keyaction == 32 || frirl->keyaction >= (unsigned int)frirl->actiondim.values_len) ? frirl_e_greedy_selection(frirl, cur_q_ant) : frirl->keyaction;
            } else {
                ap = frirl_e_greedy_selection(frirl, cur_q_ant);
            }
            cur_q_ant[ns] = frirl->actiondim.values[ap];
            if (frirl->reduction_state == 0) frirl_update_sarsa(frirl, q_ant, frirl->reward.value, cur_q_ant);
        }
        for (i = 0; i < ns; i++) states[i] = cur_states[i];
        for (i = 0; i < n; i++) q_ant[i] = cur_q_ant[i];
        frirl->reward.ep_total_steps++;
        if (frirl->reward.success == 1) break;
    }
}

/* ---- run modes ------------------------------------------------------------------------------------ */

/* Rule-base reduction, reference src/frirl/frirl_sequential_run.c:170-350 (strategies 1 and 2): take out the
 * rule with the smallest (1) or largest (2) |Q| that is not yet marked important, replay one greedy episode
 * without updates (reduction_state = 1); keep the removal if the episode still succeeds in the same number of
 * steps with a reward within reduction_reward_tolerance, else reload the saved rule base and mark the rule
 * important (its shadow consequent becomes NaN).  Every replay is GPU work (one greedy sweep per step);
 * five_remove_rule compacts the device slab.  The undo buffer is the reference's temporary file. */
static void frirl_reduce_rb(struct frirl_desc *frirl, double *prev_rconc, double prev_reward)
{
    static const char *tmpfile_name = "reduction_tmp.frirlrb.bin";
    struct FIVERB *frb = frirl->fiverb;
    const size_t bytes = sizeof(double) * frirl->five_maxnumofrules;
    double *shadow = MALLOC(bytes), *removed = MALLOC(sizeof(double) * frb->rulelength);
    int steps_incremental, iterations, redend = 0, j, k;
    unsigned int mindex = 0;
    if (!shadow || !removed) { fprintf(stderr, "frirl_sequential_run: out of memory\n"); exit(1); }
    memcpy(shadow, frb->rconc, bytes);
    frirl->original_learning = 1;
    frirl->reduction_state = 1;
    iterations = (frirl->reduction_strategy == 1 || frirl->reduction_strategy == 2) ? frb->numofrules + 1 : 10000;
    frirl_episode(frirl);
    steps_incremental = frirl->reward.ep_total_steps;
    for (frirl->episode_num = 1; (int)frirl->episode_num <= iterations; frirl->episode_num++) {
        frirl_episode(frirl);
        printf("Reduction Episode: %d\tSteps: %d\tReward: %f\tEpsilon: %f\tRules: %d\n", frirl->episode_num, frirl->reward.ep_total_steps,
               frirl->reward.ep_total_value, frirl->epsilon, frb->numofrules);
        if (frirl->reduction_strategy != 1 && frirl->reduction_strategy != 2) continue;
        if (frirl->episode_num > 1) {
            const double diff = prev_reward - frirl->reward.ep_total_value;
            if (frirl->reward.ep_total_value > frirl->reward_good_above && frirl->reward.ep_total_steps == steps_incremental &&
                fabs(diff) <= frirl->reduction_reward_tolerance) {
                if (frirl->verbose > 0) {
                    printf("Reduction Episode: %d\tEliminated rule: no: %d. - ", frirl->episode_num, mindex + 1);
                    for (j = 0; j < frb->rulelength; j++) printf(" %f", removed[j]);
                    printf(" \tReward diff was: %f\n", diff);
                }
                prev_reward = frirl->reward.ep_total_value;
            } else {
                memcpy(shadow, prev_rconc, bytes);
                shadow[mindex] = 0.0 / 0.0;                       /* important: never a candidate again */
                if (frirl_load_rb_from_bin_file(frirl, tmpfile_name) < 0) { printf("Error while loading the binary rule-base file!\n"); exit(-1); }
                frb = frirl->fiverb;
                if (frirl->verbose > 0) {
                    printf("Reduction Episode: %d\tRule stays: no: %d. - ", frirl->episode_num, mindex + 1);
                    for (j = 0; j < frb->numofantecedents; j++) printf(" %f", frb->rseqant[j][mindex]);
                    printf(" %f\n", frb->rconc[mindex]);
                }
            }
        } else {
            prev_reward = frirl->reward.ep_total_value;
        }
        {   /* next candidate; a NaN shadow value never wins against a number */
            double mvalue = fabs(shadow[0]);
            mindex = 0;
            for (j = 1; j < frb->numofrules; j++) {
                const int better = (frirl->reduction_strategy == 1) ? (mvalue > fabs(shadow[j])) : (mvalue < fabs(shadow[j]));
                if (better || (mvalue != mvalue && shadow[j] == shadow[j])) { mvalue = fabs(shadow[j]); mindex = (unsigned int)j; }
            }
            if (mvalue != mvalue) {
                if (frirl->verbose > 0) printf("Smallest rulebase found. Exiting.\n");
                redend = 1;
            } else {
                frirl_save_rb_to_bin_file(frirl, tmpfile_name);
                for (k = 0; k < frb->numofantecedents; k++) removed[k] = frb->rseqant[k][mindex];
                removed[frb->rulelength - 1] = shadow[mindex];
                memcpy(prev_rconc, shadow, bytes);
                for (k = (int)mindex; k < frb->numofrules - 1; k++) shadow[k] = shadow[k + 1];
                five_remove_rule(frb, mindex);
            }
        }
        if (redend) break;
    }
    free(shadow);
    free(removed);
}

/* reference src/frirl/frirl_sequential_run.c:24-355: incremental construction until the rule base, step count
 * and reward repeat and no consequent moved by >= qdiff_final_tolerance (:55-165), then the optional reduction
 * phase (:170-350). */
/* what the construct loop remembers of the previous episode (:68-72) */
struct episode_mark {
    int rules, steps;
    double reward;
};

static struct episode_mark mark_episode(const struct frirl_desc *frirl)
{
    struct episode_mark m;
    m.rules = frirl->fiverb->numofrules;
    m.steps = frirl->reward.ep_total_steps;
    m.reward = frirl->reward.ep_total_value;
    return m;
}

/* :83-87: the episode repeated the previous one (same rule count, steps and reward) and was a good one */
static int episode_repeats(const struct frirl_desc *frirl, const struct episode_mark *before)
{
    const struct episode_mark now = mark_episode(frirl);
    return before->rules == now.rules && before->steps == now.steps && now.reward > frirl->reward_good_above && before->reward == now.reward;
}

/* :134-148: 1 when no consequent moved by qdiff_final_tolerance or more since `old_rconc` (verbose: every mover is listed) */
static int consequents_settled(const struct frirl_desc *frirl, const double *old_rconc)
{
    const struct FIVERB *frb = frirl->fiverb;
    int settled = 1, r;
    for (r = 0; r < frb->numofrules; r++) {
        const double moved = frb->rconc[r] - old_rconc[r];
        if (fabs(moved) < frirl->qdiff_final_tolerance) continue;
        settled = 0;
        if (frirl->verbose <= 0) break;
        printf("Greater at rule %d. %.18f - %.18f = %.18f (max: %.18f)\n", r, frb->rconc[r], old_rconc[r], moved, frirl->qdiff_final_tolerance);
    }
    return settled;
}

static void report_episode(const struct frirl_desc *frirl)
{
    const double rew = frirl->reward.ep_total_value;
    printf("#%d Episode: %d\tSteps: %d\tReward: %s%f%s\tRules: %d\n", frirl->agent_id, frirl->episode_num, frirl->reward.ep_total_steps,
           rew > frirl->reward_good_above ? TERM_GREEN : TERM_RED, rew, TERM_NC, frirl->fiverb->numofrules);
}

/* the construct phase (:55-165); leaves the consequents / reward of the episode BEFORE the last one in old_rconc / *old_reward,
 * which the reduction phase starts from */
static void construct_rule_base(struct frirl_desc *frirl, double *old_rconc, double *old_reward)
{
    const size_t rconc_bytes = sizeof(double) * frirl->five_maxnumofrules;
    const int agent_mode = frirl->runmode == FRIRL_MPI || frirl->runmode == FRIRL_OMP;
    const int episode_budget = agent_mode ? FRIRL_AGENT_EPCHUNK : frirl->max_episodes;      /* at most budget - 1 episodes per call (:51,59) */
    int done;
    for (done = 1; done < episode_budget; done++) {
        const struct episode_mark before = mark_episode(frirl);
        int complete;
        *old_reward = before.reward;
        memcpy(old_rconc, frirl->fiverb->rconc, rconc_bytes);
        frirl_episode(frirl);
        report_episode(frirl);
        complete = frirl->user_exited == 1;
        if (!complete && episode_repeats(frirl, &before)) {
            frirl->epended = 1;
            if (frirl->verbose > 0) printf("Rule-base size and reward are the same as in the previous iteration.\n");
            complete = consequents_settled(frirl, old_rconc);
        } else if (complete) {
            frirl->epended = 1;
        }
        if (complete) {
            frirl->is_running = 0;
            printf("-----------------------------------------------------------------\n");
            printf("No more significant changes in rule-base. RB considered complete.\n");
            printf("-----------------------------------------------------------------\n");
            return;
        }
        frirl->episode_num++;
    }
    if (!(frirl->episode_num < (unsigned int)frirl->max_episodes)) frirl->is_running = 0;      /* budget used up (:60-62) */
}

void frirl_sequential_run(struct frirl_desc *frirl)
{
    double *old_rconc = MALLOC(sizeof(double) * frirl->five_maxnumofrules);
    double old_reward = frirl->reward.ep_total_value;
    frirl->epended = 0;
    memcpy(old_rconc, frirl->fiverb->rconc, sizeof(double) * frirl->five_maxnumofrules);
    if (frirl->construct_rb == 1) construct_rule_base(frirl, old_rconc, &old_reward);
    if (frirl->reduce_rb == 1) frirl_reduce_rb(frirl, old_rconc, old_reward);
    free(old_rconc);
}

/* reference src/frirl/frirl_agent.c:294-467: experimental OpenMP / MPI agent modes, off by default in
 * the reference build and outside the hot path.  Many agents run through the batched GPU API instead. */
void frirl_omp_run(struct frirl_desc *frirl)
{
    printf("frirl_omp_run: not available in the MI355X build; use the batched frirl_hip_* API for many agents. Running sequentially.\n");
    frirl->runmode = FRIRL_SEQ;
    frirl_sequential_run(frirl);
}

void frirl_mpi_run(struct frirl_desc *frirl)
{
    printf("frirl_mpi_run: not available in the MI355X build; use the batched frirl_hip_* API for many agents. Running sequentially.\n");
    frirl->runmode = FRIRL_SEQ;
    frirl_sequential_run(frirl);
}
