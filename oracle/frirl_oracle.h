/*
 * frirl_oracle.h -- CPU restatement of the FRIRL / FIVE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the package directory,
 * include/, the HIP C-ABI library or the drop-in host library) may include,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root).  The restatement follows the reference's DEFAULT build
 * (BUILD_AVX2, FIVE_FIXRES, FIVE_NONAN, FIVE_NOINF, FRIRL_FAST, DOUBLE_PRECISION,
 * FAST_ABS/POW/SQRT, BUILD_CHECK_STATES -- reference CMakeLists.txt:9-24).
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *  (1) the reference's own golden rule bases (reference tests/orig/ .frirlrb.txt files,
 *      mirrored as data under tests/golden/orig/), and
 *  (2) function-level vectors and whole-run step hashes produced by the genuine
 *      reference compiled in the build container (oracle/_ref, recipe in
 *      oracle/Makefile, generator oracle/ref_harness.c + oracle/make_golden.py).
 */
#ifndef FRIRL_ORACLE_H
#define FRIRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NANT 16   /* reference cap is 8 (src/five/FIVE.h:19); cfg5 needs 16 */
#define ORC_MAX_ACTIONS 64

/* ---- FIVE rule base (restates struct FIVERB, src/five/FIVE.h:24-76) -------- */
typedef struct orc_five {
    int nant;              /* numofunivs == numofantecedents (FRIRL: consequent has no VE) */
    int U;                 /* univlength */
    int p;                 /* Shepard power (FIVEInit.c:89-93: 0 -> rulelength-1 == nant) */
    int R;                 /* numofrules */
    int maxR;              /* maxnumofrules */
    double *u;             /* [nant*U]  universes                */
    double *ve;            /* [nant*U]  vague environments       */
    double udivs[ORC_MAX_NANT]; /* FIVEInit.c:244-248            */
    double *rant;          /* AoS [maxR*nant] raw antecedents    */
    double *veval;         /* SoA [nant][maxR] == rseqant_veval  */
    uint32_t *uidx;        /* SoA [nant][maxR] == rseqant_uindex */
    double *rconc;         /* [maxR] */
    double *ruledists;     /* [maxR] */
    double *weights;       /* [maxR] */
    double *wi;            /* [maxR] */
} orc_five;

/* ---- init-time tables ----------------------------------------------------- */
void orc_gen_fixres_arr(double *arr, int len, double div);                 /* frirl_app_helpers.c:32-44 */
int  orc_gsc_func(const double *u, int numofunivs, int U, const double *psc,
                  int mp, int np, double *scf);                             /* FIVEGScFunc.c:76-255 (nls=NAN, linear) */
void orc_gvagenv(const double *u, int numofunivs, int U, const double *scf,
                 double *ve);                                               /* FIVEGVagEnv.c:40-102 */
unsigned orc_snap(const double *universe, int len, double point, double div); /* min.inl:71-92 */
double orc_fast_pow(double b, int p);                                       /* fast_pow.inl:17-31 */

/* ---- FIVE engine ----------------------------------------------------------- */
orc_five *orc_five_create(const double *u, const double *ve, int p, int nant, int U,
                          int R, int maxR, const double *rant, const double *rconc); /* FIVEInit.c:55-347 */
void orc_five_destroy(orc_five *f);
int  orc_add_rule(orc_five *f, const double *rant, double rconc);           /* five_add_rule.c:47-95 (+capacity check) */
int  orc_remove_rule(orc_five *f, unsigned r);                              /* five_remove_rule.c:29-85 */
int  orc_rule_distance(orc_five *f, const double *x);                       /* five_rule_distance.c:63-295 */
unsigned orc_vag_concl(orc_five *f, const double *ant, double *conc);       /* FIVEVagConcl.c:64-351 */
unsigned orc_vag_concl_weight(orc_five *f, const double *ant, double *weights); /* FIVEVagConclWeight.c:52-188 */
double orc_bestact(orc_five *f, const double *ruledists);                   /* FIVEVagConcl_FRIRL_BestAct.c:56-299 */

/* ---- FRIRL agent (restates struct frirl_desc, src/frirl/frirl_types.h:62-169) */
enum { ORC_ENV_MOUNTAINCAR = 0, ORC_ENV_CARTPOLE = 1, ORC_ENV_ACROBOT = 2, ORC_ENV_COUNT = 3 };

typedef struct orc_dim {
    int values_len;
    double values[ORC_MAX_ACTIONS];
    double values_div, values_steep, values_def;
    double universe_div;
} orc_dim;

typedef struct orc_frirl {
    int env;
    int nstates;                   /* statedims_len */
    int U;                         /* universe_len (same for all dims, frirl_init.c:36) */
    orc_dim statedims[ORC_MAX_NANT];
    orc_dim actiondim;
    double alpha, gamma, epsilon;
    double qdiff_pos_boundary, qdiff_neg_boundary, qdiff_final_tolerance;
    double reward_good_above;
    double weight_significant;     /* rule_weight_considered_significant_for_update */
    int skip_rules, no_random;
    int max_episodes, max_steps, maxR;
    int trig_mode;                 /* 0: libm cos/sin (as the reference); 1: orc_sin/orc_cos (portable, == device) */
    /* runtime */
    orc_five *frb;
    double action_vevalues[ORC_MAX_ACTIONS];   /* frirl_init.c:156-158 */
    double fus_is_rule_inserted;               /* sticky flag, frirl_update_sarsa.c:374,378,73 */
    double *statedistsum, *ruledist;           /* fgba scratch */
    double actconc[ORC_MAX_ACTIONS];
    double reward_value, ep_total_value;
    int ep_total_steps, success;
    unsigned episode_num;
    int epended;
    /* step hash over the whole run (see orc_hash_*) */
    uint64_t step_hash;
    long total_steps;
    /* optional per-step trace sink (tests): called after each env step */
    void (*trace)(struct orc_frirl *fr, int step, double action, const double *cur_states,
                  const double *cur_q_states, void *ud);
    void *trace_ud;
} orc_frirl;

/* agent parameters of the SARSA update in bare form (any rule base) */
typedef struct orc_agent {
    double alpha, gamma, qdiff_pos_boundary, qdiff_neg_boundary, weight_significant;
    int skip_rules;
    const double *grid[ORC_MAX_NANT];   /* possible rule places per antecedent (states, then action) */
    int grid_len[ORC_MAX_NANT];
} orc_agent;
unsigned orc_five_best_action(orc_five *f, const double *states, const double *action_ve, int A, double *actconc);
void orc_five_update_sarsa(orc_five *f, const orc_agent *ag, double *fus, const double *q_ant, double reward, const double *cur_q_ant);
void orc_frirl_agent(const orc_frirl *fr, orc_agent *ag);
void orc_merge_rb(orc_five *rcvr, const orc_agent *ag, const double *newrant, const double *newrconc, int numofrules);   /* frirl_agent.c:58-117 */
int orc_gen_def_states(const orc_five *master, int id, int worldsize, int nstates, double *values_def);               /* frirl_agent.c:121-139 */

void orc_frirl_config(orc_frirl *fr, int env);     /* examples/<env>/<env>.c main(): hyper-parameters as data */
int  orc_frirl_init(orc_frirl *fr);                /* frirl_init.c:29-341, frirl_init_ve.c:25-121, frirl_init_rb.c:86-147 */
void orc_frirl_deinit(orc_frirl *fr);
unsigned orc_get_best_action(orc_frirl *fr, const double *states);                    /* frirl_get_best_action.c:31-341 */
unsigned orc_e_greedy(orc_frirl *fr, const double *states);                           /* frirl_e_greedy_selection.c:21-37 */
double orc_check_possible_states(double obs, const double *values, int values_len);   /* frirl_check_possible_states.c:96-122 */
void orc_update_sarsa(orc_frirl *fr, const double *q_ant, double reward, const double *cur_q_ant); /* frirl_update_sarsa.c:348-385 */
void orc_episode(orc_frirl *fr);                                                      /* frirl_episode.c:28-194 */
int  orc_sequential_run(orc_frirl *fr, int verbose);                                  /* frirl_sequential_run.c:24-165 (construct loop) */
int  orc_save_rb_text(orc_frirl *fr, const char *path);
int  orc_reduce_run(orc_frirl *fr, int strategy, double reward_tolerance);
void orc_episode_eval(orc_frirl *fr);                                                 /* frirl_test_run.c:20-86 */                /* frirl_sequential_run.c:170-350 */                               /* frirl_utils.c:100-144 */

/* ---- environment dynamics (examples/<env>/<env>.c do_action/get_reward/quantize_observations) */
void orc_env_do_action(const orc_frirl *fr, double action, const double *states, double *new_states);
void orc_env_get_reward(const orc_frirl *fr, const double *states, double *reward, int *success);
void orc_env_quantize(const orc_frirl *fr, const double *states, double *new_states);

/* portable FMA-free sin/cos shared (by construction, not by linkage) with the HIP env kernels */
double orc_sin(double x);
double orc_cos(double x);

/* ---- hashing / deterministic generators shared by fixtures and tests -------- */
uint64_t orc_hash_bytes(uint64_t h, const void *p, uint64_t n);   /* FNV-1a 64 */
uint64_t orc_hash_doubles(uint64_t h, const double *p, uint64_t n);
uint64_t orc_splitmix64(uint64_t *state);
double   orc_rand_unit(uint64_t *state);                          /* [0,1) with 53 bits */

/* synthetic problem generator: universes, VE tables and a duplicate-free on-grid rule base.
 * Writes u[nant*U], ve[nant*U], uidx[nant*R] (SoA), rconc[R]. */
void orc_synth_tables(int nant, int U, uint64_t seed, double *u, double *ve);
void orc_synth_rules(int nant, int U, int R, int A, uint64_t seed, uint32_t *uidx, double *rconc);

/* ---- batched helpers (cpu_baseline leg and GPU parity checks) -------------- */
/* E independent rule bases in the device layout rb[E][nant+1][maxR] (column nant = rconc). */
void orc_batch_rule_distance(int E, int nant, int U, int maxR, const double *u, const double *ve,
                             const double *rb, const int32_t *nrules, const double *x /*[E][nant]*/,
                             double *dists /*[E][maxR] or NULL*/, int32_t *hit /*[E]*/, int nthreads);

/* ---- flat accessors for the ctypes binding (tests/, bench.py) -------------- */
orc_frirl *orc_frirl_new(int env, int trig_mode, int maxR);
void orc_frirl_delete(orc_frirl *fr);
orc_five *orc_frirl_frb(orc_frirl *fr);
double *orc_frirl_actconc(orc_frirl *fr);
double *orc_frirl_action_vevalues(orc_frirl *fr);
const orc_dim *orc_frirl_dim(orc_frirl *fr, int k);
int orc_frirl_nstates(orc_frirl *fr);
int orc_frirl_nactions(orc_frirl *fr);
double orc_frirl_get_fus(orc_frirl *fr);
void orc_frirl_set_fus(orc_frirl *fr, double v);
void orc_frirl_set_max_episodes(orc_frirl *fr, int n);
void orc_frirl_set_max_steps(orc_frirl *fr, int n);
void orc_frirl_set_values_def(orc_frirl *fr, int k, double v);
uint64_t orc_frirl_hash(orc_frirl *fr);
long orc_frirl_total_steps(orc_frirl *fr);
unsigned orc_frirl_episode_num(orc_frirl *fr);
int orc_frirl_ep_steps(orc_frirl *fr);
double orc_frirl_ep_reward(orc_frirl *fr);
void orc_frirl_hparams(orc_frirl *fr, double *out8);
void orc_frirl_set_trace(orc_frirl *fr, void (*cb)(orc_frirl *, int, double, const double *, const double *, void *));
int orc_demo_run(int env, int trig_mode, const char *rb_path, uint64_t *hash, long *steps, int *episodes, int *R);

#ifdef __cplusplus
}
#endif
#endif
