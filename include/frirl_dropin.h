/*
 * frirl_dropin.h -- the complete public interface of the MI355X drop-in libfive / libfrirl replacement.
 *
 * One consolidated header: build configuration, struct FIVERB and the FIVE API, the FRIRL agent types, defaults
 * and API, rule-base I/O, application helper macros.  The header NAMES the reference's applications include
 * (FIVE.h, frirl.h, frirl_types.h, frirl_types_def.h, frirl_utils.h, frirl_app_helpers.h, frirl_test.h, config.h)
 * exist next to this file as one-line forwarders, so unmodified reference sources compile against it.
 *
 * The struct layouts, field names, function signatures and macro names below ARE the reference's ABI / API
 * (src/five/FIVE.h:24-103, src/frirl/frirl_types.h:22-172, frirl_types_def.h:22-77, frirl.h:42-64,
 * frirl_utils.h:19-25, frirl_app_helpers.h:25-46, config.h.in:1-74): they have to coincide for programs to
 * link unchanged (sizes and offsets are asserted in tests/test_dropin.py against SURVEY Appendix A).  Everything
 * behind them is new code.
 */
#ifndef FRIRL_DROPIN_H
#define FRIRL_DROPIN_H

/* ===== build configuration ========================================================================= */
#define FIVE_FIXRES
#define FIVE_NOINF
#define FIVE_NONAN
#define FRIRL_FAST
#define DOUBLE_PRECISION
#define FAST_ABS
#define FAST_POW
#define FAST_SQRT
#define BUILD_CHECK_STATES
#define BUILD_HIP_GFX950          /* the hot path runs on the MI355X; there is no CPU fallback */
/* not defined: DEBUG, PREDICT_BRANCHES, BUILD_OPENMP, BUILD_MPI, BUILD_AVX2, BUILD_VISUALIZATION */

typedef double fri_float;         /* reference config.h.in:17-21 (DOUBLE_PRECISION) */

#include <stdlib.h>

#define DEBUG_MSG(...) {};
#define SIMD_ALIGN
#define MALLOC(size) malloc(size)
#define likely(x)   x
#define unlikely(x) x

#define FRIRL_AGENT_EPCHUNK 10    /* reference config.h.in:68 */
#define CHECK_STATES 1            /* reference config.h.in:70-74 */

/* ===== FIVE engine ================================================================================== */
#define FIVE_MAX_NUM_OF_UNIVERSES  8


struct FIVERB {
	double *u;                      /* universes, numofunivs rows of univlength points             */
	double *ve;                     /* vague environments, same shape                              */
	double *psc;                    /* scaling points (unused by FRIRL)                            */
	double *scf;                    /* scaling functions (unused by FRIRL)                         */
	int numofunivs;                 /* antecedent universes (states + action)                      */
	int univlength;                 /* points per universe                                         */
	int numofrules;                 /* rules currently in the base                                 */
	int maxnumofrules;              /* capacity                                                    */
	int rulelength;                 /* antecedents + 1 consequent                                  */
	int numofantecedents;           /* rulelength - 1                                              */
	int p;                          /* Shepard power                                               */

	double *valvagp;                /* FIVEValVag output point                                     */
	double *valvagu;                /* consequent universe row                                     */
	double *valvagve;               /* consequent VE row                                           */
	double valvagdims;              /* 1                                                           */
	double *ruledists;              /* [maxnumofrules] distances of the last observation           */
	double *rant;                   /* [maxnumofrules][numofantecedents] raw antecedents (AoS)     */
	double **rseqant;               /* per dimension: [maxnumofrules] raw antecedents (SoA)        */
	double *ract;                   /* rseqant[last]                                               */
	unsigned int *rant_uindex;      /* AoS universe indices of the snapped antecedents             */
	unsigned int *rant_veval;       /* AoS VE values (storage is double, declared as in the reference) */
	unsigned int **rseqant_uindex;  /* SoA universe indices                                        */
	double **rseqant_veval;         /* SoA VE values -- what the distance kernels stream           */
	unsigned int *ract_uindex;      /* rseqant_uindex[last]                                        */
	double *ract_veval;             /* rseqant_veval[last]                                         */
	double *rconc;                  /* [maxnumofrules] consequents (Q values)                      */
	double *weights;                /* [maxnumofrules] normalised Shepard weights                  */
	unsigned int uksize;            /* univlength - 1                                              */
	double *ukdomains;              /* u_last - u_first per universe                               */
	double *udivs;                  /* step per universe                                           */
	double **uk;                    /* row pointers into u                                         */
	double **vek;                   /* row pointers into ve                                        */
	double *wi;                     /* scratch                                                     */
	double *frd_dists;              /* scratch (unused by the HIP path, kept for layout)           */
	double *fvc_vagdist;            /* scratch                                                     */
	double *newrant;                /* where the next rule's antecedents go                        */
	double *newrconc;               /* where the next rule's consequent goes                       */
	unsigned int epno;
	unsigned int avx2_rbsize;       /* ceil(numofrules / 4): kept so sizeof matches the reference's default build */
};

/* rc5 API */
struct FIVERB *FIVEInit(double *u, double *ve, int p, int numofunivs, int univlength, int numofrules, int maxnumofrules, int rulelength, double *rant, double *rconc);
double *FIVEGScFunc(double *u, int numofunivs, int univlength, double *psc, int mp, int np, double nls);
double *FIVEGVagEnv(double *u, int numofunivs, int univlength, double *scf);
int FIVEValVag(struct FIVERB *frb, double *vp);
double FIVEVagConcl(struct FIVERB *frb, double *x);
unsigned int FIVEVagConclWeight(struct FIVERB *frb, double *x);
double FIVEVagConcl_FRIRL_BestAct(struct FIVERB *frb, double *ruledists);
int FIVEAddRule(struct FIVERB *frb, double *newrule);

/* old API */
int five_vague_distance(struct FIVERB *frb, fri_float *p1, fri_float *p2, fri_float *d);
int five_vague_distance_parallel(struct FIVERB *frb, fri_float *p1, int p1_offset, fri_float *p2, fri_float *d);
int five_rule_distance(struct FIVERB *frb, fri_float *x);
int five_add_rule(struct FIVERB *frb, fri_float *ruletoadd);
int five_remove_rule(struct FIVERB *frb, unsigned int rulenotoremove);
void five_deinit(struct FIVERB *frb);

/* current API */
int FIVE_add_rule(struct FIVERB *frb, fri_float *rant, fri_float rconc);
unsigned int FIVE_vag_concl_weight(struct FIVERB *frb, double *ant, double *weights);
unsigned int FIVE_vag_concl(struct FIVERB *frb, double *ant, double *conc);
int FIVE_GSc_func(double *u, int numofunivs, int univlength, double *psc, int mp, int np, double nls, double *scf);

/* ===== FRIRL agent types ============================================================================ */
enum frirl_runmode { FRIRL_SEQ, FRIRL_OMP, FRIRL_MPI, FRIRL_TEST };
enum frirl_reduction_strategy { FRIRL_REDUCTION_STRATEGY_NOREDUCE, FRIRL_REDUCTION_STRATEGY_DEFAULT };

/* one state dimension or the action dimension */
struct frirl_dimension_desc {
    int values_len;          /* number of allowed (grid) values   */
    fri_float *values;       /* the allowed values                */
    fri_float values_div;    /* their spacing                     */
    fri_float values_steep;  /* scaling-function steepness        */
    fri_float values_def;    /* episode start value               */
    int universe_len;        /* universe resolution               */
    fri_float *universe;     /* universe points                   */
    fri_float universe_div;  /* universe step                     */
};

/* possible rule places of one dimension */
struct frirl_values_desc {
    int values_len;
    fri_float *values;
    fri_float *vevalues;     /* VE value of each action (action dimension only) */
    fri_float epsilon;
};

struct frirl_reward_desc {
    fri_float value;
    fri_float ep_total_value;
    int ep_total_steps;
    int success;
};

struct FIVERB;

struct frirl_desc {
    int argc;
    char **argv;
    int runmode;
    char *rbfile;
    int visualization;
    int gui_width;
    int gui_height;
    int verbose;
    int agent_rnd_init;

    struct frirl_dimension_desc actiondim;
    int statedims_len;
    struct frirl_dimension_desc *statedims;

    fri_float alpha;
    fri_float gamma;
    fri_float epsilon;
    fri_float qdiff_pos_boundary;
    fri_float qdiff_neg_boundary;
    fri_float qdiff_final_tolerance;
    fri_float reward_good_above;
    fri_float rule_weight_considered_significant_for_update;
    fri_float reduction_reward_tolerance;
    unsigned char skip_rules;
    unsigned char no_random;
    unsigned char construct_rb;
    unsigned char reduce_rb;
    unsigned char reduction_strategy;
    int max_episodes;
    int max_steps;
    unsigned int episode_num;
    int five_maxnumofrules;

    /* environment callbacks (host functions, exactly as in the reference) */
    void (* get_reward_func)(struct frirl_desc *frirl, fri_float *states, int states_len, struct frirl_reward_desc *reward);
    void (* do_action_func) (struct frirl_desc *frirl, fri_float action, fri_float *states, int states_len, fri_float *new_states);
    void (* quant_obs_func) (struct frirl_desc *frirl, fri_float *states, int states_len, fri_float *new_states);
    void (* draw_func) (struct frirl_desc *frirl, fri_float *new_curr_state, double action, unsigned int steps);

    /* private */
    struct FIVERB *fiverb;
    double *fiverb_ua;
    double *fiverb_vea;
    unsigned int reduction_state;
    int numofantecedents;
    struct frirl_values_desc *possible_states;
    struct frirl_values_desc *possible_actions;
    struct frirl_reward_desc reward;

    fri_float *fgba_vagdist_states;
    fri_float *fgba_ruledist;
    fri_float *fgba_actconc;
    fri_float *fgba_dists;
    fri_float *fgba_statedistsum;
    fri_float *fus_proposed_values;
    fri_float *fus_values;
    fri_float *fus_check_states;
    fri_float fus_is_rule_inserted;
    fri_float *fep_ant;
    fri_float *fep_cur_ant;
    fri_float *fep_q_ant;
    fri_float *fep_cur_q_ant;

    int is_running;
    int agent_id;
    int agent_world_size;
    int epended;

    unsigned int keyaction;
    int valid_simulation;
    int original_learning;
    int user_exited;
};

#define FRIRL frirl_desc

/* default agent parameters (reference src/frirl/frirl_types_def.h:22-77); frirl_types_def.h instantiates them */
#define FRIRL_DESC_DEFAULT_INITIALIZER { \
    .argc = 1, .argv = 0, .runmode = FRIRL_SEQ, .rbfile = 0, \
    .visualization = 0, .gui_width = 640, .gui_height = 480, .verbose = 1, .agent_rnd_init = 1, \
    .alpha = 0.5, .gamma = 1.0, .epsilon = 0.001, \
    .qdiff_pos_boundary = 1.0, .qdiff_neg_boundary = -250.0, .qdiff_final_tolerance = 250.0, \
    .reward_good_above = 0.0, \
    .rule_weight_considered_significant_for_update = 0.05, \
    .reduction_reward_tolerance = 0.0, \
    .skip_rules = 0, .no_random = 1, .construct_rb = 1, .reduce_rb = 0, \
    .reduction_strategy = FRIRL_REDUCTION_STRATEGY_DEFAULT, \
    .max_episodes = 1000, .max_steps = 1000, \
    .five_maxnumofrules = 16384, \
    .get_reward_func = 0, .do_action_func = 0, .quant_obs_func = 0, .draw_func = 0, \
    .numofantecedents = 0, .reduction_state = 0, .statedims_len = 0, \
    .fus_is_rule_inserted = 0, \
    .is_running = 0, .epended = 0, .agent_id = 0, \
    .keyaction = -1, .valid_simulation = 0, .original_learning = 1, .user_exited = 0, \
}

/* ===== rule-base I/O and command line =============================================================== */
void frirl_show_rb(struct frirl_desc *frirl);
void frirl_show_hex_rb(struct frirl_desc *frirl);
int frirl_save_rb_to_text_file(struct frirl_desc *frirl, const char *file_name);
int frirl_save_rb_to_bin_file(struct frirl_desc *frirl, const char *file_name);
int frirl_load_rb_from_bin_file(struct frirl_desc *frirl, const char *file_name);
void frirl_print_usage();
void frirl_parse_cmdline(struct frirl_desc *frirl, int argc, char **argv);

/* ===== agent API ==================================================================================== */
#define TERM_RED   "\033[0;31m"
#define TERM_GREEN "\033[0;32m"
#define TERM_NC    "\033[0m"

int frirl_init(struct frirl_desc *frirl);
void frirl_deinit(struct frirl_desc *frirl);
int frirl_init_ve(struct frirl_desc *frirl, fri_float *ve, fri_float *u, int univlength);
int frirl_init_rb(struct frirl_desc *frirl, fri_float *rant, fri_float *rconc, int *numofrules);
void frirl_episode(struct frirl_desc *frirl);
unsigned int frirl_e_greedy_selection(struct frirl_desc *frirl, fri_float *states);
unsigned int frirl_get_best_action(struct frirl_desc *frirl, fri_float *states);
fri_float frirl_check_possible_states(struct frirl_desc *frirl, fri_float observation, struct frirl_values_desc *possible_states);
void frirl_sequential_run(struct frirl_desc *frirl);
void frirl_omp_run(struct frirl_desc *frirl);
void frirl_mpi_run(struct frirl_desc *frirl);
void frirl_update_sarsa(struct frirl_desc *frirl, fri_float *q_ant, fri_float reward, fri_float *cur_q_ant);

/* ===== application helpers ========================================================================== */
/* stack storage for a dimension's value grid / universe, sized by the descriptor */
#define FRIRL_ALLOC_VALUES(x)   fri_float _local_ ## x ## _values[x.values_len];     x.values   = _local_ ## x ## _values;
#define FRIRL_ALLOC_UNIVERSE(x) fri_float _local_ ## x ## _universe[x.universe_len]; x.universe = _local_ ## x ## _universe;
#define FRIRL_ALLOC_DIM(x)  FRIRL_ALLOC_VALUES(x);  FRIRL_ALLOC_UNIVERSE(x);

/* fill them with a symmetric fixed-step grid */
#define FRIRL_GEN_FIXRES_UNIVERSE(x) frirl_gen_fixres_arr(x.universe, x.universe_len, x.universe_div);
#define FRIRL_GEN_FIXRES_VALUES(x)   frirl_gen_fixres_arr(x.values,   x.values_len,   x.values_div);
#define FRIRL_GEN_FIXRES_DIM(x)  FRIRL_GEN_FIXRES_VALUES(x);  FRIRL_GEN_FIXRES_UNIVERSE(x);

void frirl_gen_fixres_arr(fri_float *arr, int len, fri_float div);
void frirl_run(struct frirl_desc *frirl, int verbose);
void frirl_visualization_init(struct frirl_desc *frirl);
void frirl_visualization_deinit();

/* evaluation-only run mode (reference src/frirl/frirl_test.h:18) */
void frirl_test_run(struct frirl_desc *frirl);

#endif /* FRIRL_DROPIN_H */
