/*
 * frirl_types.h -- data types of the FRIRL agent, MI355X drop-in edition.
 *
 * Layout-compatible with the reference's src/frirl/frirl_types.h:22-172 (struct sizes 64 / 32 / 24 /
 * 472 bytes): applications fill a struct frirl_desc exactly as with the
 * reference (designated initialisers of frirl_desc_default in frirl_types_def.h).
 */
#ifndef FRIRL_TYPES_H
#define FRIRL_TYPES_H

#include "config.h"

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

#endif /* FRIRL_TYPES_H */
