/*
 * frirl_demo.h -- the three demo environments of the reference (examples/mountaincar, cartpole, acrobot)
 * packaged in the drop-in library: descriptors + host callbacks, so that drivers and bindings on machines
 * without the reference tree can run them.  Additive API (not in the reference).
 */
#ifndef FRIRL_DEMO_H
#define FRIRL_DEMO_H

#include "frirl_types.h"

int frirl_demo_setup(struct frirl_desc *fr, const char *env);     /* "mountaincar" | "cartpole" | "acrobot" */
void frirl_demo_release(struct frirl_desc *fr);
/* flat description for bindings; u/ve [ (nstates+1) * U ], grid [ (nstates+1) * 64 ]; pass u = ve = NULL to query sizes */
int frirl_demo_describe(const char *env, int *nstates, int *U, int *A, double *u, double *ve, double *grid, int *grid_len,
                        double *grid_div, double *values_def, double *action_ve, double *hparams, int *max_steps);

/* Many independent agents of one demo on the GPU (frirl_hip_batch_*): every agent starts from the 2^nant corner
 * rule base and learns until its rule base is complete (at most max_episodes-1 episodes).  Prints one summary line;
 * when out_txt != NULL the rule base of agent 0 is written there in the reference's text format.
 * Returns the number of converged agents, or -1 on error. */
int frirl_demo_batch_run(const char *env, int agents, int max_episodes, const char *out_txt, int verbose);
/* the same, followed by the rule-base reduction (strategy 1 | 2, frirl_sequential_run.c:170-350) of agent 0's rule base on the
 * GPU (frirl_hip_batch_reduce); out_txt then holds the REDUCED rule base.  reduce_strategy 0 = frirl_demo_batch_run. */
int frirl_demo_batch_run_reduce(const char *env, int agents, int max_episodes, int reduce_strategy, const char *out_txt, int verbose);
/* the same; load_bin != NULL: instead of learning, every agent gets a rule base from that .frirlrb.bin file (written by the
 * reference, by the drop-in library or by frirl_hip_batch_save_rulebases) and replays one greedy episode on it
 * (frirl_test_run), then the optional reduction */
int frirl_demo_batch_run_ex(const char *env, int agents, int max_episodes, int reduce_strategy, const char *load_bin, const char *save_bin,
                            const char *out_txt, int verbose);   /* save_bin != NULL: all agents' rule bases (before the reduction) go there */

/* `agents` agents over `gpus` devices of this node (0 = every visible device) through frirl_hip_multi_* (one batch + host thread per
 * device, per-episode report all-reduced with RCCL); out_txt = rule base of global agent 0.  Returns the converged agents or -1. */
/* many agents WITH the reference's rule-base exchange (frirl_omp_run: chunks of 9 episodes, merge_rb in both directions after each,
 * start states from gen_def_states) on one GPU; out_txt = the master's rule base.  Returns the number of merge rounds, or -1. */
int frirl_demo_merged_run(const char *env, int agents, int max_episodes, const char *out_txt, int verbose);
/* the same with the agents sharded over `gpus` devices (0 = all) and the rule-base exchange over RCCL (frirl_hip_multi_train_merged) */
int frirl_demo_multi_merged_run(const char *env, int agents, int gpus, int max_episodes, const char *out_txt, int verbose);
int frirl_demo_multi_run(const char *env, int agents, int gpus, int max_episodes, const char *out_txt, int verbose);

#endif
