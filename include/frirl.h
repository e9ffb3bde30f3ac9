/*
 * frirl.h -- FRIRL agent API, MI355X drop-in edition (reference src/frirl/frirl.h:42-64).
 * Same functions, same semantics; frirl_get_best_action and frirl_update_sarsa run on the GPU.
 */
#ifndef _FRIRL_H
#define _FRIRL_H

#include "config.h"
#include "frirl_types.h"
#include "frirl_utils.h"
#include "FIVE.h"

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

#endif /* _FRIRL_H */
