/*
 * frirl_app_helpers.h -- helpers applications use to describe their dimensions and to start a run
 * (reference src/frirl/frirl_app_helpers.h:25-46; the macro names and expansions are part of the
 * application-facing API, so they are kept).
 */
#ifndef FRIRL_APP_HELPERS_H
#define FRIRL_APP_HELPERS_H

#include "frirl_types.h"
#include "FIVE.h"
#include "frirl.h"

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

#endif /* FRIRL_APP_HELPERS_H */
