/* dropin_internal.h -- shared by the ANSI-C drop-in sources (not installed). */
#ifndef FRIRL_DROPIN_INTERNAL_H
#define FRIRL_DROPIN_INTERNAL_H

#include "FIVE.h"
#include "frirl.h"
#include "frirl_hip.h"

/* device mirror registered for a FIVERB (struct layout has no spare slot: side table) */
five_hip_mirror *five_dropin_mirror(struct FIVERB *frb);
/* nearest universe index, fixed step (reference src/inl/min.inl:71-92) */
unsigned int five_dropin_snap(const double *universe, int len, double point, double div);
/* abort with the HIP layer's message: the hot path has no CPU fallback */
void five_dropin_fatal(const char *where, int rc);

#endif
