/*
 * config.h -- build configuration of the MI355X drop-in libfive / libfrirl replacement.
 *
 * Plays the role of the reference's CMake-generated config.h (template: reference config.h.in:1-74)
 * for the one configuration that is the parity target: the reference's default option set
 * (CMakeLists.txt:9-24) minus visualisation.  The feature macros are kept because the reference's
 * own example sources test some of them; the arithmetic they select in the reference (fixed
 * resolution, no NaN / Inf rules, FRIRL fast path, double precision) is what the HIP kernels implement.
 */
#ifndef FRIRL_DROPIN_CONFIG_H
#define FRIRL_DROPIN_CONFIG_H

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

#endif
