/*
 * frirl_hip.h -- C ABI of the MI355X (gfx950) FRIRL / FIVE hot path.
 *
 * This is the drop-in boundary: a plain-C shared library (libfrirl_hip.so) whose entry points are
 * what a host program written against the reference's `five_*` / `FIVE_*` / `frirl_*` API binds.
 * The reference has no FFI layer -- its boundary is the C ABI of libfive.a / libfrirl.a
 * (reference src/five/FIVE.h:79-103, src/frirl/frirl.h:42-64) -- so every entry point below names
 * the reference function it replaces.  All entry points are BATCHED over E independent rule
 * bases ("one FIVERB per agent", reference src/frirl/frirl_agent.c:229-238); E = 1 is the
 * reference's single-agent call.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - every pointer marked [dev] is a device (HBM) pointer; the library never allocates or frees
 *    caller data and never synchronises the stream: results are ready when `stream` reaches the
 *    point after the call (hipStreamSynchronize / event).  `stream` is a hipStream_t passed as
 *    void* (NULL = the default stream).
 *  - return value: 0 on success, a negative FRIRL_HIP_E* code on error; frirl_hip_last_error()
 *    returns a static message for the calling thread.  Nothing falls back to the CPU: without a
 *    usable gfx950 device every compute entry point returns FRIRL_HIP_ENODEV.
 *  - "no exact hit" is 0xFFFFFFFF (== the reference's `~0`, five_rule_distance.c:294) in uint32
 *    outputs.
 *  - arithmetic is IEEE binary64 with separate multiply and add (no FMA contraction), IEEE sqrt
 *    and divide, dimension-ordered sums: rule distances and hit indices are bit-identical to the
 *    reference's AVX2/C path.  Shepard sums are reduced in a fixed lane-strided tree (run-to-run
 *    deterministic) and use a plain-double power where the reference keeps an x87 long double
 *    (src/inl/fast_pow.inl:21-25): interpolated Q values agree to <= 1e-6 relative (measured
 *    ~1e-13), never bit-exactly.
 *
 * Rule-base layout in HBM (struct frirl_hip_rulebases): per environment e one slab
 *      rb[e][k][r]   k = 0..nant-1 : VE value of antecedent k of rule r (reference
 *                                    FIVERB.rseqant_veval[k][r], src/five/FIVE.h:56)
 *      rb[e][nant][r]              : consequent Q of rule r (FIVERB.rconc[r], FIVE.h:59)
 *  i.e. a structure of arrays with row stride maxR doubles and slab stride (nant+1)*maxR doubles;
 *  maxR must be a multiple of 2 and the base 16-byte aligned (16-byte vector loads).  Rows are
 *  zero beyond nrules[e].
 */
#ifndef FRIRL_HIP_H
#define FRIRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRIRL_HIP_MAX_NANT     16   /* reference: FIVE_MAX_NUM_OF_UNIVERSES 8 (src/five/FIVE.h:19) */
#define FRIRL_HIP_MAX_ACTIONS  32
#define FRIRL_HIP_NO_HIT       0xFFFFFFFFu

#define FRIRL_HIP_OK        0
#define FRIRL_HIP_ENODEV   -1   /* no gfx950 device / HIP runtime unusable */
#define FRIRL_HIP_EINVAL   -2   /* bad argument (shape, alignment, NULL)   */
#define FRIRL_HIP_ELAUNCH  -3   /* kernel launch or runtime call failed    */

/* Universes and vague environments shared by every rule base of a batch
 * (reference FIVERB.u / .ve / .udivs, src/five/FIVE.h:25-26,64; built by frirl_init_ve.c:25-121). */
typedef struct frirl_hip_tables {
    int32_t nant;            /* numofunivs: state dims + 1 action dim                   */
    int32_t U;               /* univlength                                              */
    const double *u;         /* [dev] [nant][U] universes, fixed step, increasing        */
    const double *ve;        /* [dev] [nant][U] vague environments                       */
} frirl_hip_tables;

/* E independent rule bases (see layout above). */
typedef struct frirl_hip_rulebases {
    int32_t E;               /* number of environments / agents in the batch            */
    int32_t maxR;            /* capacity per rule base (FIVERB.maxnumofrules)           */
    double *rb;              /* [dev] [E][nant+1][maxR]                                 */
    int32_t *nrules;         /* [dev] [E] FIVERB.numofrules                             */
} frirl_hip_rulebases;

/* ---- runtime ------------------------------------------------------------------------------- */
const char *frirl_hip_version(void);
const char *frirl_hip_last_error(void);
int frirl_hip_device_count(void);                 /* number of visible gfx950 devices, <0 on error */
int frirl_hip_device_info(int device, char *name, int name_len, int32_t *cus, int64_t *hbm_bytes);

/* ---- five_rule_distance (reference src/five/five_rule_distance.c:63-295) -------------------
 * For every environment e: ruledists[e][r] = sqrt(sum_k (ve[k][snap(x[e][k])] - rb[e][k][r])^2),
 * k ascending, for r < nrules[e]; hit[e] = lowest r < nrules[e] with distance exactly 0.0, else
 * FRIRL_HIP_NO_HIT.  snap() is the reference's fixed-step nearest-index rule
 * (src/inl/min.inl:71-92).  ruledists may be NULL (index-only form); entries r >= nrules[e] are
 * not written.
 *   x         [dev] [E][nant]   observations
 *   ruledists [dev] [E][maxR]   or NULL
 *   hit       [dev] [E]         uint32
 */
int five_hip_rule_distance(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x,
                           double *ruledists, uint32_t *hit, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRIRL_HIP_H */
