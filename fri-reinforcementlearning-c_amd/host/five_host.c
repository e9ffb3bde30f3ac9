/*
 * five_host.c -- host side of the FIVE engine (ANSI C), MI355X drop-in for the reference's libfive.
 *
 * Exports the reference's complete FIVE API (include/FIVE.h == reference src/five/FIVE.h:79-103).
 * Init-time table generation and rule-base bookkeeping stay on the host, as in the reference; every
 * per-observation computation (rule distance, conclusion, weights, best-action conclusion) is one call
 * into the C-ABI HIP layer (include/frirl_hip.h, five_hip_mirror_*) operating on a device mirror of
 * the rule base.  There is no CPU fallback: if the HIP layer fails the process exits with its message
 * (the reference's own error convention is printf + exit, e.g. FIVEInit.c:61-64).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dropin_internal.h"

/* ---- FIVERB -> mirror side table ----------------------------------------------------------
 * The only state shared between rule bases; guarded, so that distinct FIVERB / frirl_desc instances can be driven from
 * different threads (the reference's frirl_omp_run model: one agent per thread, frirl_agent.c:309-325). */
#define MAX_MIRRORS 1024
static struct { struct FIVERB *frb; five_hip_mirror *m; } g_mirrors[MAX_MIRRORS];
static pthread_mutex_t g_mirrors_lock = PTHREAD_MUTEX_INITIALIZER;

five_hip_mirror *five_dropin_mirror(struct FIVERB *frb)
{
    int i;
    five_hip_mirror *m = NULL;
    pthread_mutex_lock(&g_mirrors_lock);
    for (i = 0; i < MAX_MIRRORS; i++) if (g_mirrors[i].frb == frb) { m = g_mirrors[i].m; break; }
    pthread_mutex_unlock(&g_mirrors_lock);
    if (m) return m;
    fprintf(stderr, "FIVE: rule base %p was not created by FIVEInit of this library\n", (void *)frb);
    exit(30);
}

static void mirror_register(struct FIVERB *frb, five_hip_mirror *m)
{
    int i;
    pthread_mutex_lock(&g_mirrors_lock);
    for (i = 0; i < MAX_MIRRORS; i++) if (!g_mirrors[i].frb) { g_mirrors[i].frb = frb; g_mirrors[i].m = m; break; }
    pthread_mutex_unlock(&g_mirrors_lock);
    if (i < MAX_MIRRORS) return;
    fprintf(stderr, "FIVE: too many live rule bases (%d)\n", MAX_MIRRORS);
    exit(31);
}

static void mirror_unregister(struct FIVERB *frb)
{
    int i;
    five_hip_mirror *m = NULL;
    pthread_mutex_lock(&g_mirrors_lock);
    for (i = 0; i < MAX_MIRRORS; i++) if (g_mirrors[i].frb == frb) { m = g_mirrors[i].m; g_mirrors[i].frb = 0; g_mirrors[i].m = 0; break; }
    pthread_mutex_unlock(&g_mirrors_lock);
    if (m) five_hip_mirror_destroy(m);
}

void five_dropin_fatal(const char *where, int rc)
{
    fprintf(stderr, "%s: MI355X hot path failed (rc=%d): %s\n", where, rc, frirl_hip_last_error());
    exit(32);
}

/* reference src/inl/min.inl:71-92; the stray read one past the row is guarded */
unsigned int five_dropin_snap(const double *universe, int len, double point, double div)
{
    int low = (int)((point - universe[0]) / div);
    double d1, d2;
    if (low < 0) return 0;
    if (low >= len) return (unsigned int)(len - 1);
    if (low + 1 >= len) return (unsigned int)low;
    d1 = fabs(universe[low] - point);
    d2 = fabs(universe[low + 1] - point);
    return (d1 <= d2) ? (unsigned int)low : (unsigned int)(low + 1);
}

/* ---- scaling function / vague environment (init time, host) -------------------------------- */

/* reference src/five/FIVEGScFunc.c:76-255.  nls only has to be positive or NaN; the reference then
 * forces its exponent to NaN (:84-92), so non-constant segments are always interpolated linearly. */
int FIVE_GSc_func(double *u, int numofunivs, int univlength, double *psc, int mp, int np, double nls, double *scf)
{
    int i, j, s;
    if (nls <= 0) return -1;
    if (mp > 1) {
        for (i = 0; i < numofunivs; i++) {
            for (j = 0; j < univlength; j++) {
                const double x = u[i * univlength + j];
                double *out = scf + i * univlength + j;
                const double *last = psc + (mp - 1) * np;
                if (x < psc[0]) *out = psc[1];
                for (s = 0; s + 1 < mp; s++) {
                    const double *a = psc + s * np, *b = a + np;
                    if (x >= a[0] && x < b[0]) {
                        const double sa = (np == 2) ? a[1] : a[2], sb = b[1];
                        if (sa == sb) *out = sa;
                        else *out = ((sb - sa) / (b[0] - a[0])) * (x - a[0]) + sa;
                    }
                }
                if (x >= last[0]) {
                    if (np == 2) *out = last[1];
                    else if (j == univlength - 1 && x == last[0]) *out = last[1];
                    else *out = last[2];
                }
            }
        }
        return 0;
    }
    if (np == 1) {
        for (i = 0; i < numofunivs * univlength; i++) scf[i] = psc[0];
        return 0;
    }
    for (i = 0; i < numofunivs; i++)
        for (j = 0; j < univlength; j++) {
            if (u[i * univlength + j] < psc[0]) scf[i * univlength + j] = psc[1];
            else scf[i * univlength + j] = (np == 2) ? psc[i * np + 1] : psc[i * np + 2];
        }
    return 0;
}

double *FIVEGScFunc(double *u, int numofunivs, int univlength, double *psc, int mp, int np, double nls)
{
    double *scf = MALLOC(sizeof(double) * numofunivs * univlength);
    if (!scf) return NULL;
    if (FIVE_GSc_func(u, numofunivs, univlength, psc, mp, np, nls, scf) != 0) { free(scf); return NULL; }
    return scf;
}

/* reference src/five/FIVEGVagEnv.c:40-102: cumulative trapezoid integral of the scaling function */
double *FIVEGVagEnv(double *u, int numofunivs, int univlength, double *scf)
{
    int k, j, inf;
    double *ve = MALLOC(sizeof(double) * numofunivs * univlength);
    if (!ve) return NULL;
    for (k = 0; k < numofunivs; k++) {
        const double *uk = u + k * univlength, *sk = scf + k * univlength;
        double *vk = ve + k * univlength;
        inf = 0;
        for (j = 0; j < univlength; j++) if (sk[j] == INFINITY) { inf = 1; break; }
        vk[0] = inf ? -1 : 0;
        for (j = 0; j + 1 < univlength; j++) {
            const double area = (uk[j + 1] - uk[j]) * (sk[j] + sk[j + 1]) * 0.5;
            vk[j + 1] = inf ? area : vk[j] + area;
        }
    }
    return ve;
}

/* ---- rule-base container -------------------------------------------------------------------- */

static void snap_rule_into(struct FIVERB *frb, int r, const double *rant)
{
    int k;
    const int n = frb->numofantecedents;
    for (k = 0; k < n; k++) {
        const unsigned int j = five_dropin_snap(frb->uk[k], frb->univlength, rant[k], frb->udivs[k]);
        const double v = frb->vek[k][j];
        frb->rant[r * n + k] = rant[k];
        frb->rseqant[k][r] = rant[k];
        frb->rseqant_uindex[k][r] = j;
        frb->rant_uindex[r * n + k] = j;
        frb->rseqant_veval[k][r] = v;
        ((double *)frb->rant_veval)[r * n + k] = v;      /* declared unsigned*, used as double storage (FIVEInit.c:146,265) */
    }
}

/* reference src/five/FIVEInit.c:55-347.  u, ve, rant, rconc are borrowed (owned by the caller), the rest
 * is owned here.  Additionally creates the device mirror and uploads the initial rules. */
struct FIVERB *FIVEInit(double *u, double *ve, int p, int numofunivs, int univlength, int numofrules, int maxnumofrules, int rulelength,
                        double *rant, double *rconc)
{
    struct FIVERB *frb;
    five_hip_mirror *m;
    int k, r, n = rulelength - 1, rc;
    if (numofunivs > FIVE_MAX_NUM_OF_UNIVERSES) {
        printf("FIVEInit: fatal error: given number of universes greater than FIVE_MAX_NUM_OF_UNIVERSES (%d)\n", FIVE_MAX_NUM_OF_UNIVERSES);
        exit(1);
    }
    if (numofrules > maxnumofrules || n != numofunivs) {
        printf("FIVEInit: fatal error: inconsistent sizes (rules %d/%d, antecedents %d, universes %d)\n", numofrules, maxnumofrules, n, numofunivs);
        exit(1);
    }
    frb = calloc(1, sizeof(*frb));
    if (!frb) return NULL;
    frb->u = u; frb->ve = ve;
    frb->numofunivs = numofunivs; frb->univlength = univlength;
    frb->numofrules = numofrules; frb->maxnumofrules = maxnumofrules;
    frb->rulelength = rulelength; frb->numofantecedents = n;
    frb->p = (p == 0) ? n : p;
    frb->avx2_rbsize = (unsigned int)((numofrules + 3) / 4);
    frb->uksize = (unsigned int)(univlength - 1);
    frb->rant = rant; frb->rconc = rconc;

    frb->uk = MALLOC(sizeof(double *) * numofunivs);
    frb->vek = MALLOC(sizeof(double *) * numofunivs);
    frb->ukdomains = MALLOC(sizeof(double) * numofunivs);
    frb->udivs = MALLOC(sizeof(double) * numofunivs);
    frb->ruledists = calloc(maxnumofrules + 4, sizeof(double));
    frb->weights = calloc(maxnumofrules + 4, sizeof(double));
    frb->wi = calloc(maxnumofrules + 4, sizeof(double));
    frb->frd_dists = calloc(4 * FIVE_MAX_NUM_OF_UNIVERSES, sizeof(double));
    frb->fvc_vagdist = calloc(numofunivs, sizeof(double));
    frb->valvagp = calloc(numofunivs, sizeof(double));
    frb->rant_uindex = calloc((size_t)maxnumofrules * n, sizeof(unsigned int));
    frb->rant_veval = calloc((size_t)maxnumofrules * n, sizeof(double));
    frb->rseqant = MALLOC(sizeof(double *) * n);
    frb->rseqant_uindex = MALLOC(sizeof(unsigned int *) * n);
    frb->rseqant_veval = MALLOC(sizeof(double *) * n);
    if (!frb->uk || !frb->vek || !frb->ukdomains || !frb->udivs || !frb->ruledists || !frb->weights || !frb->wi || !frb->frd_dists ||
        !frb->fvc_vagdist || !frb->valvagp || !frb->rant_uindex || !frb->rant_veval || !frb->rseqant || !frb->rseqant_uindex || !frb->rseqant_veval)
        return NULL;
    for (k = 0; k < numofunivs; k++) {
        frb->uk[k] = u + k * univlength;
        frb->vek[k] = ve + k * univlength;
        frb->ukdomains[k] = frb->uk[k][univlength - 1] - frb->uk[k][0];
        frb->udivs[k] = frb->ukdomains[k] / (frb->uksize);
    }
    for (k = 0; k < n; k++) {
        frb->rseqant[k] = calloc(maxnumofrules, sizeof(double));
        frb->rseqant_uindex[k] = calloc(maxnumofrules, sizeof(unsigned int));
        frb->rseqant_veval[k] = calloc(maxnumofrules, sizeof(double));
        if (!frb->rseqant[k] || !frb->rseqant_uindex[k] || !frb->rseqant_veval[k]) return NULL;
    }
    frb->ract = frb->rseqant[n - 1];
    frb->ract_uindex = frb->rseqant_uindex[n - 1];
    frb->ract_veval = frb->rseqant_veval[n - 1];
    frb->valvagu = frb->uk[numofunivs - 1];
    frb->valvagve = frb->vek[numofunivs - 1];
    frb->valvagdims = 1;
    for (r = 0; r < numofrules; r++) snap_rule_into(frb, r, rant + (size_t)r * n);
    frb->newrconc = rconc + numofrules;
    frb->newrant = rant + (size_t)numofrules * n;

    m = five_hip_mirror_create(numofunivs, univlength, u, ve, maxnumofrules, frb->p);
    if (!m) five_dropin_fatal("FIVEInit", FRIRL_HIP_ENODEV);
    rc = five_hip_mirror_upload(m, numofrules, (const double *const *)frb->rseqant_veval, rconc);
    if (rc) five_dropin_fatal("FIVEInit(upload)", rc);
    mirror_register(frb, m);
    return frb;
}

/* reference src/five/five_deinit.c:22-51 (u, ve, rant, rconc belong to the caller) */
void five_deinit(struct FIVERB *frb)
{
    int k;
    if (!frb) return;
    mirror_unregister(frb);
    for (k = 0; k < frb->numofantecedents; k++) { free(frb->rseqant[k]); free(frb->rseqant_uindex[k]); free(frb->rseqant_veval[k]); }
    free(frb->rseqant); free(frb->rseqant_uindex); free(frb->rseqant_veval);
    free(frb->rant_uindex); free(frb->rant_veval);
    free(frb->uk); free(frb->vek); free(frb->ukdomains); free(frb->udivs);
    free(frb->ruledists); free(frb->weights); free(frb->wi); free(frb->frd_dists); free(frb->fvc_vagdist); free(frb->valvagp);
    free(frb);
}

/* reference src/five/five_add_rule.c:47-95 -- with the capacity check the reference lacks */
int FIVE_add_rule(struct FIVERB *frb, fri_float *rant, fri_float rconc)
{
    int rc;
    const int r = frb->numofrules;
    if (r >= frb->maxnumofrules) { fprintf(stderr, "FIVE_add_rule: rule base full (%d rules)\n", r); return -1; }
    frb->rconc[r] = rconc;
    snap_rule_into(frb, r, rant);
    rc = five_hip_mirror_add_rule(five_dropin_mirror(frb), rant, rconc);
    if (rc) five_dropin_fatal("FIVE_add_rule", rc);
    frb->numofrules = r + 1;
    frb->newrconc = frb->rconc + frb->numofrules;
    frb->newrant = frb->rant + (size_t)frb->numofrules * frb->numofantecedents;
    frb->avx2_rbsize = (unsigned int)((frb->numofrules + 3) / 4);
    return 0;
}

int FIVEAddRule(struct FIVERB *frb, double *newrule) { return FIVE_add_rule(frb, newrule, newrule[frb->numofantecedents]); }
int five_add_rule(struct FIVERB *frb, fri_float *ruletoadd) { return FIVE_add_rule(frb, ruletoadd, ruletoadd[frb->numofantecedents]); }

/* host-side record of a rule the DEVICE appended (frirl_update_sarsa): same bookkeeping, no device call */
void five_dropin_note_appended(struct FIVERB *frb, const double *rant, double rconc)
{
    const int r = frb->numofrules;
    frb->rconc[r] = rconc;
    snap_rule_into(frb, r, rant);
    frb->numofrules = r + 1;
    frb->newrconc = frb->rconc + frb->numofrules;
    frb->newrant = frb->rant + (size_t)frb->numofrules * frb->numofantecedents;
    frb->avx2_rbsize = (unsigned int)((frb->numofrules + 3) / 4);
}

/* reference src/five/five_remove_rule.c:29-85: compact every per-rule array */
int five_remove_rule(struct FIVERB *frb, unsigned int rr)
{
    const int n = frb->numofantecedents, R = frb->numofrules;
    const size_t tail = (size_t)(R - 1 - (int)rr);
    int k, rc;
    if ((int)rr > R - 1) { printf("FIVERemoveRule - FATAL: Invalid rule: %u max: %d !\n", rr, R - 1); exit(6); }
    memmove(frb->rconc + rr, frb->rconc + rr + 1, tail * sizeof(double));
    memmove(frb->rant + (size_t)rr * n, frb->rant + (size_t)(rr + 1) * n, tail * n * sizeof(double));
    memmove(frb->rant_uindex + (size_t)rr * n, frb->rant_uindex + (size_t)(rr + 1) * n, tail * n * sizeof(unsigned int));
    memmove((double *)frb->rant_veval + (size_t)rr * n, (double *)frb->rant_veval + (size_t)(rr + 1) * n, tail * n * sizeof(double));
    for (k = 0; k < n; k++) {
        memmove(frb->rseqant[k] + rr, frb->rseqant[k] + rr + 1, tail * sizeof(double));
        memmove(frb->rseqant_uindex[k] + rr, frb->rseqant_uindex[k] + rr + 1, tail * sizeof(unsigned int));
        memmove(frb->rseqant_veval[k] + rr, frb->rseqant_veval[k] + rr + 1, tail * sizeof(double));
    }
    rc = five_hip_mirror_remove_rule(five_dropin_mirror(frb), rr);
    if (rc) five_dropin_fatal("five_remove_rule", rc);
    frb->numofrules = R - 1;
    frb->newrconc = frb->rconc + frb->numofrules;
    frb->newrant = frb->rant + (size_t)frb->numofrules * n;
    frb->avx2_rbsize = (unsigned int)((frb->numofrules + 3) / 4);
    return 0;
}

/* ---- per-observation calls: one trip to the GPU each ------------------------------------------ */

/* reference src/five/five_rule_distance.c:63-295: fills frb->ruledists, returns the first exact hit or -1 */
int five_rule_distance(struct FIVERB *frb, fri_float *x)
{
    uint32_t hit;
    int rc = five_hip_mirror_rule_distance(five_dropin_mirror(frb), x, frb->ruledists, &hit);
    if (rc) five_dropin_fatal("five_rule_distance", rc);
    return (hit == FRIRL_HIP_NO_HIT) ? -1 : (int)hit;
}

/* reference src/five/FIVEVagConcl.c:64-351 */
unsigned int FIVE_vag_concl(struct FIVERB *frb, double *ant, double *conc)
{
    uint32_t hit;
    int rc = five_hip_mirror_vag_concl(five_dropin_mirror(frb), ant, conc, &hit);
    if (rc) five_dropin_fatal("FIVE_vag_concl", rc);
    return (unsigned int)hit;
}

double FIVEVagConcl(struct FIVERB *frb, double *x)
{
    double conc;
    FIVE_vag_concl(frb, x, &conc);
    return conc;
}

/* reference src/five/FIVEVagConclWeight.c:52-188 */
unsigned int FIVE_vag_concl_weight(struct FIVERB *frb, double *ant, double *weights)
{
    uint32_t hit;
    int rc = five_hip_mirror_vag_concl_weight(five_dropin_mirror(frb), ant, weights, &hit);
    if (rc) five_dropin_fatal("FIVE_vag_concl_weight", rc);
    return (unsigned int)hit;
}

unsigned int FIVEVagConclWeight(struct FIVERB *frb, double *x) { return FIVE_vag_concl_weight(frb, x, frb->weights); }

/* reference src/five/FIVEVagConcl_FRIRL_BestAct.c:56-299 */
double FIVEVagConcl_FRIRL_BestAct(struct FIVERB *frb, double *ruledists)
{
    double conc;
    int rc = five_hip_mirror_bestact(five_dropin_mirror(frb), ruledists, &conc);
    if (rc) five_dropin_fatal("FIVEVagConcl_FRIRL_BestAct", rc);
    return conc;
}

/* ---- off-path API kept for link compatibility (not reached by FRIRL's default build) --- */

/* reference src/five/five_vague_distance.c:51-113 (fixed resolution, primitive-integral VE rows) */
int five_vague_distance(struct FIVERB *frb, fri_float *p1, fri_float *p2, fri_float *d)
{
    int k;
    for (k = 0; k < frb->numofunivs; k++) {
        unsigned int i = five_dropin_snap(frb->uk[k], frb->univlength, p1[k], frb->udivs[k]);
        unsigned int j = five_dropin_snap(frb->uk[k], frb->univlength, p2[k], frb->udivs[k]);
        if (i > j) { unsigned int t = i; i = j; j = t; }
        if (frb->vek[k][0] >= 0) d[k] = frb->vek[k][j] - frb->vek[k][i];
        else { unsigned int q; d[k] = 0; for (q = i + 1; q < j; q++) d[k] += frb->vek[k][q]; }
    }
    return 0;
}

/* reference src/five/five_vague_distance_parallel.c:43-71: never called in the reference tree */
int five_vague_distance_parallel(struct FIVERB *frb, fri_float *p1, int p1_offset, fri_float *p2, fri_float *d)
{
    return five_vague_distance(frb, p1 + p1_offset, p2, d);
}

/* reference src/five/FIVEValVag.c:45-138: only meaningful when the consequent has a vague environment,
 * which FRIRL rule bases never have (rulelength != univlength); not provided by this library. */
int FIVEValVag(struct FIVERB *frb, double *vp)
{
    (void)frb; (void)vp;
    fprintf(stderr, "FIVEValVag: consequent vague environments are outside the FRIRL hot path and not supported\n");
    return -1;
}
