/*
 * frirl_oracle.c -- CPU restatement of the FRIRL / FIVE hot path (see frirl_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Never linked into, imported by or called from the product.
 *
 * Numerics follow the reference's default x86-64 build bit for bit: IEEE double, separate
 * multiply/add (build with -ffp-contract=off, no -mfma), dimension-ordered sums, IEEE sqrt
 * and divide, the x87 `long double` running product of fast_pow, sequential Shepard sums.
 * All file:line citations are relative to the reference root.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* sincos() */
#endif
#include "frirl_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* hashing + generators                                                        */
/* ------------------------------------------------------------------------- */
uint64_t orc_hash_bytes(uint64_t h, const void *p, uint64_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    if (h == 0) h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

uint64_t orc_hash_doubles(uint64_t h, const double *p, uint64_t n)
{
    return orc_hash_bytes(h, p, n * sizeof(double));
}

uint64_t orc_splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double orc_rand_unit(uint64_t *s)
{
    return (double)(orc_splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------------- */
/* init-time tables                                                            */
/* ------------------------------------------------------------------------- */

/* src/frirl/frirl_app_helpers.c:32-44 -- symmetric fixed-step grid; the upper half is the
 * mirrored lower half, so the grid is exactly symmetric about 0. */
void orc_gen_fixres_arr(double *arr, int len, double div)
{
    double from = -((len - 1) * div) / 2;
    int i;
    for (i = 0; i < len / 2 + 1; i++) arr[i] = from + div * i;
    for (; i < len; i++) arr[i] = arr[len - 1 - i] * -1;
}

/* src/five/FIVEGScFunc.c:76-255 with nls == NAN: `c` ends up NAN (:84-92), so every
 * non-constant segment takes the linear branch (:141-151).  psc holds mp rows of np values
 * (point, S) or (point, Sleft, Sright). */
int orc_gsc_func(const double *u, int numofunivs, int U, const double *psc, int mp, int np, double *scf)
{
    if (mp > 1) {
        for (int i = 0; i < numofunivs; i++) {
            for (int j = 0; j < U; j++) {
                const double x = u[i * U + j];
                double *out = &scf[i * U + j];
                if (x < psc[0]) *out = psc[1];                                   /* :106-108 */
                for (int p = 0; p < mp - 1; p++) {                               /* :113-203 */
                    const double p1 = psc[p * np], p2 = psc[(p + 1) * np];
                    if (x >= p1 && x < p2) {
                        const double s1 = (np == 2) ? psc[p * np + 1] : psc[p * np + 2];
                        const double s2 = psc[(p + 1) * np + 1];
                        if (s1 == s2) { *out = s1; continue; }                   /* :135-138 */
                        *out = ((s2 - s1) / (p2 - p1)) * (x - p1) + s1;          /* :146 */
                    }
                }
                if (x >= psc[(mp - 1) * np]) {                                   /* :209-228 */
                    if (np == 2) { *out = psc[(mp - 1) * np + 1]; continue; }
                    if (j == U - 1 && x == psc[(mp - 1) * np]) *out = psc[(mp - 1) * np + 1];
                    else *out = psc[(mp - 1) * np + 2];
                }
            }
        }
        return 0;
    }
    if (np == 1) {                                                               /* :236-242 */
        for (int i = 0; i < numofunivs * U; i++) scf[i] = psc[0];
        return 0;
    }
    for (int i = 0; i < numofunivs; i++)                                         /* :244-262 */
        for (int j = 0; j < U; j++) {
            if (u[i * U + j] < psc[0]) { scf[i * U + j] = psc[1]; continue; }
            scf[i * U + j] = (np == 2) ? psc[i * np + 1] : psc[i * np + 2];
        }
    return 0;
}

/* src/five/FIVEGVagEnv.c:40-102 -- primitive integral (trapezoid rule) of the scaling function;
 * rows containing an infinite scaling factor store per-interval areas behind a -1 marker. */
void orc_gvagenv(const double *u, int numofunivs, int U, const double *scf, double *ve)
{
    for (int k = 0; k < numofunivs; k++) {
        const double *uk = u + k * U, *sk = scf + k * U;
        double *vk = ve + k * U;
        int has_inf = 0;
        for (int i = 0; i < U; i++) if (sk[i] == INFINITY) { has_inf = 1; break; }
        if (has_inf) {
            vk[0] = -1;                                                          /* :61-70 */
            for (int j = 0; j < U - 1; j++) vk[j + 1] = (uk[j + 1] - uk[j]) * (sk[j] + sk[j + 1]) * 0.5;
        } else {
            vk[0] = 0;                                                           /* :81-94 */
            for (int j = 0; j < U - 1; j++) vk[j + 1] = vk[j] + (uk[j + 1] - uk[j]) * (sk[j] + sk[j + 1]) * 0.5;
        }
    }
}

/* src/inl/fast_abs.inl:19-46 */
static inline double orc_fast_abs(double a)
{
    union { double d; uint64_t i; } v = { a };
    v.i &= 0x7fffffffffffffffULL;
    return v.d;
}

/* src/inl/min.inl:71-92 -- nearest grid index on a fixed-step universe; C truncation of the
 * quotient, ties go to the lower index.  The reference reads universe[low+1] even when
 * low == len-1 (one past the row); here that read is guarded and `low` is returned, which is
 * what the reference yields for every sane universe layout. */
unsigned orc_snap(const double *universe, int len, double point, double div)
{
    int low = (int)((point - universe[0]) / div);
    if (low < 0) return 0;
    if (low >= len) return (unsigned)(len - 1);
    if (low + 1 >= len) return (unsigned)low;
    double d1 = universe[low] - point, d2 = universe[low + 1] - point;
    return (orc_fast_abs(d1) <= orc_fast_abs(d2)) ? (unsigned)low : (unsigned)(low + 1);
}

/* src/inl/fast_pow.inl:17-31 -- running product kept in x87 extended precision, rounded to
 * double once on return.  p <= 1 returns b. */
double orc_fast_pow(double b, int p)
{
    long double ret = b;
    for (int i = 0; i < p - 1; i++) ret *= b;
    return (double)ret;
}

/* ------------------------------------------------------------------------- */
/* FIVE engine                                                                 */
/* ------------------------------------------------------------------------- */

/* src/five/FIVEInit.c:55-347 -- owns copies of every table (the reference borrows u/ve/rant/rconc). */
orc_five *orc_five_create(const double *u, const double *ve, int p, int nant, int U,
                          int R, int maxR, const double *rant, const double *rconc)
{
    if (nant > ORC_MAX_NANT || R > maxR) return NULL;
    orc_five *f = (orc_five *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->nant = nant; f->U = U; f->R = 0; f->maxR = maxR;
    f->p = (p == 0) ? nant : p;                                                   /* :89-93 */
    f->u = (double *)malloc(sizeof(double) * nant * U);
    f->ve = (double *)malloc(sizeof(double) * nant * U);
    memcpy(f->u, u, sizeof(double) * nant * U);
    memcpy(f->ve, ve, sizeof(double) * nant * U);
    for (int k = 0; k < nant; k++)                                                /* :244-248 */
        f->udivs[k] = (f->u[k * U + (U - 1)] - f->u[k * U]) / (U - 1);
    f->rant = (double *)calloc((size_t)maxR * nant, sizeof(double));
    f->veval = (double *)calloc((size_t)maxR * nant, sizeof(double));
    f->uidx = (uint32_t *)calloc((size_t)maxR * nant, sizeof(uint32_t));
    f->rconc = (double *)calloc(maxR, sizeof(double));
    f->ruledists = (double *)calloc(maxR, sizeof(double));
    f->weights = (double *)calloc(maxR, sizeof(double));
    f->wi = (double *)calloc(maxR, sizeof(double));
    for (int r = 0; r < R; r++) orc_add_rule(f, rant + (size_t)r * nant, rconc[r]);  /* :258-266 */
    return f;
}

void orc_five_destroy(orc_five *f)
{
    if (!f) return;
    free(f->u); free(f->ve); free(f->rant); free(f->veval); free(f->uidx);
    free(f->rconc); free(f->ruledists); free(f->weights); free(f->wi); free(f);
}

/* src/five/five_add_rule.c:47-95 -- append; every stored antecedent is snapped to its universe
 * (:71-84), so veval[k][r] == ve[k][uidx[k][r]] exactly.  The reference has no capacity check. */
int orc_add_rule(orc_five *f, const double *rant, double rconc)
{
    if (f->R >= f->maxR) return -1;
    const int r = f->R, n = f->nant, U = f->U;
    f->rconc[r] = rconc;
    for (int k = 0; k < n; k++) {
        f->rant[(size_t)r * n + k] = rant[k];
        unsigned j = orc_snap(f->u + k * U, U, rant[k], f->udivs[k]);
        f->uidx[(size_t)k * f->maxR + r] = j;
        f->veval[(size_t)k * f->maxR + r] = f->ve[k * U + j];
    }
    f->R++;
    return 0;
}

/* src/five/five_remove_rule.c:29-85 -- compaction of every per-rule array. */
int orc_remove_rule(orc_five *f, unsigned rr)
{
    if ((int)rr >= f->R) return -1;
    const int n = f->nant;
    size_t tail = (size_t)(f->R - 1 - rr);
    memmove(f->rant + (size_t)rr * n, f->rant + (size_t)(rr + 1) * n, tail * n * sizeof(double));
    memmove(f->rconc + rr, f->rconc + rr + 1, tail * sizeof(double));
    for (int k = 0; k < n; k++) {
        double *v = f->veval + (size_t)k * f->maxR; uint32_t *x = f->uidx + (size_t)k * f->maxR;
        memmove(v + rr, v + rr + 1, tail * sizeof(double));
        memmove(x + rr, x + rr + 1, tail * sizeof(uint32_t));
    }
    f->R--;
    return 0;
}

/* src/five/five_rule_distance.c:63-295.  Pass 1 (K1, :70-102): per dim (q_k - veval[k][r])^2
 * with q_k = ve[k][snap(x_k)].  Pass 2 (K2, :160-236): slots summed in dimension order
 * (unused slots add +0.0), IEEE sqrt, first exact hit (:241-262).  Returns the lowest rule
 * index < R with distance exactly 0.0, else -1.  (The reference stops at the hit group and
 * leaves later ruledists stale, :265-266; callers then use only the index, so all distances
 * are computed here.) */
int orc_rule_distance(orc_five *f, const double *x)
{
    const int n = f->nant, U = f->U, R = f->R;
    double q[ORC_MAX_NANT];
    for (int k = 0; k < n; k++) q[k] = f->ve[k * U + orc_snap(f->u + k * U, U, x[k], f->udivs[k])];
    int hit = -1;
    for (int r = 0; r < R; r++) {
        double d0 = q[0] - f->veval[r];
        double acc = d0 * d0;
        for (int k = 1; k < n; k++) {
            double d = q[k] - f->veval[(size_t)k * f->maxR + r];
            double sq = d * d;
            acc = acc + sq;
        }
        double dist = sqrt(acc);
        f->ruledists[r] = dist;
        if (hit < 0 && dist == 0.0) hit = r;
    }
    return hit;
}

/* Shepard loop shared by FIVEVagConcl.c:224-235 and FIVEVagConcl_FRIRL_BestAct.c:212-217:
 * sequential over rules, wi = 1/fast_pow(|d|,p), vagc += wi*Q, ws += wi; result vagc/ws. */
static double orc_shepard(const orc_five *f, const double *d)
{
    double vagc = 0, ws = 0;
    for (int i = 0; i < f->R; i++) {
        double wi = 1.0 / orc_fast_pow(orc_fast_abs(d[i]), f->p);
        double t = wi * f->rconc[i];
        vagc = vagc + t;
        ws = ws + wi;
    }
    return vagc / ws;
}

/* src/five/FIVEVagConcl.c:64-351, live path (FRIRL_FAST && FIVE_NONAN && FIVE_NOINF):
 * exact hit -> its consequent (:94-99), else Shepard interpolation (:224-244,302,347). */
unsigned orc_vag_concl(orc_five *f, const double *ant, double *conc)
{
    int hit = orc_rule_distance(f, ant);
    if (hit != -1) { *conc = f->rconc[hit]; return (unsigned)hit; }
    *conc = orc_shepard(f, f->ruledists);
    return ~0u;
}

/* src/five/FIVEVagConcl_FRIRL_BestAct.c:56-299: first exact hit (:89-93) or Shepard (:212-217,265). */
double orc_bestact(orc_five *f, const double *d)
{
    for (int i = 0; i < f->R; i++) if (d[i] == 0.0) return f->rconc[i];
    return orc_shepard(f, d);
}

/* src/five/FIVEVagConclWeight.c:52-188: exact hit -> index, weights untouched (:67-69);
 * else wi[r] = 1/fast_pow(|d_r|,p), ws sequential (:125-132), weights[r] = wi[r]/ws (K6, :153-166). */
unsigned orc_vag_concl_weight(orc_five *f, const double *ant, double *weights)
{
    int hit = orc_rule_distance(f, ant);
    if (hit != -1) return (unsigned)hit;
    double ws = 0.0;
    for (int i = 0; i < f->R; i++) {
        f->wi[i] = 1.0 / orc_fast_pow(orc_fast_abs(f->ruledists[i]), f->p);
        ws = ws + f->wi[i];
    }
    for (int i = 0; i < f->R; i++) weights[i] = f->wi[i] / ws;
    return ~0u;
}

/* ------------------------------------------------------------------------- */
/* portable trig: FMA-free, same source text as the HIP env kernels            */
/* ------------------------------------------------------------------------- */
/* Not from the reference (which calls glibc).  Cody-Waite reduction by pi/2 in three parts
 * and minimax-style Taylor kernels; every operation is a plain IEEE mul/add so that host and
 * device produce identical bits.  Accuracy: < 1 ulp-ish on |x| <= 1e4 (checked in tests). */
static inline double orc_k_sin(double x)   /* |x| <= pi/4 */
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double r = S6;
    r = r * z; r = r + S5;
    r = r * z; r = r + S4;
    r = r * z; r = r + S3;
    r = r * z; r = r + S2;
    r = r * z; r = r + S1;
    double t = z * x;
    t = t * r;
    return x + t;
}

static inline double orc_k_cos(double x)   /* |x| <= pi/4 */
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = C6;
    r = r * z; r = r + C5;
    r = r * z; r = r + C4;
    r = r * z; r = r + C3;
    r = r * z; r = r + C2;
    r = r * z; r = r + C1;
    r = r * z;
    r = r * z;          /* z^2 * poly */
    double h = 0.5 * z;
    double w = 1.0 - h;
    double e = (1.0 - w) - h;   /* rounding error of 1 - h */
    e = e + r;
    return w + e;
}

static inline double orc_reduce(double x, int *quad)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double P2 = 6.07710050630396597660e-11;   /* next 33 bits */
    const double P3 = 2.02226624871116645580e-21;   /* tail */
    double fn = x * INV_PIO2;
    fn = (fn >= 0.0) ? floor(fn + 0.5) : -floor(0.5 - fn);
    double a = fn * P1, b = fn * P2, c = fn * P3;
    double r = x - a;
    r = r - b;
    r = r - c;
    long long n = (long long)fn;
    *quad = (int)(n & 3);
    return r;
}

double orc_sin(double x)
{
    int q; double r = orc_reduce(x, &q);
    switch (q) {
        case 0: return orc_k_sin(r);
        case 1: return orc_k_cos(r);
        case 2: return -orc_k_sin(r);
        default: return -orc_k_cos(r);
    }
}

double orc_cos(double x)
{
    int q; double r = orc_reduce(x, &q);
    switch (q) {
        case 0: return orc_k_cos(r);
        case 1: return -orc_k_sin(r);
        case 2: return -orc_k_cos(r);
        default: return orc_k_sin(r);
    }
}

static inline double t_sin(const orc_frirl *fr, double x) { return fr->trig_mode ? orc_sin(x) : sin(x); }
static inline double t_cos(const orc_frirl *fr, double x) { return fr->trig_mode ? orc_cos(x) : cos(x); }
/* sin and cos of the SAME argument: gcc -O2 fuses the reference's sin(x)/cos(x) pairs into one glibc
 * sincos() call (cartpole.c:60-62, acrobot.c:58-66), whose sine differs from sin() by 1 ulp on rare
 * inputs (observed: acrobot step 16475 of the demo).  The oracle mirrors the reference as compiled. */
static inline void t_sincos(const orc_frirl *fr, double x, double *s, double *c)
{
    if (fr->trig_mode) { *s = orc_sin(x); *c = orc_cos(x); } else sincos(x, s, c);
}

/* ------------------------------------------------------------------------- */
/* environments                                                                */
/* ------------------------------------------------------------------------- */
#define ORC_PI 3.14159265358979323846264338327   /* the literal every example defines */

/* examples/mountaincar/mountaincar.c:37-74 */
static void mc_do_action(const orc_frirl *fr, double a, const double *s, double *ns)
{
    double pos = s[0], vel = s[1];
    double v1 = (vel + (0.001 * a) + (-0.0025 * t_cos(fr, 3.0 * pos))) * 0.999;
    if (v1 < -0.07) v1 = -0.07;
    if (v1 > +0.07) v1 = +0.07;
    double p1 = pos + v1;
    if (p1 <= -1.5) { p1 = -1.5; v1 = 0.0; }
    ns[0] = p1; ns[1] = v1;
}

/* examples/mountaincar/mountaincar.c:77-95 */
static void mc_get_reward(const double *s, double *r, int *f)
{
    *r = -10; *f = 0;
    if (s[0] >= 0.45) { *r = 1000; *f = 1; }
}

/* examples/mountaincar/mountaincar.c:98-125 and examples/acrobot/acrobot.c:165-192 */
static void generic_quantize(const orc_frirl *fr, const double *s, double *ns)
{
    for (int i = 0; i < fr->nstates; i++) {
        const orc_dim *d = &fr->statedims[i];
        int where = (int)round((s[i] + fabs(d->values[0])) / d->values_div);
        if (where < 0) where = 0;
        else if (where > d->values_len - 1) where = d->values_len - 1;
        ns[i] = d->values[where];
    }
}

/* examples/cartpole/cartpole.c:35-77 -- Euler step, tau = 0.02 */
static void cp_do_action(const orc_frirl *fr, double a, const double *s, double *ns)
{
    double x = s[0], xd = s[1], th = s[2], thd = s[3];
    const double g = 9.8, mc = 1.0, mp = 0.1, mt = mc + mp, len = 0.5, pml = mp * len;
    const double fmag = 10.0, tau = 0.02, fourthirds = 4.0 / 3.0;
    double force = a * fmag;
    double sn, cs; t_sincos(fr, th, &sn, &cs);
    double temp = (force + pml * thd * thd * sn) / mt;
    double thacc = (g * sn - cs * temp) / (len * (fourthirds - mp * cs * cs / mt));
    double xacc = temp - pml * thacc * cs / mt;
    ns[0] = x + tau * xd;
    ns[1] = xd + tau * xacc;
    ns[2] = th + tau * thd;
    ns[3] = thd + tau * thacc;
}

/* examples/cartpole/cartpole.c:79-112 */
static void cp_get_reward(const double *s, double *r, int *f)
{
    double x = s[0], th = s[2], thd = s[3];
    const double deg45 = ORC_PI / 4;
    if ((x < -4.0) || (x > 4.0) || (th < (-1 * deg45)) || (th > deg45)) {
        *r = -10000 - 50 * fabs(x) - 100 * fabs(th); *f = 1;
    } else {
        *r = 10 - 1000 * th * th - 5 * fabs(x) - 10 * thd; *f = 0;
    }
}

/* examples/cartpole/cartpole.c:114-168 -- bespoke quantiser */
static void cp_quantize(const double *s, double *ns)
{
    const double deg12 = ORC_PI / 15, deg3 = ORC_PI / 60;
    double q0 = s[0], q1 = round(s[1]), q2 = floor(s[2] / deg3) * deg3, q3 = s[3];
    if (q0 < 0) q0 = -1;
    if (q0 > 0) q0 = 1;
    if (q1 < -1) q1 = -1;
    if (q1 > 1) q1 = 1;
    if (q2 > deg12) q2 = deg12;
    if (q2 < (-1 * deg12)) q2 = -1 * deg12;
    if (q3 < 0) q3 = -1;
    if (q3 > 0) q3 = 1;
    ns[0] = q0; ns[1] = q1; ns[2] = q2; ns[3] = q3;
}

/* examples/acrobot/acrobot.c:31-130 -- accelerations once, then 4 Euler sub-steps of 0.05 s */
static void ab_do_action(const orc_frirl *fr, double torque, const double *s, double *ns)
{
    const double vmax1 = 4 * ORC_PI, vmax2 = 9 * ORC_PI;
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8, dt = 0.05;
    const double l1sq = l1 * l1, lc1sq = lc1 * lc1, lc2sq = lc2 * lc2;
    double t1 = s[0], t2 = s[1], t1d = s[2], t2d = s[3];
    double c2, s2; t_sincos(fr, t2, &s2, &c2);
    double d1 = m1 * lc1sq + m2 * (l1sq + lc2sq + 2 * l1 * lc2 * c2) + I1 + I2;
    double d2 = m2 * (lc2sq + l1 * lc2 * c2) + I2;
    double phi2 = m2 * lc2 * g * t_cos(fr, t1 + t2 - ORC_PI / 2);
    double phi1 = -m2 * l1 * lc2 * t2d * s2 * (t2d - 2 * t1d) + (m1 * lc1 + m2 * l1) * g * t_cos(fr, t1 - (ORC_PI / 2)) + phi2;
    double acc2 = (torque + phi1 * (d2 / d1) - m2 * l1 * lc2 * t1d * t1d * s2 - phi2);
    acc2 = acc2 / (m2 * lc2sq + I2 - (d2 * d2 / d1));
    double acc1 = -(d2 * acc2 + phi1) / d1;
    for (int i = 0; i < 4; i++) {
        t1d = t1d + acc1 * dt;
        if (t1d < -vmax1) t1d = -vmax1;
        if (t1d > vmax1) t1d = vmax1;
        t1 = t1 + t1d * dt;
        t2d = t2d + acc2 * dt;
        if (t2d < -vmax2) t2d = -vmax2;
        if (t2d > vmax2) t2d = vmax2;
        t2 = t2 + t2d * dt;
    }
    if (t1 < -ORC_PI) t1 = -ORC_PI;
    if (t1 > ORC_PI) t1 = ORC_PI;
    if (t2 < -ORC_PI) t2 = -ORC_PI;
    if (t2 > ORC_PI) t2 = ORC_PI;
    ns[0] = t1; ns[1] = t2; ns[2] = t1d; ns[3] = t2d;
}

/* examples/acrobot/acrobot.c:133-162 */
static void ab_get_reward(const orc_frirl *fr, const double *s, double *r, int *f)
{
    double y1 = 0.0 - t_cos(fr, s[0]);
    double y2 = y1 - t_cos(fr, s[1]);
    double goal = 0.0 + 1.0;
    *r = -10; *f = 0;
    if (y2 >= goal) { *r = 1000; *f = 1; }
}

void orc_env_do_action(const orc_frirl *fr, double a, const double *s, double *ns)
{
    switch (fr->env) {
        case ORC_ENV_MOUNTAINCAR: mc_do_action(fr, a, s, ns); break;
        case ORC_ENV_CARTPOLE:    cp_do_action(fr, a, s, ns); break;
        default:                  ab_do_action(fr, a, s, ns); break;
    }
}

void orc_env_get_reward(const orc_frirl *fr, const double *s, double *r, int *f)
{
    switch (fr->env) {
        case ORC_ENV_MOUNTAINCAR: mc_get_reward(s, r, f); break;
        case ORC_ENV_CARTPOLE:    cp_get_reward(s, r, f); break;
        default:                  ab_get_reward(fr, s, r, f); break;
    }
}

void orc_env_quantize(const orc_frirl *fr, const double *s, double *ns)
{
    if (fr->env == ORC_ENV_CARTPOLE) cp_quantize(s, ns); else generic_quantize(fr, s, ns);
}

/* ------------------------------------------------------------------------- */
/* demo configurations (hyper-parameters are data)                             */
/* ------------------------------------------------------------------------- */
static void dim_set(orc_dim *d, int n, const double *vals, double vdiv, double steep, double def, double udiv)
{
    memset(d, 0, sizeof(*d));
    d->values_len = n; d->values_div = vdiv; d->values_steep = steep; d->values_def = def; d->universe_div = udiv;
    if (vals) memcpy(d->values, vals, sizeof(double) * n);
    else orc_gen_fixres_arr(d->values, n, vdiv);      /* FRIRL_GEN_FIXRES_VALUES, frirl_app_helpers.h:35 */
}

/* src/frirl/frirl_types_def.h:22-77 (defaults) overridden as each example's main() does:
 * examples/mountaincar/mountaincar.c:222-279 (run here in construct mode),
 * examples/cartpole/cartpole.c:254-402, examples/acrobot/acrobot.c:244-324. */
void orc_frirl_config(orc_frirl *fr, int env)
{
    memset(fr, 0, sizeof(*fr));
    fr->env = env;
    fr->alpha = 0.5; fr->gamma = 1.0; fr->epsilon = 0.001;
    fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -250.0; fr->qdiff_final_tolerance = 250.0;
    fr->reward_good_above = 0.0; fr->weight_significant = 0.05;
    fr->skip_rules = 0; fr->no_random = 1; fr->max_episodes = 1000; fr->max_steps = 1000; fr->maxR = 16384;
    if (env == ORC_ENV_MOUNTAINCAR) {
        static const double v0[] = { -1.5, -1.295, -1.09, -0.885, -0.68, -0.475, -0.27, -0.065, 0.14, 0.345 };
        static const double v1[] = { -0.07, -0.042, -0.014, 0.014, 0.042, 0.07 };
        fr->reward_good_above = -5000.0;
        fr->alpha = 0.5; fr->gamma = 1.0; fr->epsilon = 0.01;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -4.0; fr->qdiff_final_tolerance = 500.0;
        fr->skip_rules = 1; fr->U = 41; fr->nstates = 2;
        dim_set(&fr->statedims[0], 10, v0, 0.205, 1.0, -0.5, 0.1);
        dim_set(&fr->statedims[1], 6, v1, 0.028, 1.0, 0.0, 0.005);
        dim_set(&fr->actiondim, 3, NULL, 1.0, 0.0, 0.0, 0.1);
    } else if (env == ORC_ENV_CARTPOLE) {
        static const double v2[] = { -0.2094, -0.1571, -0.1047, -0.0524, 0.0, 0.0524, 0.1047, 0.1571, 0.2094 };
        static const double va[] = { -1.0, -0.9, -0.8, -0.7, -0.6, -0.5, -0.3999999999999999,
            -0.29999999999999992, -0.19999999999999995, -0.09999999999999998, 0.0,
            +0.09999999999999998, +0.19999999999999995, +0.29999999999999992,
            +0.3999999999999999, +0.5, +0.6, +0.7, +0.8, +0.9, +1.0 };
        fr->reward_good_above = 0.0;
        fr->alpha = 0.3; fr->gamma = 1.0; fr->epsilon = 0.001;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -200.0; fr->qdiff_final_tolerance = 250.0;
        fr->skip_rules = 1; fr->U = 1001; fr->nstates = 4;
        dim_set(&fr->statedims[0], 2, NULL, 2.0, 1.0, 1.0, 0.016);
        dim_set(&fr->statedims[1], 3, NULL, 1.0, 1.0, 0.0, 0.032);
        dim_set(&fr->statedims[2], 9, v2, 0.0524, 21.485917317405871, 0.0, 0.0031415926535897933);
        dim_set(&fr->statedims[3], 2, NULL, 2.0, 1.0, 0.0, 0.016);
        dim_set(&fr->actiondim, 21, va, 0.1, 0.0, 0.0, 0.008);
    } else {
        static const double v0[] = { -1.570796326794897, -0.785398163397448, 0, 0.785398163397448, 1.570796326794897 };
        fr->reward_good_above = 0.0;
        fr->alpha = 0.5; fr->gamma = 1.0; fr->epsilon = 0.001;
        fr->qdiff_pos_boundary = 1.0; fr->qdiff_neg_boundary = -200.0; fr->qdiff_final_tolerance = 50.0;
        fr->skip_rules = 1; fr->U = 41; fr->nstates = 4;
        dim_set(&fr->statedims[0], 5, v0, 0.785398163397448, 1.0, 0.0, 0.1);
        dim_set(&fr->statedims[1], 5, v0, 0.785398163397448, 1.0, 0.0, 0.1);
        dim_set(&fr->statedims[2], 3, NULL, 0.785398163397448, 1.0, 0.0, 0.05);
        dim_set(&fr->statedims[3], 3, NULL, 0.785398163397448, 1.0, 0.0, 0.05);
        dim_set(&fr->actiondim, 3, NULL, 1.0, 0.0, 0.0, 0.1);
    }
}

/* src/frirl/frirl_init.c:29-341: universes (:36-48), VE tables (frirl_init_ve.c:25-121), the
 * 2^nant corner rule base (frirl_init_rb.c:86-147), FIVEInit, per-action VE values (:156-158). */
int orc_frirl_init(orc_frirl *fr)
{
    const int ns = fr->nstates, nant = ns + 1, U = fr->U, A = fr->actiondim.values_len;
    double *u = (double *)malloc(sizeof(double) * nant * U);
    double *ve = (double *)malloc(sizeof(double) * nant * U);
    double *scf = (double *)malloc(sizeof(double) * U);
    for (int k = 0; k < ns; k++) orc_gen_fixres_arr(u + k * U, U, fr->statedims[k].universe_div);
    orc_gen_fixres_arr(u + ns * U, U, fr->actiondim.universe_div);

    for (int k = 0; k < ns; k++) {                                   /* frirl_init_ve.c:37-80 */
        const orc_dim *d = &fr->statedims[k];
        double *sp = (double *)malloc(sizeof(double) * 3 * d->values_len);
        for (int c = 0; c < d->values_len; c++) { sp[3 * c] = d->values[c]; sp[3 * c + 1] = sp[3 * c + 2] = d->values_steep; }
        orc_gsc_func(u + k * U, 1, U, sp, d->values_len, 3, scf);
        orc_gvagenv(u + k * U, 1, U, scf, ve + k * U);
        free(sp);
    }
    {                                                                /* frirl_init_ve.c:82-107 */
        double *sp = (double *)malloc(sizeof(double) * 3 * A);
        double divratio = 1.0 / (A - 1) * 2.0;
        for (int i = 0; i < A; i++) {
            sp[3 * i] = i * divratio - 1.0;
            sp[3 * i + 1] = sp[3 * i + 2] = (A - 1) / 2;             /* integer division, :91-92 */
        }
        orc_gsc_func(u + ns * U, 1, U, sp, A, 3, scf);
        orc_gvagenv(u + ns * U, 1, U, scf, ve + ns * U);
        free(sp);
    }
    free(scf);

    int R0 = 1 << nant;                                              /* frirl_init_rb.c:99 */
    double *rant = (double *)malloc(sizeof(double) * R0 * nant);
    double *rconc = (double *)calloc(R0, sizeof(double));
    for (int i = 0; i < nant; i++) {                                 /* frirl_init_rb.c:111-126 */
        const orc_dim *d = (i < ns) ? &fr->statedims[i] : &fr->actiondim;
        double mn = d->values[0], mx = d->values[0];
        for (int c = 0; c < d->values_len; c++) { if (d->values[c] > mx) mx = d->values[c]; if (d->values[c] < mn) mn = d->values[c]; }
        unsigned divider = (unsigned)R0 >> (i + 1);
        for (int j = 0; j < R0; j++) rant[j * nant + i] = (((j / divider) % 2) == 0) ? mn : mx;
    }
    fr->frb = orc_five_create(u, ve, 0, nant, U, R0, fr->maxR, rant, rconc);
    free(rant); free(rconc);
    if (!fr->frb) { free(u); free(ve); return -1; }
    for (int j = 0; j < A; j++)                                      /* frirl_init.c:156-158: len = uksize = U-1 */
        fr->action_vevalues[j] = ve[ns * U + orc_snap(u + ns * U, U - 1, fr->actiondim.values[j], fr->frb->udivs[nant - 1])];
    free(u); free(ve);
    fr->statedistsum = (double *)calloc(fr->maxR, sizeof(double));
    fr->ruledist = (double *)calloc(fr->maxR, sizeof(double));
    fr->ep_total_value = -1; fr->ep_total_steps = -1;                /* frirl_init.c:149-150 */
    fr->fus_is_rule_inserted = 0; fr->episode_num = 1; fr->epended = 0;
    fr->step_hash = 0; fr->total_steps = 0;
    return 0;
}

void orc_frirl_deinit(orc_frirl *fr)
{
    orc_five_destroy(fr->frb); fr->frb = NULL;
    free(fr->statedistsum); free(fr->ruledist);
    fr->statedistsum = fr->ruledist = NULL;
}

/* ------------------------------------------------------------------------- */
/* agent                                                                       */
/* ------------------------------------------------------------------------- */

/* src/frirl/frirl_get_best_action.c:31-341.  K3/K4 (:58-155): state-only squared distances
 * summed in dimension order; K5 (:252-275) per action: sqrt((vevalues[a]-ract_veval[r])^2 +
 * statedistsum[r]); conclusion per action by FIVEVagConcl_FRIRL_BestAct (:325); first maximum
 * wins (src/inl/max.inl:16-28).  Bare form: any rule base + the per-action VE values. */
unsigned orc_five_best_action(orc_five *f, const double *states, const double *action_ve, int A, double *actconc)
{
    const int n = f->nant, ns = n - 1, U = f->U, R = f->R;
    double q[ORC_MAX_NANT];
    double *statedistsum = f->wi;          /* scratch: wi[] is only live inside orc_vag_concl_weight */
    double *ruledist = f->ruledists;
    for (int k = 0; k < ns; k++) q[k] = f->ve[k * U + orc_snap(f->u + k * U, U, states[k], f->udivs[k])];
    for (int r = 0; r < R; r++) {
        double d0 = q[0] - f->veval[r];
        double acc = d0 * d0;
        for (int k = 1; k < ns; k++) {
            double d = q[k] - f->veval[(size_t)k * f->maxR + r];
            double sq = d * d;
            acc = acc + sq;
        }
        statedistsum[r] = acc;
    }
    const double *av = f->veval + (size_t)ns * f->maxR;
    for (int a = 0; a < A; a++) {
        for (int r = 0; r < R; r++) {
            double da = action_ve[a] - av[r];
            double sq = da * da;
            ruledist[r] = sqrt(sq + statedistsum[r]);
        }
        actconc[a] = orc_bestact(f, ruledist);
    }
    int best = 0;
    for (int a = 1; a < A; a++) if (actconc[best] < actconc[a]) best = a;
    return (unsigned)best;
}

unsigned orc_get_best_action(orc_frirl *fr, const double *states)
{
    return orc_five_best_action(fr->frb, states, fr->action_vevalues, fr->actiondim.values_len, fr->actconc);
}

/* src/frirl/frirl_e_greedy_selection.c:21-37.  Every shipped demo keeps no_random = 1, so the
 * libc rand() branch (:28-33) is never taken; the oracle restates the greedy branch only. */
unsigned orc_e_greedy(orc_frirl *fr, const double *states)
{
    return orc_get_best_action(fr, states);
}

/* src/frirl/frirl_check_possible_states.c:96-122 with hit_between_possible_places (:53-88).
 * The grid-refinement branch (:68-75, increment_place) needs epsilon <= diff/2 with
 * epsilon = values_div (frirl_init.c:74), which never holds on these grids. */
double orc_check_possible_states(double obs, const double *v, int n)
{
    int i, found = 0;
    for (i = 0; i < n; i++) if (obs < v[i]) { found = 1; break; }
    if (!found) return v[n - 1];
    if (obs < v[0]) return v[0];
    i--;
    double rel = obs - v[i], rel_next = v[i + 1] - obs;
    return (rel < rel_next) ? v[i] : v[i + 1];
}

/* src/frirl/frirl_update_sarsa.c:22-143 */
static void orc_update_rules(orc_five *f, const orc_agent *ag, double *fus, const double *values, double qnow, double qdiff)
{
    int rules = f->R;
    if (*fus) rules--;                                                            /* :30-33 */
    unsigned hit = orc_vag_concl_weight(f, values, f->weights);                   /* :40 */
    if (hit != ~0u && (ag->skip_rules == 0 || (ag->skip_rules == 1 && hit < (unsigned)rules))) {
        f->rconc[hit] = qnow + qdiff;                                             /* :55 */
        return;
    } else if (ag->skip_rules == 1 && hit == (unsigned)rules) {
        return;                                                                   /* :61-63 */
    }
    double save = 0;
    if (ag->skip_rules == 0) *fus = 0;                                            /* :70-73 */
    else save = f->rconc[f->R - 1];                                               /* :76 */
    for (int r = 0; r < f->R; r++)                                                /* K7, :89-120 */
        if (f->weights[r] > ag->weight_significant) {
            double t = qdiff * f->weights[r];
            f->rconc[r] = qnow + t;
        }
    if (*fus) f->rconc[f->R - 1] = save;                                          /* :124-126 */
}

/* src/frirl/frirl_update_sarsa.c:348-385 (check_possible_states :146-170, CHECK_STATES = 1).
 * Bare form: rule base + agent parameters + the sticky fus_is_rule_inserted flag. */
void orc_five_update_sarsa(orc_five *f, const orc_agent *ag, double *fus, const double *q_ant, double reward, const double *cur_q_ant)
{
    double qp, qnow;
    const int n = f->nant;
    orc_vag_concl(f, cur_q_ant, &qp);
    unsigned rule_i = orc_vag_concl(f, q_ant, &qnow);
    double qdiff = ag->alpha * (reward + ag->gamma * qp - qnow);
    if (qdiff > ag->qdiff_pos_boundary || qdiff < ag->qdiff_neg_boundary) {
        double rant[ORC_MAX_NANT], rconc = qnow;
        for (int i = 0; i < n; i++) rant[i] = orc_check_possible_states(q_ant[i], ag->grid[i], ag->grid_len[i]);
        rule_i = orc_vag_concl(f, rant, &rconc);
        if (rule_i == ~0u) {
            *fus = 1;
            orc_add_rule(f, rant, rconc + qdiff);
            return;
        }
        *fus = 0;
    }
    orc_update_rules(f, ag, fus, q_ant, qnow, qdiff);
}

/* src/frirl/frirl_agent.c:58-117 (merge_rb, the BUILD_OPENMP / BUILD_MPI body; CHECK_STATES = 1): the receiver takes over the
 * sender's rules one after the other.  Per sender rule: receiver's Shepard weights (left untouched on an exact hit, so a later
 * hit re-uses the weights of the last interpolated rule -- :72 passes the receiver's own weights array) and conclusion at the
 * sender's antecedents; qdiff = sender Q - receiver Q (:82).  Outside the qdiff boundaries the antecedents are snapped to the
 * receiver's rule grid (:96-99): a new place gets a new rule with the mean of the two conclusions (:101), an existing one moves
 * 10 % towards the sender (:105); inside the boundaries every rule with weight > delta is OVERWRITTEN with q * weight,
 * q = 0.9 receiver + 0.1 sender (:45-53 -- the agent file's own update_rules, not frirl_update_sarsa.c's). */
void orc_merge_rb(orc_five *rcvr, const orc_agent *ag, const double *newrant, const double *newrconc, int numofrules)
{
    const int n = rcvr->nant;
    for (int r = 0; r < numofrules; r++) {
        const double *sndr_rant = newrant + (size_t)r * n;
        const double sndr_rconc = newrconc[r];
        unsigned rcvr_rule_i = orc_vag_concl_weight(rcvr, sndr_rant, rcvr->weights);
        double rcvr_rconc;
        orc_vag_concl(rcvr, sndr_rant, &rcvr_rconc);
        const double qdiff = -rcvr_rconc + sndr_rconc;
        if (qdiff > ag->qdiff_pos_boundary || qdiff < ag->qdiff_neg_boundary) {
            double rant[ORC_MAX_NANT], rconc = rcvr_rconc;
            for (int i = 0; i < n; i++) rant[i] = orc_check_possible_states(sndr_rant[i], ag->grid[i], ag->grid_len[i]);
            rcvr_rule_i = orc_vag_concl(rcvr, rant, &rconc);
            if (rcvr_rule_i == ~0u) { orc_add_rule(rcvr, rant, 0.5 * rconc + 0.5 * sndr_rconc); continue; }
            rcvr->rconc[rcvr_rule_i] = 0.9 * rconc + 0.1 * sndr_rconc;
            continue;
        }
        const double q = 0.9 * rcvr_rconc + 0.1 * sndr_rconc;
        for (int w = 0; w < rcvr->R; w++)
            if (rcvr->weights[w] > ag->weight_significant) rcvr->rconc[w] = q * rcvr->weights[w];
    }
}

/* src/frirl/frirl_agent.c:121-139 (gen_def_states): agent `id` of `worldsize` starts its episodes from the state antecedents of
 * the master's rule (id-1) * gap, gap = numofrules / (worldsize - 2) (integer division as written: worldsize 2 divides by zero
 * in the reference; callers here use worldsize >= 3).  Agent 0 keeps its start state (returns 0). */
int orc_gen_def_states(const orc_five *master, int id, int worldsize, int nstates, double *values_def)
{
    if (id == 0 || worldsize < 3) return 0;
    const int gap = master->R / (worldsize - 2);
    size_t rule = (size_t)(id - 1) * gap;
    if (rule >= (size_t)master->R) rule = (size_t)master->R - 1;   /* the reference reads one rule past its list for the last agent (uninitialised): last rule */
    for (int i = 0; i < nstates; i++) values_def[i] = master->rant[rule * master->nant + i];
    return 1;
}

void orc_frirl_agent(const orc_frirl *fr, orc_agent *ag)
{
    memset(ag, 0, sizeof(*ag));
    ag->alpha = fr->alpha; ag->gamma = fr->gamma;
    ag->qdiff_pos_boundary = fr->qdiff_pos_boundary; ag->qdiff_neg_boundary = fr->qdiff_neg_boundary;
    ag->weight_significant = fr->weight_significant; ag->skip_rules = fr->skip_rules;
    for (int i = 0; i < fr->nstates; i++) { ag->grid[i] = fr->statedims[i].values; ag->grid_len[i] = fr->statedims[i].values_len; }
    ag->grid[fr->nstates] = fr->actiondim.values; ag->grid_len[fr->nstates] = fr->actiondim.values_len;
}

void orc_update_sarsa(orc_frirl *fr, const double *q_ant, double reward, const double *cur_q_ant)
{
    orc_agent ag;
    orc_frirl_agent(fr, &ag);
    orc_five_update_sarsa(fr->frb, &ag, &fr->fus_is_rule_inserted, q_ant, reward, cur_q_ant);
}

/* src/frirl/frirl_episode.c:28-194.  The first action is chosen on the un-quantised default
 * state (:46-48,78).  The step hash mixes, in callback order, exactly what a wrapper around the
 * three callbacks can observe (see oracle/ref_harness.c): action + new states, reward + success,
 * quantised states + current rule count. */
void orc_episode(orc_frirl *fr)
{
    const int ns = fr->nstates, n = ns + 1;
    double states[ORC_MAX_NANT], cur_states[ORC_MAX_NANT], q_ant[ORC_MAX_NANT], cur_q_ant[ORC_MAX_NANT];
    for (int i = 0; i < ns; i++) q_ant[i] = states[i] = fr->statedims[i].values_def;
    fr->ep_total_value = 0; fr->ep_total_steps = 0;
    unsigned ai = orc_e_greedy(fr, states);
    q_ant[ns] = fr->actiondim.values[ai];
    for (int step = 1; step <= fr->max_steps; step++) {
        orc_env_do_action(fr, q_ant[ns], states, cur_states);
        fr->step_hash = orc_hash_doubles(fr->step_hash, &q_ant[ns], 1);
        fr->step_hash = orc_hash_doubles(fr->step_hash, cur_states, ns);
        orc_env_get_reward(fr, cur_states, &fr->reward_value, &fr->success);
        fr->ep_total_value += fr->reward_value;
        { double rs[2] = { fr->reward_value, (double)fr->success }; fr->step_hash = orc_hash_doubles(fr->step_hash, rs, 2); }
        orc_env_quantize(fr, cur_states, cur_q_ant);
        { double nr = (double)fr->frb->R; fr->step_hash = orc_hash_doubles(fr->step_hash, cur_q_ant, ns);
          fr->step_hash = orc_hash_doubles(fr->step_hash, &nr, 1); }
        unsigned pa = orc_e_greedy(fr, cur_q_ant);
        cur_q_ant[ns] = fr->actiondim.values[pa];
        if (fr->trace) fr->trace(fr, step, q_ant[ns], cur_states, cur_q_ant, fr->trace_ud);
        orc_update_sarsa(fr, q_ant, fr->reward_value, cur_q_ant);
        for (int i = 0; i < ns; i++) states[i] = cur_states[i];
        for (int i = 0; i < n; i++) q_ant[i] = cur_q_ant[i];
        fr->ep_total_steps++; fr->total_steps++;
        if (fr->success == 1) break;
    }
}

/* src/frirl/frirl_sequential_run.c:24-165 -- construct loop: at most max_episodes-1 episodes
 * (:51,59); converged when #rules, #steps and reward repeat, reward > reward_good_above and no
 * consequent moved by >= qdiff_final_tolerance (:83-148). */
int orc_sequential_run(orc_frirl *fr, int verbose)
{
    orc_five *f = fr->frb;
    double *prev = (double *)malloc(sizeof(double) * fr->maxR);
    int epchunk = 1;
    fr->epended = 0;
    for (;;) {
        if (!(epchunk < fr->max_episodes)) break;
        int prev_R = f->R, prev_steps = fr->ep_total_steps;
        double prev_reward = fr->ep_total_value;
        memcpy(prev, f->rconc, sizeof(double) * fr->maxR);
        orc_episode(fr);
        if (verbose) printf("#0 Episode: %u\tSteps: %d\tReward: %f\tRules: %d\n", fr->episode_num, fr->ep_total_steps, fr->ep_total_value, f->R);
        int epend = 0;
        if (prev_R == f->R && prev_steps == fr->ep_total_steps && fr->ep_total_value > fr->reward_good_above &&
            prev_reward == fr->ep_total_value) {
            epend = 1;
            for (int i = 0; i < f->R; i++) if (fabs(f->rconc[i] - prev[i]) >= fr->qdiff_final_tolerance) { epend = 0; break; }
        }
        if (epend) { fr->epended = 1; break; }
        fr->episode_num++; epchunk++;
    }
    free(prev);
    return fr->epended;
}

/* src/frirl/frirl_utils.c:100-144 -- "%.18f " per antecedent, "%.18f \n" for Q */
int orc_save_rb_text(orc_frirl *fr, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    orc_five *f = fr->frb;
    for (int i = 0; i < f->R; i++) {
        for (int j = 0; j < f->nant; j++) fprintf(fp, "%.18f ", f->rant[(size_t)i * f->nant + j]);
        fprintf(fp, "%.18f \n", f->rconc[i]);
    }
    fclose(fp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* synthetic problems                                              */
/* ------------------------------------------------------------------------- */
/* Universes: symmetric fixed step; dim k spans +-(k+1); scaling function drawn in [0.5,1.5)
 * (smooth enough, strictly positive => strictly increasing VE, no duplicate VE values). */
void orc_synth_tables(int nant, int U, uint64_t seed, double *u, double *ve)
{
    uint64_t s = seed ^ 0x7AB1E5ULL;
    double *scf = (double *)malloc(sizeof(double) * U);
    for (int k = 0; k < nant; k++) {
        double div = 2.0 * (k + 1) / (U - 1);
        orc_gen_fixres_arr(u + k * U, U, div);
        for (int j = 0; j < U; j++) scf[j] = 0.5 + orc_rand_unit(&s);
        orc_gvagenv(u + k * U, 1, U, scf, ve + k * U);
    }
    free(scf);
}

/* Rules on the universe grid, duplicate-free: rule r's dim-0 index is a fixed permutation-like
 * function of r when R <= U^... is not guaranteed, so duplicates are removed by construction:
 * the tuple (idx_0..idx_{n-1}) is derived from a bijective mixed-radix counter scrambled per
 * dimension.  Action dim (last) uses A evenly spaced grid indices when A > 0. */
void orc_synth_rules(int nant, int U, int R, int A, uint64_t seed, uint32_t *uidx, double *rconc)
{
    uint64_t s = seed ^ 0x5EED0000ULL;
    /* per-dimension affine scramble idx -> (a*idx + b) mod U with gcd(a,U)=1 keeps bijectivity */
    uint32_t a[ORC_MAX_NANT], b[ORC_MAX_NANT], radix[ORC_MAX_NANT];
    for (int k = 0; k < nant; k++) {
        radix[k] = (uint32_t)((A > 0 && k == nant - 1) ? A : U);
        uint32_t cand;
        do {
            cand = (uint32_t)(orc_splitmix64(&s) % radix[k]);
            uint32_t x = cand, y = radix[k];
            while (y) { uint32_t t = x % y; x = y; y = t; }
            if (x == 1 || radix[k] == 1) break;
        } while (1);
        a[k] = cand ? cand : 1; b[k] = (uint32_t)(orc_splitmix64(&s) % radix[k]);
    }
    /* stride through the mixed-radix space with a step coprime to its size (when it fits 64 bits) */
    long double space = 1; for (int k = 0; k < nant; k++) space *= radix[k];
    uint64_t total = (space > 1.8e19L) ? 0 : (uint64_t)space;   /* 0 => treat as 2^64 */
    uint64_t step = 0x9E3779B97F4A7C15ULL;
    if (total) { step %= total; if (!step) step = 1;
        for (;;) { uint64_t x = step, y = total; while (y) { uint64_t t = x % y; x = y; y = t; } if (x == 1) break; step++; } }
    uint64_t c = total ? (orc_splitmix64(&s) % total) : orc_splitmix64(&s);
    for (int r = 0; r < R; r++) {
        uint64_t t = c;
        for (int k = 0; k < nant; k++) {
            uint32_t digit = (uint32_t)(t % radix[k]); t /= radix[k];
            uint32_t idx = (uint32_t)(((uint64_t)a[k] * digit + b[k]) % radix[k]);
            if (A > 0 && k == nant - 1) idx = (A == 1) ? 0 : (uint32_t)(((uint64_t)idx * (U - 1)) / (A - 1));
            uidx[(size_t)k * R + r] = idx;
        }
        rconc[r] = -1500.0 + 3000.0 * orc_rand_unit(&s);
        c = total ? (c + step) % total : c + step;
    }
}

/* ------------------------------------------------------------------------- */
/* batched distance over the device layout rb[E][nant+1][maxR]                 */
/* ------------------------------------------------------------------------- */
void orc_batch_rule_distance(int E, int nant, int U, int maxR, const double *u, const double *ve,
                             const double *rb, const int32_t *nrules, const double *x,
                             double *dists, int32_t *hit, int nthreads)
{
    double udivs[ORC_MAX_NANT];
    for (int k = 0; k < nant; k++) udivs[k] = (u[k * U + (U - 1)] - u[k * U]) / (U - 1);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int e = 0; e < E; e++) {
        const double *base = rb + (size_t)e * (nant + 1) * maxR;
        double q[ORC_MAX_NANT];
        for (int k = 0; k < nant; k++) q[k] = ve[k * U + orc_snap(u + k * U, U, x[(size_t)e * nant + k], udivs[k])];
        int h = -1;
        const int R = nrules[e];
        for (int r = 0; r < R; r++) {
            double d0 = q[0] - base[r];
            double acc = d0 * d0;
            for (int k = 1; k < nant; k++) {
                double d = q[k] - base[(size_t)k * maxR + r];
                double sq = d * d;
                acc = acc + sq;
            }
            double dist = sqrt(acc);
            if (dists) dists[(size_t)e * maxR + r] = dist;
            if (h < 0 && dist == 0.0) h = r;
        }
        hit[e] = h;
    }
    (void)nthreads;
}

/* ------------------------------------------------------------------------- */
/* flat accessors for the ctypes binding used by tests/ and bench.py           */
/* ------------------------------------------------------------------------- */
orc_frirl *orc_frirl_new(int env, int trig_mode, int maxR)
{
    orc_frirl *fr = (orc_frirl *)malloc(sizeof(*fr));
    orc_frirl_config(fr, env);
    fr->trig_mode = trig_mode;
    if (maxR > 0) fr->maxR = maxR;
    if (orc_frirl_init(fr) != 0) { free(fr); return NULL; }
    return fr;
}
void orc_frirl_delete(orc_frirl *fr) { if (fr) { orc_frirl_deinit(fr); free(fr); } }
orc_five *orc_frirl_frb(orc_frirl *fr) { return fr->frb; }
double *orc_frirl_actconc(orc_frirl *fr) { return fr->actconc; }
double *orc_frirl_action_vevalues(orc_frirl *fr) { return fr->action_vevalues; }
const orc_dim *orc_frirl_dim(orc_frirl *fr, int k) { return (k < fr->nstates) ? &fr->statedims[k] : &fr->actiondim; }
int orc_frirl_nstates(orc_frirl *fr) { return fr->nstates; }
int orc_frirl_nactions(orc_frirl *fr) { return fr->actiondim.values_len; }
double orc_frirl_get_fus(orc_frirl *fr) { return fr->fus_is_rule_inserted; }
void orc_frirl_set_fus(orc_frirl *fr, double v) { fr->fus_is_rule_inserted = v; }
void orc_frirl_set_max_episodes(orc_frirl *fr, int n) { fr->max_episodes = n; }
void orc_frirl_set_max_steps(orc_frirl *fr, int n) { fr->max_steps = n; }
void orc_frirl_set_values_def(orc_frirl *fr, int k, double v) { fr->statedims[k].values_def = v; }   /* episode start state (frirl_episode.c:46-48) */
uint64_t orc_frirl_hash(orc_frirl *fr) { return fr->step_hash; }
long orc_frirl_total_steps(orc_frirl *fr) { return fr->total_steps; }
unsigned orc_frirl_episode_num(orc_frirl *fr) { return fr->episode_num; }
int orc_frirl_ep_steps(orc_frirl *fr) { return fr->ep_total_steps; }
double orc_frirl_ep_reward(orc_frirl *fr) { return fr->ep_total_value; }
void orc_frirl_hparams(orc_frirl *fr, double *out8)
{
    out8[0] = fr->alpha; out8[1] = fr->gamma; out8[2] = fr->qdiff_pos_boundary; out8[3] = fr->qdiff_neg_boundary;
    out8[4] = fr->weight_significant; out8[5] = (double)fr->skip_rules; out8[6] = fr->reward_good_above; out8[7] = fr->qdiff_final_tolerance;
}

/* whole construct-mode demo: returns 1 when the rule base converged */
int orc_demo_run(int env, int trig_mode, const char *rb_path, uint64_t *hash, long *steps, int *episodes, int *R)
{
    orc_frirl *fr = orc_frirl_new(env, trig_mode, 0);
    if (!fr) return -1;
    int ok = orc_sequential_run(fr, 0);
    if (rb_path) orc_save_rb_text(fr, rb_path);
    if (hash) *hash = fr->step_hash;
    if (steps) *steps = fr->total_steps;
    if (episodes) *episodes = (int)fr->episode_num;
    if (R) *R = fr->frb->R;
    orc_frirl_delete(fr);
    return ok;
}
void orc_frirl_set_trace(orc_frirl *fr, void (*cb)(orc_frirl *, int, double, const double *, const double *, void *))
{
    fr->trace = cb;
}

/* ------------------------------------------------------------------------- */
/* rule-base reduction                                          */
/* ------------------------------------------------------------------------- */
/* src/frirl/frirl_sequential_run.c:170-350, strategies 1 (drop the smallest |Q| first) and 2 (largest |Q|
 * first): remove a candidate rule, replay one greedy episode without updates (reduction_state = 1,
 * frirl_episode.c:155); keep the removal if the episode still succeeds with the same step count and a reward
 * within reduction_reward_tolerance (0.0 in every demo), otherwise restore the rule base and mark the rule as
 * important (its shadow consequent becomes NaN).  The reference restores through a temporary .bin file whose
 * loader re-adds every rule (frirl_utils.c:253-276); the in-memory snapshot below re-adds them the same way. */
static void orc_episode_noupdate(orc_frirl *fr)
{
    const int ns = fr->nstates, n = ns + 1;
    double states[ORC_MAX_NANT], cur_states[ORC_MAX_NANT], q_ant[ORC_MAX_NANT], cur_q_ant[ORC_MAX_NANT];
    for (int i = 0; i < ns; i++) q_ant[i] = states[i] = fr->statedims[i].values_def;
    fr->ep_total_value = 0; fr->ep_total_steps = 0;
    unsigned ai = orc_e_greedy(fr, states);
    q_ant[ns] = fr->actiondim.values[ai];
    for (int step = 1; step <= fr->max_steps; step++) {
        orc_env_do_action(fr, q_ant[ns], states, cur_states);
        orc_env_get_reward(fr, cur_states, &fr->reward_value, &fr->success);
        fr->ep_total_value += fr->reward_value;
        orc_env_quantize(fr, cur_states, cur_q_ant);
        unsigned pa = orc_e_greedy(fr, cur_q_ant);
        cur_q_ant[ns] = fr->actiondim.values[pa];
        for (int i = 0; i < ns; i++) states[i] = cur_states[i];
        for (int i = 0; i < n; i++) q_ant[i] = cur_q_ant[i];
        fr->ep_total_steps++;
        if (fr->success == 1) break;
    }
}

/* frirl_test_run's episode (src/frirl/frirl_test_run.c:20-86): one greedy roll-out, rule base untouched */
void orc_episode_eval(orc_frirl *fr) { orc_episode_noupdate(fr); }

int orc_reduce_run(orc_frirl *fr, int strategy, double reward_tolerance)
{
    orc_five *f = fr->frb;
    const int n = f->nant, maxR = fr->maxR;
    double *tmp_rconc = (double *)malloc(sizeof(double) * maxR), *prev_rconc = (double *)malloc(sizeof(double) * maxR);
    double *snap_rant = (double *)malloc(sizeof(double) * (size_t)maxR * n), *snap_rconc = (double *)malloc(sizeof(double) * maxR);
    int snap_R = 0;
    memcpy(tmp_rconc, f->rconc, sizeof(double) * maxR);
    memcpy(prev_rconc, f->rconc, sizeof(double) * maxR);
    const int iterations = f->R + 1;
    double prev_reward = fr->ep_total_value;
    unsigned mindex = 0;
    int redend = 0;
    orc_episode_noupdate(fr);                                   /* :196-197 */
    const int steps_incremental = fr->ep_total_steps;
    for (fr->episode_num = 1; (int)fr->episode_num <= iterations; fr->episode_num++) {
        orc_episode_noupdate(fr);
        if (fr->episode_num > 1) {                              /* :210-246 */
            double diff = prev_reward - fr->ep_total_value;
            if (fr->ep_total_value > fr->reward_good_above && fr->ep_total_steps == steps_incremental && fabs(diff) <= reward_tolerance) {
                prev_reward = fr->ep_total_value;
            } else {
                memcpy(tmp_rconc, prev_rconc, sizeof(double) * maxR);
                tmp_rconc[mindex] = 0.0 / 0.0;
                while (f->R) f->R--;                            /* reload: numofrules = 0, then add every saved rule */
                for (int r = 0; r < snap_R; r++) orc_add_rule(f, snap_rant + (size_t)r * n, snap_rconc[r]);
                fr->fus_is_rule_inserted = 0;
            }
        } else {
            prev_reward = fr->ep_total_value;
        }
        /* next candidate: smallest (strategy 1, :253-262) or largest (strategy 2, :291-300) |shadow Q|; NaN = important */
        double mvalue = fabs(tmp_rconc[0]);
        mindex = 0;
        for (int j = 1; j < f->R; j++) {
            int better = (strategy == 1) ? (mvalue > fabs(tmp_rconc[j])) : (mvalue < fabs(tmp_rconc[j]));
            if (better || (mvalue != mvalue && tmp_rconc[j] == tmp_rconc[j])) { mvalue = fabs(tmp_rconc[j]); mindex = (unsigned)j; }
        }
        if (mvalue != mvalue) redend = 1;                       /* every remaining rule is important */
        else {
            snap_R = f->R;
            memcpy(snap_rant, f->rant, sizeof(double) * (size_t)snap_R * n);
            memcpy(snap_rconc, f->rconc, sizeof(double) * snap_R);
            memcpy(prev_rconc, tmp_rconc, sizeof(double) * maxR);
            for (int k = (int)mindex; k < f->R - 1; k++) tmp_rconc[k] = tmp_rconc[k + 1];
            orc_remove_rule(f, mindex);
        }
        if (redend) break;
    }
    free(tmp_rconc); free(prev_rconc); free(snap_rant); free(snap_rconc);
    return f->R;
}
