/*
 * FIVE.h -- public interface of the FIVE engine (Fuzzy Inference by Interpolation in Vague
 * Environment), MI355X drop-in edition.
 *
 * Source- and layout-compatible with the reference's src/five/FIVE.h:24-103: the same struct
 * FIVERB (field names, types and order; 280 bytes on x86-64 with the trailing avx2_rbsize slot,
 * SURVEY Appendix A) and the same three API generations, so programs written against the
 * reference compile and link unchanged.  The implementation differs: rule distance, conclusion,
 * weights and best-action evaluation run as HIP kernels on a device mirror of the rule base
 * (include/frirl_hip.h); the host arrays below stay valid and host-visible after every call.
 */
#ifndef FIVE_H
#define FIVE_H

#define FIVE_MAX_NUM_OF_UNIVERSES  8

#include "config.h"

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

#endif /* FIVE_H */
