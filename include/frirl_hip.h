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
#define FRIRL_HIP_MAX_GRID     64   /* possible rule places per antecedent */
#define FRIRL_HIP_NO_HIT       0xFFFFFFFFu

#define FRIRL_HIP_ENV_MOUNTAINCAR 0   /* reference examples/mountaincar/mountaincar.c */
#define FRIRL_HIP_ENV_CARTPOLE    1   /* reference examples/cartpole/cartpole.c       */
#define FRIRL_HIP_ENV_ACROBOT     2   /* reference examples/acrobot/acrobot.c         */

/* outcome of one SARSA update per environment (frirl_hip_envs.status) */
#define FRIRL_HIP_UPD_INACTIVE   0    /* environment masked out / episode already ended              */
#define FRIRL_HIP_UPD_EXACT      1    /* exact hit: rconc[hit] = qnow + qdiff   (frirl_update_sarsa.c:55)  */
#define FRIRL_HIP_UPD_SPREAD     2    /* weighted spread over rules with w > threshold (K7, :89-120)       */
#define FRIRL_HIP_UPD_INSERTED   3    /* new rule appended (FIVE_add_rule, :373-377)                       */
#define FRIRL_HIP_UPD_SKIPPED    4    /* hit on the just-inserted rule under skip_rules (:61-63)           */
#define FRIRL_HIP_UPD_FULL       5    /* rule base at capacity: append refused, nothing changed            */

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
    uint16_t *uidx;          /* [dev] [E][nant][maxR] universe index of every antecedent (FIVERB.rseqant_uindex,
                                src/five/FIVE.h:55), or NULL.  Every stored antecedent is snapped to its universe
                                (five_add_rule.c:76-81), so rb[e][k][r] == ve[k][uidx[e][k][r]] EXACTLY: when this
                                2-byte mirror is present the scans stream it (2*nant B/rule instead of 8*nant) and
                                look the VE values up in an LDS copy of the tables -- bit-identical results, a
                                quarter of the antecedent traffic.  Appends keep it in sync.  Needs U <= 65536 and
                                nant*U*8 <= 48 KiB (five_hip_rule_distance: <= 150 KiB, one table copy per 1024-thread
                                workgroup); otherwise the f64 columns are streamed. */
} frirl_hip_rulebases;

/* ---- runtime ------------------------------------------------------------------------------- */
const char *frirl_hip_version(void);
const char *frirl_hip_last_error(void);
int frirl_hip_device_count(void);                 /* number of visible gfx950 devices, <0 on error */
int frirl_hip_device_info(int device, char *name, int name_len, int32_t *cus, int64_t *hbm_bytes);

/* Experiment / test switches by name: "no_uidx" (1 = ignore the 16-bit index mirror), "rd_unroll", "rd_chunk", "rd_nt",
 * "rd_persist", "rd_order", "step_wave", "step_track", "lanes_slices", "lanes_wpe", "rollout_group", "rollout_slices", "rollout_resident", "rollout_cap", "rollout_pair", "rollout_wps", "learn_slices", "learn_alone", "learn_persistent", "multi_loopback", "no_many", "mirror_sync".  Their defaults
 * (the shipped configuration) are read ONCE from the matching FRIRL_HIP_<NAME> environment variable, never per launch;
 * results do not depend on any of them (only the kernel variant / launch shape does). */
int frirl_hip_set_option(const char *name, int value);
int frirl_hip_get_option(const char *name, int *value);
/* Which layout the scans stream for a shape when the caller provides the 16-bit index mirror (bench / tests: the bytes
 * a launch moves): 1 = compressed indices + LDS tables, 0 = the f64 columns. */
int five_hip_rule_distance_uses_uidx(int32_t nant, int32_t U);
int frirl_hip_step_uses_uidx(int32_t nant, int32_t U, int32_t maxR, int32_t E);

/* ---- five_rule_distance (reference src/five/five_rule_distance.c:63-295) -------------------
 * For every environment e: ruledists[e][r] = sqrt(sum_k (ve[k][snap(x[e][k])] - rb[e][k][r])^2),
 * k ascending, for r < nrules[e]; hit[e] = lowest r < nrules[e] with distance exactly 0.0, else
 * FRIRL_HIP_NO_HIT.  snap() is the reference's fixed-step nearest-index rule
 * (src/inl/min.inl:71-92).  ruledists may be NULL (index-only form); entries at or beyond nrules[e]
 * rounded up to the next even index are not written (rules are processed in 16-byte pairs).
 *   x         [dev] [E][nant]   observations
 *   ruledists [dev] [E][maxR]   or NULL
 *   hit       [dev] [E]         uint32
 */
int five_hip_rule_distance(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *x,
                           double *ruledists, uint32_t *hit, void *stream);

/* ---- FIVE_vag_concl (reference src/five/FIVEVagConcl.c:64-351, default-flag live path) ------
 * Q value of one observation per rule base: the consequent of the first exact-hit rule, else the
 * Shepard interpolation sum_r wi*Q_r / sum_r wi with wi = 1 / d_r^p (:224-235,302).  p <= 0 selects
 * the reference default p = nant (FIVEInit.c:89-93).
 *   x [dev][E][nant], conc [dev][E], hit [dev][E] (uint32, FRIRL_HIP_NO_HIT when interpolated) */
int five_hip_vag_concl(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *x,
                       double *conc, uint32_t *hit, void *stream);

/* ---- FIVE_vag_concl_weight (reference src/five/FIVEVagConclWeight.c:52-188) ------------------
 * Normalised Shepard weights weights[e][r] = wi_r / sum wi for r < nrules[e]; when the observation
 * hits a rule exactly, hit[e] is that rule and weights[e][*] is NOT written (as the reference).
 *   weights [dev][E][maxR] (16-byte aligned) */
int five_hip_vag_concl_weight(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *x,
                              double *weights, uint32_t *hit, void *stream);

/* ---- frirl_get_best_action (reference src/frirl/frirl_get_best_action.c:31-341) --------------
 * Greedy action per environment: actconc[e][a] = Q(states[e], action a) for the A discrete actions
 * (FIVEVagConcl_FRIRL_BestAct, src/five/FIVEVagConcl_FRIRL_BestAct.c:56-299), best[e] = first
 * maximum (src/inl/max.inl:16-28).  action_ve[a] is the VE value of action a
 * (frirl_desc.possible_actions->vevalues, frirl_init.c:156-158).  nant must be 2..9, A <= 32.
 *   states [dev][E][nant-1], action_ve [dev][A], actconc [dev][E][A], best [dev][E] int32 */
int frirl_hip_get_best_action(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const double *states,
                              const double *action_ve, int A, double *actconc, int32_t *best, void *stream);

/* Agent hyper-parameters and grids (reference struct frirl_desc, src/frirl/frirl_types.h:62-169;
 * defaults src/frirl/frirl_types_def.h:22-77).  Passed by pointer, copied by value into the launch. */
typedef struct frirl_hip_agent {
    double alpha, gamma;                       /* frirl_desc.alpha / .gamma                                */
    double qdiff_pos_boundary;                 /* insert a rule when qdiff > this ...                      */
    double qdiff_neg_boundary;                 /* ... or qdiff < this (frirl_update_sarsa.c:363)           */
    double weight_significant;                 /* rule_weight_considered_significant_for_update (0.05)     */
    int32_t skip_rules;                        /* frirl_desc.skip_rules (MATLAB compatibility, 1 in demos) */
    int32_t p;                                 /* Shepard power, <= 0 -> nant (FIVEInit.c:89-93)           */
    int32_t A;                                 /* number of discrete actions                               */
    int32_t env_kind;                          /* FRIRL_HIP_ENV_*                                          */
    int32_t max_steps;                         /* frirl_desc.max_steps                                     */
    int32_t no_random;                         /* frirl_desc.no_random: 1 = always greedy (every demo)     */
    int32_t grid_len[FRIRL_HIP_MAX_NANT];      /* possible rule places per antecedent (states.., action)   */
    double grid_div[FRIRL_HIP_MAX_NANT];       /* statedims[i].values_div (generic quantiser)              */
    double values_def[FRIRL_HIP_MAX_NANT];     /* statedims[i].values_def: episode start state             */
    const double *grid_values;                 /* [dev] [nant][FRIRL_HIP_MAX_GRID]; row nant-1 = action values */
    const double *action_ve;                   /* [dev] [A] possible_actions->vevalues (frirl_init.c:156-158) */
    double epsilon;                            /* frirl_desc.epsilon: exploration rate when no_random == 0 */
    double reward_good_above;                  /* frirl_desc.reward_good_above   (convergence test)        */
    double qdiff_final_tolerance;              /* frirl_desc.qdiff_final_tolerance (convergence test)      */
    uint64_t seed;                             /* base seed of the per-environment counter-based RNG       */
    int32_t evaluate;                          /* 1 = policy roll-out only: no SARSA update (frirl_desc.reduction_state == 1,
                                                  frirl_episode.c:155; frirl_test_run / the reduction replays)        */
    int32_t debug_flags;                       /* 0; bit 0 (tests only): update_rules always re-sweeps the rule base instead of using the
                                                  candidates tracked during the Q(s,a) sweep -- both give the same bits */
    uint64_t env_id_base;                      /* global id of environment 0 of this batch: RNG streams are keyed by the
                                                  GLOBAL environment id, so trajectories do not depend on the sharding */
} frirl_hip_agent;

/* frirl_test_run's greedy roll-out (reference src/frirl/frirl_test_run.c:66-70 -> frirl_episode with reduction_state == 1,
 * frirl_episode.c:28-194 without the update at :155) for Q environments sharing ONE read-only rule base: lane =
 * environment, whole episodes in one launch (a Shepard power agent->p != nant runs the variants without rule slices).  `agent` as for the episode entry points (env_kind, max_steps, grids, action
 * values / VE points, epsilon-greedy stream keyed by env_id_base + row; alpha/gamma/... unused).
 * Optional "try-remove" view of the rule base (the replays of the rule-base reduction, frirl_sequential_run.c:170-350):
 * rule r carries a candidate slot rule_slot[r] (0..31, 255 = not a candidate); environment q ignores every rule whose
 * slot bit is set in exclude_mask[q] -- bit-identical to running on the rule base compacted by five_remove_rule. */
typedef struct frirl_hip_rollout {
    const double *start_states;    /* [dev] [Q][nant-1] or NULL = agent->values_def                       */
    const uint32_t *exclude_mask;  /* [dev] [Q] or NULL                                                   */
    const uint8_t *rule_slot;      /* [dev] [maxR] or NULL (together with exclude_mask)                   */
    int32_t *steps;                /* [dev] [Q] reward.ep_total_steps                                     */
    double *reward;                /* [dev] [Q] reward.ep_total_value                                     */
    int32_t *success;              /* [dev] [Q] reward.success of the last step, or NULL                  */
    double *final_states;          /* [dev] [Q][nant-1] or NULL                                           */
} frirl_hip_rollout;
int frirl_hip_rollout_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, int32_t Q,
                             const frirl_hip_rollout *ro, void *stream);
/* Rule-base size up to which frirl_hip_rollout_shared runs its LDS-resident, queue-fed form for this shape (csrc/rollout.hip: the
 * whole rule base in LDS, H = 4 / 16 / 64 lanes per environment splitting the rules, finished groups refilled from an in-order
 * queue, long episodes parked and finished by whole waves, exact hits by hash lookup); 0 = the tiled kernel serves the shape. */
int frirl_hip_rollout_resident_rules(int32_t nant, int32_t A, int32_t p, int32_t env_kind);

/* Rule-base reduction (reference frirl_sequential_run.c:170-350, strategies 1 = smallest |Q| first, 2 = largest |Q|
 * first) as a batched "try-remove" on the GPU.  The reference tests ONE candidate per replayed episode: remove it, replay
 * greedily, keep the removal iff the episode still succeeds (reward > agent->reward_good_above) in the same number of
 * steps with |reward change| <= reward_tolerance, else restore it and never try it again.  Consequents do not change
 * during the reduction, so the candidate order is known in advance (stable sort by |Q|), and the replays of the next
 * `depth` candidates can all run at once: one lane per node of the binary accept/reject tree (2^depth - 1 roll-outs per
 * launch through frirl_hip_rollout_shared's exclude masks); the host then walks the tree along the outcomes that
 * actually happened.  Same decisions, same surviving rules in the same order as the sequential loop.  Replays are capped
 * at steps_incremental + 1 steps (a longer episode is rejected anyway, :212).
 *   b      ONE rule base (E == 1); compacted in place (rb columns, nrules[0], uidx if present)
 *   rant   [dev] [nant][maxR] raw antecedents compacted alongside, or NULL
 *   kept   [host] [rules_before] receives the ORIGINAL index of each surviving rule (first rules_after entries), or NULL
 *   depth  candidates per launch, 1..12 (0 = default 10) */
typedef struct frirl_hip_reduce_result {
    int32_t rules_before, rules_after;
    int32_t rounds;              /* kernel launches after the baseline replay                          */
    int32_t rollouts;            /* episodes replayed (speculative ones included)                      */
    int32_t steps_incremental;   /* steps of the un-reduced rule base's episode (:196-198)             */
    int32_t reserved;
    double reward;               /* prev_reward after the last accepted removal                        */
} frirl_hip_reduce_result;
int frirl_hip_reduce_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, double *rant, int strategy,
                            double reward_tolerance, int depth, int32_t *kept, frirl_hip_reduce_result *result, void *stream);

/* Per-environment episode state (reference: fields of frirl_desc + frirl_reward_desc that
 * frirl_episode() carries from step to step, src/frirl/frirl_episode.c:28-194). */
typedef struct frirl_hip_envs {
    double *states;          /* [dev] [E][nant-1] continuous state (fep_ant)                              */
    double *q_ant;           /* [dev] [E][nant]   quantised state + action of the pending update (fep_q_ant) */
    int32_t *fus;            /* [dev] [E] sticky fus_is_rule_inserted (frirl_update_sarsa.c:374,378,73)   */
    int32_t *done;           /* [dev] [E] 1 once the episode ended (success or max_steps)                 */
    int32_t *ep_steps;       /* [dev] [E] reward.ep_total_steps                                           */
    double *ep_reward;       /* [dev] [E] reward.ep_total_value                                           */
    double *rant;            /* [dev] [E][nant][maxR] raw antecedents of the rules (FIVERB.rant, SoA), or NULL */
    int32_t *status;         /* [dev] [E] FRIRL_HIP_UPD_* of the last update, or NULL                     */
    const double *start_states; /* [dev] [E][nant-1] per-environment episode start state, or NULL = agent->values_def
                                   (reference: gen_def_states randomises it per agent, frirl_agent.c:121-139) */
    int32_t *episode;        /* [dev] [E] episodes started so far (RNG stream position), or NULL          */
    double *spread_ant;      /* [dev] [E][nant] antecedents of the last INTERPOLATED update_rules call, or NULL   */
    int32_t *spread_R;       /* [dev] [E] rule count at that call (0 = none since the last frirl_hip_weights_from_spread), or NULL.
                                Together they determine FIVERB.weights as the reference leaves it after learning (the array is
                                only rewritten when FIVE_vag_concl_weight interpolates, frirl_update_sarsa.c:40, FIVEVagConclWeight.c:67-69;
                                antecedents of existing rules never change), without the learning kernels materialising it:
                                frirl_hip_weights_from_spread rebuilds it where the rule-base merge needs it */
} frirl_hip_envs;

/* Per-environment convergence state of the construct loop (reference frirl_sequential_run.c:55-165). */
typedef struct frirl_hip_convergence {
    int32_t *prev_nrules;    /* [dev] [E] rule count after the previous episode                           */
    int32_t *prev_steps;     /* [dev] [E] steps of the previous episode                                   */
    double *prev_reward;     /* [dev] [E] reward of the previous episode                                  */
    double *prev_rconc;      /* [dev] [E][maxR] consequents after the previous episode                    */
    int32_t *converged;      /* [dev] [E] 1 once "RB considered complete" (sticky)                        */
    int32_t *episodes;       /* [dev] [E] episodes run until convergence (counts while not converged)     */
    int32_t *epended;        /* [dev] [E] or NULL: set to 1 by frirl_hip_convergence_update when the CHEAP test alone holds (same #rules,
                                #steps and good reward as the previous episode, frirl_sequential_run.c:83-90 -- `frirl_desc.epended`, which
                                stays set even when the tolerance check then finds a consequent that moved); never cleared by the
                                library's kernels: the many-agent loop clears it at the start of every chunk and gates that round's
                                rule-base exchange with it (frirl_agent.c:338,352) */
} frirl_hip_convergence;

/* ---- one SHARED, read-only rule base, many observations (evaluation of a trained rule base, e.g.
 *      frirl_test_run's policy for many environments at once; reference FIVE_vag_concl / frirl_get_best_action
 *      called Q times on the same FIVERB).  `b` describes ONE rule base (E == 1).  Each lane owns one observation and
 *      walks the rules in index order through LDS-staged tiles, so the Shepard sums are accumulated sequentially in the
 *      reference's own order (FIVEVagConcl.c:224-235); the rule tiles are reused by every observation of the workgroup
 *      (compute-bound, not HBM-bound).
 *   x [dev][Q][nant] / states [dev][Q][nant-1]; outputs as the per-environment forms, one row per observation */
int five_hip_vag_concl_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, int32_t Q, const double *x,
                              double *conc, uint32_t *hit, void *stream);
int frirl_hip_get_best_action_shared(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, int32_t Q, const double *states,
                                     const double *action_ve, int A, double *actconc, int32_t *best, void *stream);

/* ---- FIVE_add_rule (reference src/five/five_add_rule.c:47-95) ---------------------------------
 * Appends one rule per environment where active[e] != 0 (active == NULL: all): rb[e][k][R] =
 * ve[k][snap(rant[e][k])], rb[e][nant][R] = rconc[e], nrules[e]++.  Unlike the reference the
 * capacity is checked: a full rule base is left unchanged and added[e] = 0.
 *   rant [dev][E][nant], rconc [dev][E], active [dev][E] uint8 or NULL, rant_store = envs-style
 *   [dev][E][nant][maxR] or NULL, added [dev][E] int32 or NULL */
int five_hip_add_rule(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const double *rant, const double *rconc,
                      const uint8_t *active, double *rant_store, int32_t *added, void *stream);

/* ---- frirl_update_sarsa (reference src/frirl/frirl_update_sarsa.c:348-385 + update_rules :22-143,
 *      check_possible_states :146-170, CHECK_STATES = 1) ---------------------------------------------
 * One TD step per environment: qdiff = alpha*(reward + gamma*Q(s',a') - Q(s,a)); either a rule is
 * appended at the grid-snapped antecedents or qdiff is written to the exact-hit rule / spread over
 * the rules whose normalised Shepard weight exceeds the threshold.  envs->fus is read and updated;
 * envs->rant / envs->status may be NULL.  active == NULL: every environment.
 *   q_ant [dev][E][nant], reward [dev][E], cur_q_ant [dev][E][nant] */
int frirl_hip_update_sarsa(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                           const frirl_hip_envs *envs, const double *q_ant, const double *reward, const double *cur_q_ant,
                           const uint8_t *active, void *stream);

/* ---- env step: do_action + get_reward + quantize_observations of the three demo environments
 *      (reference examples/<env>/<env>.c; FMA-free portable sin/cos) -------------
 *   action [dev][E] action VALUES, states [dev][E][ns] -> new_states, reward [dev][E],
 *   success [dev][E] int32, q_states [dev][E][ns] quantised new states */
int frirl_hip_env_step(const frirl_hip_agent *agent, int32_t E, int32_t nstates, const double *action, const double *states,
                       double *new_states, double *reward, int32_t *success, double *q_states, void *stream);

/* ---- frirl_episode (reference src/frirl/frirl_episode.c:28-194), batched -----------------------
 * frirl_hip_episode_begin: states = q_states = values_def, first action chosen greedily on the
 * un-quantised default state (:46-48,78), counters cleared, done = 0.
 * frirl_hip_episode_step: ONE environment step for every environment that is not done:
 * do_action, get_reward, quantise, greedy action for the new state (one sweep giving Q(s',a') for
 * all a'), SARSA update, bookkeeping (:86-185).  Fused: one workgroup owns one environment. */
int frirl_hip_episode_begin(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                            const frirl_hip_envs *envs, void *stream);
int frirl_hip_episode_step(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                           const frirl_hip_envs *envs, void *stream);
/* nsteps consecutive frirl_hip_episode_step launches (finished environments are skipped inside the kernel) */
int frirl_hip_episode_steps(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                            const frirl_hip_envs *envs, int32_t nsteps, void *stream);

/* Lane-group form for MANY agents with SMALL rule bases (the demos' learning regime; the reference's frirl_omp_run model
 * of one agent per core, frirl_agent.c:294-325, at GPU width): G = 4 or 8 consecutive lanes own one environment, each
 * lane evaluates its share of the A + 1 conclusions of a step over ALL rules sequentially -- the reference's summation
 * order -- so a step needs no reduction and no barrier (with few agents, H = 2 / 4 / 8 lanes additionally share every
 * conclusion: lane h sums the rules r = h mod H, the partial sums are added in slice order); up to nsteps consecutive steps
 * per launch (same contract as frirl_hip_episode_steps: finished environments sit out, status[e] = update of the last step).  The rule bases are transposed into
 * `workspace` ([dev], >= frirl_hip_lanes_workspace_bytes) on entry and back on exit.  Decisions (actions, hits,
 * inserted rules) and distances are those of the step kernel; interpolated Q agrees to ~1e-15 (different summation
 * order than the tree of the per-environment kernels, same as the reference's).  The default Shepard power (agent->p <= 0 or == nant,
 * FIVEInit.c:89-93) is a compile-time constant of the kernels; any other p runs the run-time-power variants (no rule slices). */
size_t frirl_hip_lanes_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A);
/* 1 when the lane-group form is expected to beat the per-environment kernels for this batch shape: with up to 8 rule
 * slices it did at every size measured (96 ... 65 536 agents of the three demos), so this is 1 for every valid shape */
int frirl_hip_lanes_preferred(int32_t nant, int32_t E, int32_t A);
int frirl_hip_episode_run_lanes(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                                const frirl_hip_envs *envs, int32_t nsteps, void *workspace, size_t workspace_bytes, void *stream);

/* The whole construct loop of frirl_sequential_run (reference src/frirl/frirl_sequential_run.c:55-165) for many agents with small
 * rule bases, persistent: every agent of `live` ([dev] nlive agent ids, NULL = agents 0..nlive-1) runs episode after episode at its
 * own pace -- frirl_episode's loop, the SARSA update and, at each episode's end, the loop's bookkeeping (same rule count, steps and
 * good reward as the previous episode and no consequent moved by qdiff_final_tolerance => "RB considered complete", :83-148; then the
 * snapshot, :68-72) -- until it has converged, has used its budget for this call, or has run max_episodes - 1 episodes; the budget is
 * the WORK of budget_steps steps of an agent with the call's mean rule count (agents with larger rule bases make fewer steps, with
 * smaller ones more, at most 4 x budget_steps: all waves of the launch then finish together; where a call cuts an agent's run does not
 * change what the agent computes)
 * (:51,59).  Nothing in a launch waits for the longest episode of the batch: the reference's many-agent modes diversify the start
 * states (frirl_agent.c:121-139), so agents are never in step.  Between calls the host compacts the agents that are still learning
 * into `live`; the fewer they are, the more lanes each gets (1 ... 64 rule slices per agent: the largest power of two that keeps all of them resident; one lane per agent when more than 65 536 are alive on a 256-CU chip).
 * State: envs->done[e] != 0 on entry means "between two episodes" (set it to 1 for a fresh agent; frirl_hip_convergence_init first);
 * on return done[e] = 1 iff the agent stopped at an episode boundary, ep_steps / ep_reward = the running or last episode,
 * conv->episodes / converged / prev_* as frirl_hip_convergence_update leaves them, status[e] = FRIRL_HIP_UPD_FULL iff an append was
 * refused.  work ([dev][E][2] int64, or NULL) accumulates the rule visits of the fused sweeps (one visit = one rule evaluated for
 * all A + 1 conclusions of a step) and of the extra single-conclusion sweeps (the snapped point; a weighted spread that has to
 * walk the whole rule base -- normally it visits only the handful of rules the fused sweep flagged as possibly significant, which
 * are not counted); steps_total
 * ([dev][E] int64, or NULL) the environment steps.  Needs the 16-bit index mirror.  Covered shapes: frirl_hip_learn_supported
 * (the demos' shapes: mountaincar and acrobot -- 3 actions, universes of <= 64 points -- and cartpole -- 21 actions, <= 1024 points: its
 * rules are walked twice per step, 11 conclusions each; other shapes: frirl_hip_episode_run_lanes).
 * Decisions follow the oracle exactly on the demos (tests/test_hip_learn.py); interpolated Q within the 1e-6 contract (per-lane sums in
 * descending rule order, slices added in butterfly order). */
int frirl_hip_learn_supported(int32_t nant, int32_t U, int32_t A, int32_t p, int32_t env_kind);
/* How many of the `nlive` agents that are still learning the next frirl_hip_learn_run should take (*agents_per_launch <= nlive) and
 * with how many lanes each (*slices): the launch that fills the chip at the lane-group size with the best product of occupancy and
 * rule-work share (mean_rules: mean rule count of the live agents, 0 = unknown).  When fewer agents fit than are alive, the caller
 * rotates: the ones left out go first in the next launch. */
int frirl_hip_learn_plan(int32_t nlive, int32_t mean_rules, int32_t *slices, int32_t *agents_per_launch);
size_t frirl_hip_learn_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A);
int frirl_hip_learn_run(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                        const frirl_hip_convergence *conv, const int32_t *live, int32_t nlive, int32_t budget_steps, int32_t max_episodes,
                        int64_t *work, int64_t *steps_total, void *workspace, size_t workspace_bytes, void *stream);
/* The whole many-agent construct loop (every agent of the batch runs frirl_sequential_run's loop, :55-165, to its end): launch plan,
 * the queue of agents that are still learning (those that had to wait go first), each launch's agents ordered by rule count, and the
 * compaction between launches, all on the device; the host reads one pair of counters per launch.  envs->done[e] != 0 for a fresh agent
 * (see frirl_hip_learn_run).  refused ([dev][E] bytes or NULL): set to 1 for agents whose rule base refused an append.  on_chunk (or
 * NULL) is called after every launch, the stream idle, with the launch's agents ([dev] ids): the place for a per-chunk report.
 * workspace: [dev] >= frirl_hip_learn_train_workspace_bytes, 16-byte aligned.  *launches_out = number of launches made. */
typedef void (*frirl_hip_learn_chunk_fn)(void *user, int32_t launch, const int32_t *live, int32_t nlive);
size_t frirl_hip_learn_train_workspace_bytes(int32_t nant, int32_t E, int32_t maxR, int32_t A);
int frirl_hip_learn_train(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                          const frirl_hip_convergence *conv, int32_t budget_steps, int32_t max_episodes, int64_t *work, int64_t *steps_total,
                          uint8_t *refused, void *workspace, size_t workspace_bytes, int32_t *launches_out,
                          frirl_hip_learn_chunk_fn on_chunk, void *user, void *stream);

/* Persistent form for SMALL rule bases (the demos' learning regime): one wave keeps its environment's rule base,
 * the tables and the episode state in LDS and runs up to nsteps consecutive steps without a global round trip per
 * step; results are bit-identical to nsteps calls of frirl_hip_episode_step.  lds_rules (<= 1024) is the LDS slab
 * capacity: an environment whose rule base does not fit, or fills the slab while appending, stops with status
 * FRIRL_HIP_UPD_FULL and done == 0 -- continue it with frirl_hip_episode_step(s).  Needs A <= 8 and
 * 2*nant*U*8 <= 16 KiB (mountaincar, acrobot). */
int frirl_hip_episode_run(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent,
                          const frirl_hip_envs *envs, int32_t nsteps, int32_t lds_rules, void *stream);

/* ---- construct-loop bookkeeping of frirl_sequential_run (reference src/frirl/frirl_sequential_run.c:55-165),
 *      per environment: after an episode, converged[e] = same #rules, #steps and reward as the previous episode,
 *      reward > reward_good_above and no consequent moved by >= qdiff_final_tolerance (:83-148); then the
 *      snapshot (prev_*) is refreshed (:68-72).  frirl_hip_convergence_init sets prev_steps = prev_reward = -1
 *      (frirl_init.c:149-150) and snapshots the initial consequents.  Converged environments stay converged;
 *      the caller masks them out of further episodes through envs->done. */
int frirl_hip_convergence_init(const frirl_hip_rulebases *b, int nant, const frirl_hip_convergence *c, void *stream);
int frirl_hip_convergence_update(const frirl_hip_rulebases *b, int nant, const frirl_hip_agent *agent, const frirl_hip_envs *envs,
                                 const frirl_hip_convergence *c, void *stream);
/* after the rule bases were changed outside an episode (frirl_hip_merge_rb): retake the snapshot (rule count, consequents) the next
 * convergence test compares with -- what the reference does at the top of every episode (frirl_sequential_run.c:66-72).  The previous
 * episode's steps and reward are NOT touched: they live in frirl_desc.reward and survive a merge, so an agent can be found complete
 * on its first episode after one.  Converged agents keep their snapshot. */
int frirl_hip_convergence_refresh(const frirl_hip_rulebases *b, int nant, const frirl_hip_convergence *c, void *stream);

/* ---- FIVEVagConcl_FRIRL_BestAct (reference src/five/FIVEVagConcl_FRIRL_BestAct.c:56-299) -------
 * Conclusion from PRECOMPUTED rule distances: first exact hit (d == 0) or Shepard interpolation.
 *   ruledists [dev][E][maxR], conc [dev][E] */
int five_hip_bestact(const frirl_hip_rulebases *b, int nant, int p, const double *ruledists, double *conc, void *stream);

/* =================================================================================================
 * Single rule base, HOST buffers: the form the ANSI-C drop-in library (libfive / libfrirl
 * replacement under fri-reinforcementlearning-c_amd/host/) calls.  A mirror is the device-resident
 * copy of ONE struct FIVERB (tables + rule slab in HBM, E = 1) plus pinned staging buffers and a
 * private stream; every call below takes and returns HOST pointers and returns after the result is
 * in host memory.  Same kernels as the batched entry points above.
 * ================================================================================================= */
typedef struct five_hip_mirror five_hip_mirror;

/* FIVEInit (reference src/five/FIVEInit.c:55-347): u, ve are host [nant][U]; p <= 0 -> nant. */
five_hip_mirror *five_hip_mirror_create(int32_t nant, int32_t U, const double *u, const double *ve, int32_t maxR, int32_t p);
void five_hip_mirror_destroy(five_hip_mirror *m);
/* replace the whole rule base: veval is host SoA [nant][ld] (FIVERB.rseqant_veval rows), rconc host [R] */
int five_hip_mirror_upload(five_hip_mirror *m, int32_t R, const double *const *veval_rows, const double *rconc);
/* FIVE_add_rule (five_add_rule.c:47-95): append one rule given its raw antecedents (host [nant]) */
int five_hip_mirror_add_rule(five_hip_mirror *m, const double *rant, double rconc);
/* five_remove_rule (five_remove_rule.c:29-85): delete rule r, compacting the slab */
int five_hip_mirror_remove_rule(five_hip_mirror *m, uint32_t r);
/* overwrite the consequents with host values (callers that edit FIVERB.rconc on the host) */
int five_hip_mirror_set_rconc(five_hip_mirror *m, const double *rconc, int32_t R);
int five_hip_mirror_get_rconc(five_hip_mirror *m, double *rconc, int32_t R);
int32_t five_hip_mirror_numofrules(const five_hip_mirror *m);
/* five_rule_distance: ruledists host [>= R] (may be NULL); *hit = index or FRIRL_HIP_NO_HIT */
int five_hip_mirror_rule_distance(five_hip_mirror *m, const double *x, double *ruledists, uint32_t *hit);
int five_hip_mirror_vag_concl(five_hip_mirror *m, const double *x, double *conc, uint32_t *hit);
int five_hip_mirror_vag_concl_weight(five_hip_mirror *m, const double *x, double *weights, uint32_t *hit);
int five_hip_mirror_bestact(five_hip_mirror *m, const double *ruledists, double *conc);
/* frirl_get_best_action: states host [nant-1], action_ve host [A], actconc host [A] */
int five_hip_mirror_get_best_action(five_hip_mirror *m, const double *states, const double *action_ve, int32_t A,
                                    double *actconc, uint32_t *best);
/* frirl_update_sarsa: `agent` with HOST grid_values [nant][FRIRL_HIP_MAX_GRID] (action_ve unused);
 * *fus in/out; *status = FRIRL_HIP_UPD_*; when a rule was appended its raw antecedents are written to
 * new_rant (host [nant]) and its consequent to *new_rconc; rconc (host [>= numofrules]) receives the
 * consequents after the update. */
int five_hip_mirror_update_sarsa(five_hip_mirror *m, const frirl_hip_agent *agent, const double *q_ant, double reward,
                                 const double *cur_q_ant, int32_t *fus, int32_t *status, double *new_rant, double *new_rconc,
                                 double *rconc);
/* One greedy environment step of frirl_episode() in ONE launch and ONE synchronisation (reference
 * src/frirl/frirl_episode.c:148-160): greedy action for the new quantised state (frirl_get_best_action) and the SARSA
 * update of the pending (q_ant, reward) pair towards it (frirl_update_sarsa).  Inputs travel as kernel arguments, results
 * are written by the kernel straight into pinned host memory.  cur_q_states host [nant-1], action_ve / action_values host
 * [A]; outputs: *best, actconc host [A], cur_q_ant host [nant] (state + chosen action), then as five_hip_mirror_update_sarsa. */
int five_hip_mirror_greedy_step(five_hip_mirror *m, const frirl_hip_agent *agent, const double *q_ant, double reward,
                                const double *cur_q_states, const double *action_ve, const double *action_values, int32_t A,
                                uint32_t *best, double *actconc, double *cur_q_ant, int32_t *fus, int32_t *status, double *new_rant,
                                double *new_rconc, double *rconc);

/* =================================================================================================
 * Batch object, HOST descriptors: E agents of one problem owned by the library (device memory, stream,
 * convergence state).  The C-level counterpart of the reference's many-agent run modes
 * (frirl_omp_run / frirl_mpi_run, src/frirl/frirl_agent.c:294-467); frirl_hip_batch_train learns without the rule-base exchange,
 * frirl_hip_batch_train_merged / _merge_round (below) with it: every
 * agent owns its rule base and learns independently; only statistics leave the device.  The environment
 * must be one of the built-in kinds (agent.env_kind), because its step() runs on the device.
 * ================================================================================================= */
typedef struct frirl_hip_batch frirl_hip_batch;

typedef struct frirl_hip_batch_desc {
    int32_t nant, U, E, maxR;
    const double *u, *ve;             /* HOST [nant][U]                                                        */
    frirl_hip_agent agent;            /* agent.grid_values: HOST [nant][FRIRL_HIP_MAX_GRID]; agent.action_ve: HOST [A] */
    int32_t R0;                       /* initial rules of every agent (the 2^nant corner rules, frirl_init_rb.c:99-126) */
    const double *rant0;              /* HOST [R0][nant] raw antecedents (AoS, like FIVERB.rant)               */
    const double *rconc0;             /* HOST [R0]                                                             */
    const double *start_states;       /* HOST [E][nant-1] per-agent episode start state, or NULL = agent.values_def */
    int32_t device_select;            /* 0 = the calling thread's current device; 1 = the device with ordinal `device`          */
    int32_t device;                   /* device ordinal when device_select == 1; every frirl_hip_batch_* call switches to it   */
} frirl_hip_batch_desc;

/* statistics of frirl_hip_batch_stats(): what the reference prints per episode (frirl_sequential_run.c:77-80), summed */
typedef struct frirl_hip_batch_stats_t {
    double reward_sum, steps_sum, rules_sum, reward_min, reward_max;
    int64_t agents, converged, episodes_max, total_env_steps;
    int64_t full_agents;       /* agents whose rule base is at capacity (numofrules == maxR): further appends are refused
                                  (FRIRL_HIP_UPD_FULL) and those TD updates dropped -- size maxR so that this stays 0 */
} frirl_hip_batch_stats_t;

frirl_hip_batch *frirl_hip_batch_create(const frirl_hip_batch_desc *d);
void frirl_hip_batch_destroy(frirl_hip_batch *b);
/* one episode for every not-yet-converged agent (frirl_episode), then the convergence bookkeeping */
int frirl_hip_batch_episode(frirl_hip_batch *b);
/* frirl_sequential_run's construct loop for all agents: at most max_episodes-1 episodes, stops when every agent's
 * rule base is "considered complete"; *episodes_run receives the number of episodes executed (the most any agent ran).  Shapes the
 * persistent learner covers (frirl_hip_learn_supported) run through frirl_hip_learn_train -- every agent at its own pace; the others
 * episode by episode (frirl_hip_batch_episode).  Same results per agent either way. */
int frirl_hip_batch_train(frirl_hip_batch *b, int32_t max_episodes, int32_t *episodes_run);
int frirl_hip_batch_stats(frirl_hip_batch *b, frirl_hip_batch_stats_t *out);
/* rule base of agent e: *R rules, rant HOST [>= *R][nant] (AoS), rconc HOST [>= *R] (pass NULL to query *R only) */
int frirl_hip_batch_get_rulebase(frirl_hip_batch *b, int32_t e, int32_t *R, double *rant, double *rconc);
/* Rule bases of all agents to / from one file.  The file is E records back to back, each exactly what
 * frirl_save_rb_to_bin_file writes for one agent (reference src/frirl/frirl_utils.c:151-205): int32 numofrules, then per
 * rule nant raw antecedents + the consequent as doubles -- so a file saved by the reference (or by the drop-in library) is
 * a valid one-record file, and the first record of a batch file loads in the reference.  Loading re-adds every rule
 * through FIVE_add_rule's snap (as frirl_load_rb_from_bin_file does, :237-281), clears fus_is_rule_inserted and the
 * convergence state; a file with fewer than E records gives the remaining agents a copy of its LAST record (one trained
 * rule base for every agent).  Unlike the reference's loader it is bounds-checked: a truncated file, a rule count outside
 * 1..maxR or a non-finite value fails with FRIRL_HIP_EINVAL and leaves the batch untouched. */
int frirl_hip_batch_save_rulebases(frirl_hip_batch *b, const char *path);
int frirl_hip_batch_load_rulebases(frirl_hip_batch *b, const char *path, int32_t *records_read);
/* frirl_sequential_run's reduction phase (frirl_sequential_run.c:170-350) for agent e's rule base, in place, through
 * frirl_hip_reduce_shared (speculative batched try-remove); the other agents are untouched */
int frirl_hip_batch_reduce(frirl_hip_batch *b, int32_t e, int strategy, double reward_tolerance, int depth, frirl_hip_reduce_result *result);
/* One round of the reference's multi-agent rule-base exchange (frirl_omp_run, frirl_agent.c:426-462) inside the batch: every agent
 * id >= 1 takes over the master's (agent 0's) rules -- all of them in one launch of frirl_hip_merge_rb --, then the master takes
 * over the rules of agent 1, 2, ... in turn.  Agents whose rule base is complete do not send (:432,:444).  *full_agents (or NULL):
 * agents at capacity afterwards.  Give the agents different start states (frirl_hip_gen_def_states -> desc.start_states). */
int frirl_hip_batch_merge_round(frirl_hip_batch *b, int32_t *full_agents);
/* frirl_omp_run's loop: rounds of chunk - 1 episodes per agent (the reference: FRIRL_AGENT_EPCHUNK = 10), a merge round after each,
 * until the master's rule base is complete or max_episodes - 1 episodes have run */
int frirl_hip_batch_train_merged(frirl_hip_batch *b, int32_t max_episodes, int32_t chunk, int32_t *episodes_run, int32_t *rounds);

/* ---- multi-agent rule-base merge (reference src/frirl/frirl_agent.c:58-117 merge_rb, the body of its BUILD_OPENMP / BUILD_MPI
 *      run modes) ----------------------------------------------------------------------------------------------
 * Every receiver rule base e of the batch (active[e] != 0, or all) takes over the sender's S rules one after the other: with
 * Qr = the receiver's conclusion at the sender rule's antecedents and Qs the sender's consequent, qdiff = Qs - Qr;
 *   qdiff outside [qdiff_neg_boundary, qdiff_pos_boundary]: antecedents snapped to the receiver's rule grid; a free place gets a
 *     NEW rule with 0.5 Q(snapped) + 0.5 Qs, an occupied one moves to 0.9 Q + 0.1 Qs;
 *   else: every receiver rule with normalised Shepard weight > weight_significant is OVERWRITTEN with (0.9 Qr + 0.1 Qs) * weight
 *     (the agent file's own update_rules, :45-53).  `weights` [dev][E][maxR] is the receivers' FIVERB.weights: it persists
 *     between sender rules and calls (an exact hit leaves it untouched, FIVEVagConclWeight.c:67-69); zero it once.
 * The sender rules are read as rant[r * rule_stride + k * dim_stride] (AoS like FIVERB.rant: rule_stride = nant, dim_stride = 1;
 * a row set of frirl_hip_envs.rant: rule_stride = 1, dim_stride = maxR) and rconc[r]; S_dev != NULL: the count is read on the
 * device (e.g. &nrules[sender]).  The sender must not be one of the active receivers.  full[e] = 1 when an append was refused. */
/* FIVERB.weights of every rule base as the reference's learning loop left it: for environments with envs->spread_R[e] > 0 the
 * normalised Shepard weights of envs->spread_ant[e] over the first spread_R[e] rules are written to weights[e][0 .. spread_R[e]) and
 * spread_R[e] is reset to 0; other rows are left as they are (e.g. what the previous merge left). */
int frirl_hip_weights_from_spread(const frirl_hip_tables *t, const frirl_hip_rulebases *b, int p, const frirl_hip_envs *envs, double *weights, void *stream);
typedef struct frirl_hip_sender {
    const double *rant;         /* [dev] */
    int64_t rule_stride, dim_stride;
    const double *rconc;        /* [dev] [S] */
    int32_t S;
    int32_t reserved;
    const int32_t *S_dev;       /* [dev] or NULL */
} frirl_hip_sender;
int frirl_hip_merge_rb(const frirl_hip_tables *t, const frirl_hip_rulebases *b, const frirl_hip_agent *agent, double *rant_store,
                       const frirl_hip_sender *sender, double *weights, const uint8_t *active, int32_t *full, void *stream);
/* gen_def_states (reference frirl_agent.c:121-139), HOST arrays: start_states[id][nant-1] for id < world from the master's rule list
 * master_rant [R][nant] (AoS); agent 0 (and every agent when world < 3: the reference divides by world - 2) keeps values_def. */
int frirl_hip_gen_def_states(const double *master_rant, int32_t R, int32_t nant, int32_t world, const double *values_def, double *start_states);

/* =================================================================================================
 * Many agents over several GPUs of one node, from plain C (the reference's frirl_omp_run / frirl_mpi_run shape,
 * src/frirl/frirl_agent.c:294-467): `total_agents` agents are sharded over `ngpus` visible
 * devices by GLOBAL environment id (frirl_hip_shard: balanced contiguous partition; RNG streams and start states are keyed by
 * the global id, so trajectories do not depend on the sharding), one frirl_hip_batch and one host thread per device.  Learning has no
 * data-path collective: the only exchange is the per-episode report (sums of reward / steps / rules / converged, reward
 * min / max; frirl_sequential_run.c:74-80): RCCL all-reduce over xGMI, one communicator per device in this single process
 * (ncclCommInitAll).  RCCL is loaded at first use (dlopen "librccl.so.1"); the library does not link against it.
 * ================================================================================================= */
typedef struct frirl_hip_multi frirl_hip_multi;
/* rank's slice [*start, *start + *count) of `total` environment ids split over `world` ranks (counts differ by at most 1) */
int frirl_hip_shard(int64_t total, int32_t world, int32_t rank, int64_t *start, int64_t *count);
/* d: as for frirl_hip_batch_create (d->E, d->device* ignored; d->start_states, if given, HOST [total_agents][nant-1]);
 * ngpus <= 0: every visible device */
frirl_hip_multi *frirl_hip_multi_create(const frirl_hip_batch_desc *d, int64_t total_agents, int32_t ngpus);
void frirl_hip_multi_destroy(frirl_hip_multi *m);
/* frirl_sequential_run's construct loop on every device at once; stops when the ALL-REDUCED report says every agent converged */
int frirl_hip_multi_train(frirl_hip_multi *m, int32_t max_episodes, int32_t *episodes_run);
/* The many-agent mode WITH the rule-base exchange across the devices (frirl_omp_run / frirl_mpi_run, frirl_agent.c:424-462): rounds of
 * chunk - 1 episodes per agent, then one merge round -- the master (global agent 0, device 0) is broadcast to every device (RCCL broadcast)
 * and taken over by all other agents, the devices send their agents' rule lists to device 0 (RCCL send / receive) and the master takes
 * them over in GLOBAL agent order -- until the master's rule base is complete or max_episodes - 1 episodes have run.  With one device
 * this is frirl_hip_batch_train_merged.  Give the agents different start states (frirl_hip_gen_def_states -> d->start_states). */
int frirl_hip_multi_train_merged(frirl_hip_multi *m, int32_t max_episodes, int32_t chunk, int32_t *episodes_run, int32_t *rounds);
/* the report of the whole job (all-reduced over the devices) */
int frirl_hip_multi_stats(frirl_hip_multi *m, frirl_hip_batch_stats_t *out);
/* *ngpus devices in use, RCCL version code, per-device shard [start, count) (arrays of >= *ngpus entries, or NULL) */
int frirl_hip_multi_info(const frirl_hip_multi *m, int32_t *ngpus, int32_t *rccl_version, int64_t *shard_start, int64_t *shard_count);
/* rule base of the agent with GLOBAL id `agent` (as frirl_hip_batch_get_rulebase) */
int frirl_hip_multi_get_rulebase(frirl_hip_multi *m, int64_t agent, int32_t *R, double *rant, double *rconc);

#ifdef __cplusplus
}
#endif
#endif /* FRIRL_HIP_H */
