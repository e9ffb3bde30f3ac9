/*
 * ref_merge_harness.c -- golden vectors for the multi-agent rule-base merge from the GENUINE reference.
 *
 * merge_rb, gen_def_states and omp_init are `static` in the reference's src/frirl/frirl_agent.c and their bodies exist only
 * under BUILD_OPENMP.  This translation unit therefore #includes that file where it lies (REF_AGENT_C, given on the command
 * line by oracle/Makefile; nothing is copied) and is compiled with -DBUILD_OPENMP -fopenmp; it is linked against the other
 * reference objects of oracle/_ref WITHOUT their frirl_agent.o.  No reference code is written here: the harness only calls it.
 *
 *   ref_merge_harness <env> <out.jsonl> <master_episodes> <agent_episodes>
 *
 * Flow (the shape of one frirl_omp_run round, frirl_agent.c:294-467, on two agents):
 *   1. the example's own main() initialises the master and hands it over at frirl_run() (renamed on the command line);
 *      the master learns <master_episodes> episodes;
 *   2. omp_init() makes agent 1 of a world of 3: fresh initial rule base, start state moved by gen_def_states();
 *      it learns <agent_episodes> episodes (frirl_episode);
 *   3. merge_rb(agent 1 <- master's rules)   -> record "agent_after";
 *   4. merge_rb(master  <- agent 1's rules)  -> record "master_after".
 * Every record holds the rule base as hex floats.
 *
 *   ref_merge_harness <env> <out.jsonl> omprun <world> <max_episodes>
 *
 * The reference's whole many-agent loop, unmodified: runmode = FRIRL_OMP, omp_set_num_threads(world), frirl_omp_run()
 * (frirl_agent.c:294-385: omp_init of every agent, chunks of FRIRL_AGENT_EPCHUNK - 1 episodes through frirl_sequential_run, the
 * exchange gated by `epended`, until the master stops).  Recorded: the master's rule base as frirl_omp_run leaves it in the
 * caller's frirl_desc.  `world` is chosen so that gen_def_states stays inside the initial rule list (world - 2 must not divide
 * the 2^nant corner rules: 5 for every demo).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include REF_AGENT_C

int ref_main_mountaincar(int, char **);
int ref_main_cartpole(int, char **);
int ref_main_acrobot(int, char **);

static FILE *g_fp;
static int g_master_eps, g_agent_eps;
static int g_world, g_max_episodes;          /* omprun mode when g_world > 0 */

static void jd(FILE *fp, double v) { fprintf(fp, "\"%a\"", v); }
static void jarr(FILE *fp, const char *key, const double *v, int n)
{
    fprintf(fp, "\"%s\":[", key);
    for (int i = 0; i < n; i++) { if (i) fputc(',', fp); jd(fp, v[i]); }
    fputc(']', fp);
}
static void emit_rb(const char *kind, struct frirl_desc *fr)
{
    struct FIVERB *f = fr->fiverb;
    fprintf(g_fp, "{\"k\":\"%s\",\"R\":%d,", kind, f->numofrules);
    jarr(g_fp, "rant", f->rant, f->numofrules * f->numofantecedents); fputc(',', g_fp);
    jarr(g_fp, "rconc", f->rconc, f->numofrules);
    fprintf(g_fp, "}\n");
}

/* the examples call frirl_run(frirl, verbose); here it is this function */
void harness_run(struct frirl_desc *fr, int verbose)
{
    (void)verbose;
    fr->verbose = 0;
    if (g_world > 0) {
        fr->runmode = FRIRL_OMP;
        fr->max_episodes = g_max_episodes;
        fr->construct_rb = 1; fr->reduce_rb = 0;      /* mountaincar ships reduce-only (mountaincar.c:240-242); construct mode for all */
        omp_set_dynamic(0);
        omp_set_num_threads(g_world);
        frirl_omp_run(fr);
        emit_rb("master_final", fr);
        return;
    }
    for (int e = 0; e < g_master_eps; e++) frirl_episode(fr);
    emit_rb("master_before", fr);
    struct frirl_desc ag;
    omp_init(fr, &ag, 1, 3);
    {
        double vd[8];
        for (unsigned i = 0; i < ag.statedims_len; i++) vd[i] = ag.statedims[i].values_def;
        fprintf(g_fp, "{\"k\":\"agent_start\","); jarr(g_fp, "values_def", vd, ag.statedims_len); fprintf(g_fp, "}\n");
    }
    for (int e = 0; e < g_agent_eps; e++) frirl_episode(&ag);
    emit_rb("agent_before", &ag);
    merge_rb(&ag, 0, fr->fiverb->rant, fr->fiverb->rconc, fr->fiverb->numofrules);
    emit_rb("agent_after", &ag);
    merge_rb(fr, 1, ag.fiverb->rant, ag.fiverb->rconc, ag.fiverb->numofrules);
    emit_rb("master_after", fr);
    omp_deinit(&ag);
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: ref_merge_harness <env> <out.jsonl> <master_episodes> <agent_episodes>\n"); return 2; }
    g_fp = fopen(argv[2], "w");
    if (!g_fp) { perror("open"); return 1; }
    if (!strcmp(argv[3], "omprun")) {
        if (argc < 6) { fprintf(stderr, "usage: ref_merge_harness <env> <out.jsonl> omprun <world> <max_episodes>\n"); return 2; }
        g_world = atoi(argv[4]); g_max_episodes = atoi(argv[5]);
    } else {
        g_master_eps = atoi(argv[3]); g_agent_eps = atoi(argv[4]);
    }
    if (chdir("/tmp") != 0) return 1;
    char *av[] = { "ref", "-q", NULL };
    if (g_world > 0) fprintf(g_fp, "{\"k\":\"hdr\",\"env\":\"%s\",\"world\":%d,\"max_episodes\":%d}\n", argv[1], g_world, g_max_episodes);
    else fprintf(g_fp, "{\"k\":\"hdr\",\"env\":\"%s\",\"master_episodes\":%d,\"agent_episodes\":%d}\n", argv[1], g_master_eps, g_agent_eps);
    if (!strcmp(argv[1], "mountaincar")) ref_main_mountaincar(2, av);
    else if (!strcmp(argv[1], "cartpole")) ref_main_cartpole(2, av);
    else if (!strcmp(argv[1], "acrobot")) ref_main_acrobot(2, av);
    else return 2;
    fclose(g_fp);
    return 0;
}
