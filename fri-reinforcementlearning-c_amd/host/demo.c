/*
 * demo.c -- the three demo applications driven through the drop-in C API (one binary, --env NAME).
 *
 * Own driver for machines without the reference tree (the GPU box): frirl_demo_setup() (demo_envs.c) fills
 * struct frirl_desc with the hyper-parameters of the reference's examples and host callbacks implementing
 * the same dynamics with libm trig; then frirl_init / frirl_run (construct mode) and the <env>.frirlrb.txt/.bin
 * dumps, exactly the sequence of the reference's examples.  The reference's own, unchanged example
 * sources also compile and link against these headers and this library.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "frirl.h"
#include "frirl_types_def.h"
#include "frirl_app_helpers.h"
#include "frirl_demo.h"

int main(int argc, char **argv)
{
    struct frirl_desc fr = frirl_desc_default;
    const char *env = "mountaincar";
    char name[128];
    int i, max_episodes = 0, fargc = 0, reduce = 0, agents = 0, gpus = -1, merge = 0;
    const char *load_bin = NULL, *save_bin = NULL;
    char *fargv[16];
    fargv[fargc++] = argv[0];
    for (i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--env") && i + 1 < argc) env = argv[++i];
        else if (!strcmp(argv[i], "--max-episodes") && i + 1 < argc) max_episodes = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--agents") && i + 1 < argc) agents = atoi(argv[++i]);      /* batched: N independent agents on the GPU */
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);          /* with --agents: shard over G devices (0 = all), RCCL report */
        else if (!strcmp(argv[i], "--merge")) merge = 1;                                       /* with --agents: the agents exchange rule bases (frirl_omp_run) */
        else if (!strcmp(argv[i], "--load") && i + 1 < argc) load_bin = argv[++i];             /* batched: start from a .frirlrb.bin file */
        else if (!strcmp(argv[i], "--save") && i + 1 < argc) save_bin = argv[++i];             /* batched: all agents' rule bases to one .bin */
        else if (!strcmp(argv[i], "--reduce") && i + 1 < argc) reduce = atoi(argv[++i]);   /* construct, then reduce with strategy 1|2 */
        else if (fargc < 15) fargv[fargc++] = argv[i];
    }
    frirl_parse_cmdline(&fr, fargc, fargv);
    if (agents > 0 && merge && gpus >= 0) {       /* rule-base exchange across devices */
        snprintf(name, sizeof name, "%s.multi.merged.frirlrb.txt", env);
        return frirl_demo_multi_merged_run(env, agents, gpus, max_episodes > 0 ? max_episodes : fr.max_episodes, name, 1) >= 0 ? 0 : 3;
    }
    if (agents > 0 && merge) {
        snprintf(name, sizeof name, "%s.merged.frirlrb.txt", env);
        return frirl_demo_merged_run(env, agents, max_episodes > 0 ? max_episodes : fr.max_episodes, name, 1) >= 0 ? 0 : 3;
    }
    if (agents > 0 && gpus >= 0) {
        snprintf(name, sizeof name, "%s.multi.frirlrb.txt", env);
        return frirl_demo_multi_run(env, agents, gpus, max_episodes > 0 ? max_episodes : fr.max_episodes, name, 1) == agents ? 0 : 3;
    }
    if (agents > 0) {
        snprintf(name, sizeof name, "%s.batch.frirlrb.txt", env);
        if (reduce == 1 || reduce == 2) snprintf(name, sizeof name, "%s.batch.reduced%d.frirlrb.txt", env, reduce);
        if (load_bin) return frirl_demo_batch_run_ex(env, agents, 2, reduce, load_bin, save_bin, name, 1) >= 0 ? 0 : 3;
        return frirl_demo_batch_run_ex(env, agents, max_episodes > 0 ? max_episodes : fr.max_episodes, reduce, NULL, save_bin, name, 1) == agents ? 0 : 3;
    }
    if (frirl_demo_setup(&fr, env) != 0) return 2;
    if (max_episodes > 0) fr.max_episodes = max_episodes;
    if (frirl_init(&fr) != 0) { fprintf(stderr, "frirl_init failed\n"); return 1; }
    frirl_run(&fr, 1);
    if (reduce == 1 || reduce == 2) {          /* the reference runs this phase from a saved .bin (mountaincar.c:240-242); same code path */
        snprintf(name, sizeof name, "%s.frirlrb.txt", env);
        frirl_save_rb_to_text_file(&fr, name);
        fr.construct_rb = 0; fr.reduce_rb = 1; fr.reduction_strategy = (unsigned char)reduce;
        frirl_sequential_run(&fr);
        snprintf(name, sizeof name, "%s.reduced%d.frirlrb.txt", env, reduce);
        frirl_save_rb_to_text_file(&fr, name);
        printf("demo %s: reduced to %d rules (strategy %d)\n", env, fr.fiverb->numofrules, reduce);
        frirl_deinit(&fr);
        frirl_demo_release(&fr);
        return 0;
    }
    snprintf(name, sizeof name, "%s.frirlrb.bin", env);
    frirl_save_rb_to_bin_file(&fr, name);
    snprintf(name, sizeof name, "%s.frirlrb.txt", env);
    frirl_save_rb_to_text_file(&fr, name);
    printf("demo %s: episodes %u steps(last) %d rules %d converged %d\n", env, fr.episode_num, fr.reward.ep_total_steps, fr.fiverb->numofrules, fr.epended);
    frirl_deinit(&fr);
    frirl_demo_release(&fr);
    return 0;
}
