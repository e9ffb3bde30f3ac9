/*
 * frirl_io.c -- application helpers, rule-base files and command line of the drop-in libfrirl
 * (reference src/frirl/frirl_app_helpers.c, frirl_utils.c, frirl_test_run.c, frirl_imitation.c).
 * Host-only glue; the file formats are what the reference's tests diff, so they are kept exactly.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dropin_internal.h"
#include "frirl_app_helpers.h"
#include "frirl_test.h"

/* reference src/frirl/frirl_app_helpers.c:32-44: symmetric fixed-step grid, upper half mirrored */
void frirl_gen_fixres_arr(fri_float *arr, int len, fri_float div)
{
    int i;
    const fri_float from = -((len - 1) * div) / 2;
    for (i = 0; i < len / 2 + 1; i++) arr[i] = from + div * i;
    for (; i < len; i++) arr[i] = arr[len - 1 - i] * -1;
}

/* reference src/frirl/frirl_utils.c:32-59 */
void frirl_show_rb(struct frirl_desc *frirl)
{
    const struct FIVERB *frb = frirl->fiverb;
    int i, j;
    for (i = 0; i < frb->numofrules; i++) {
        printf("%d. ", i);
        for (j = 0; j < frb->numofantecedents; j++) printf("%.18f ", frb->rant[i * frb->numofantecedents + j]);
        printf("Q: %.18f\n", frb->rconc[i]);
    }
}

void frirl_show_hex_rb(struct frirl_desc *frirl)
{
    const struct FIVERB *frb = frirl->fiverb;
    int i, j;
    for (i = 0; i < frb->numofrules; i++) {
        printf("%d. ", i);
        for (j = 0; j < frb->numofantecedents; j++) printf("%a ", frb->rant[i * frb->numofantecedents + j]);
        printf("Q: %a\n", frb->rconc[i]);
    }
}

/* reference src/frirl/frirl_utils.c:100-144: "%.18f " per antecedent, "%.18f \n" for Q */
int frirl_save_rb_to_text_file(struct frirl_desc *frirl, const char *file_name)
{
    const struct FIVERB *frb = frirl->fiverb;
    FILE *fp = fopen(file_name, "w");
    int i, j;
    if (!fp) { perror("frirl_save_rb_to_text_file"); return -1; }
    for (i = 0; i < frb->numofrules; i++) {
        for (j = 0; j < frb->numofantecedents; j++) fprintf(fp, "%.18f ", frb->rant[i * frb->numofantecedents + j]);
        fprintf(fp, "%.18f \n", frb->rconc[i]);
    }
    fclose(fp);
    return 0;
}

/* reference src/frirl/frirl_utils.c:151-230: int count, then per rule nant doubles + 1 double Q */
int frirl_save_rb_to_bin_file(struct frirl_desc *frirl, const char *file_name)
{
    const struct FIVERB *frb = frirl->fiverb;
    FILE *fp = fopen(file_name, "wb");
    int i;
    if (!fp) { perror("frirl_save_rb_to_bin_file"); return -1; }
    fwrite(&frb->numofrules, sizeof(int), 1, fp);
    for (i = 0; i < frb->numofrules; i++) {
        fwrite(frb->rant + (size_t)i * frb->numofantecedents, sizeof(double), frb->numofantecedents, fp);
        fwrite(frb->rconc + i, sizeof(double), 1, fp);
    }
    fclose(fp);
    return 0;
}

/* reference src/frirl/frirl_utils.c:237-281: replaces the rule base by the file's rules (bounds-checked here) */
int frirl_load_rb_from_bin_file(struct frirl_desc *frirl, const char *file_name)
{
    struct FIVERB *frb = frirl->fiverb;
    FILE *fp = fopen(file_name, "rb");
    double rule[FIVE_MAX_NUM_OF_UNIVERSES + 1];
    int n = 0, r, rc;
    if (!fp) { perror("frirl_load_rb_from_bin_file"); return -1; }
    if (fread(&n, sizeof(int), 1, fp) != 1 || n < 0 || n > frb->maxnumofrules) { fclose(fp); fprintf(stderr, "frirl_load_rb_from_bin_file: bad rule count %d\n", n); return -1; }
    rc = five_hip_mirror_upload(five_dropin_mirror(frb), 0, NULL, NULL);
    if (rc) five_dropin_fatal("frirl_load_rb_from_bin_file", rc);
    frb->numofrules = 0;
    frb->newrant = frb->rant;
    frb->newrconc = frb->rconc;
    frirl->fus_is_rule_inserted = 0;
    for (r = 0; r < n; r++) {
        if (fread(rule, sizeof(double), frb->rulelength, fp) != (size_t)frb->rulelength) { fclose(fp); fprintf(stderr, "frirl_load_rb_from_bin_file: truncated file\n"); return -1; }
        five_add_rule(frb, rule);
    }
    fclose(fp);
    return n * frb->rulelength * (int)sizeof(double);
}

void frirl_print_usage()
{
    printf("FRIRL learning usage:\n\n"
           "    -m --runmode <seq|omp|mpi|test>\n\tMode of operation (default: seq). 'test' needs a rule-base file.\n\n"
           "    -f --rbfile <path>\n\tBinary rule-base file to start from.\n\n"
           "    -r --reductionstrategy <noreduce|default>\n\tRule-base reduction strategy (default: noreduce).\n\n"
           "    -d --draw\n\tEnable visualization.\n\n"
           "    -q --quiet\n\tQuiet mode.\n\n");
}

/* reference src/frirl/frirl_utils.c:344-425: same five switches */
void frirl_parse_cmdline(struct frirl_desc *frirl, int argc, char **argv)
{
    static struct option opts[] = {
        {"runmode", required_argument, 0, 'm'}, {"rbfile", required_argument, 0, 'f'}, {"reductionstrategy", required_argument, 0, 'r'},
        {"draw", no_argument, 0, 'd'}, {"quiet", no_argument, 0, 'q'}, {0, 0, 0, 0}};
    int opt, idx = 0;
    frirl->argc = argc;
    frirl->argv = argv;
    if (argc == 1) return;
    optind = 1;
    while ((opt = getopt_long(argc, argv, "m:f:r:dq", opts, &idx)) > 0) {
        switch (opt) {
            case 'm':
                if (!strcmp(optarg, "seq")) frirl->runmode = FRIRL_SEQ;
                else if (!strcmp(optarg, "omp")) frirl->runmode = FRIRL_OMP;
                else if (!strcmp(optarg, "mpi")) frirl->runmode = FRIRL_MPI;
                else if (!strcmp(optarg, "test")) frirl->runmode = FRIRL_TEST;
                else { printf("Invalid runmode!\n\n"); frirl_print_usage(); exit(-1); }
                break;
            case 'f':
                frirl->rbfile = malloc(strlen(optarg) + 1);
                if (frirl->rbfile) strcpy(frirl->rbfile, optarg);
                break;
            case 'r':
                if (!strcmp(optarg, "noreduce")) { frirl->reduce_rb = 0; frirl->reduction_strategy = FRIRL_REDUCTION_STRATEGY_NOREDUCE; }
                else if (!strcmp(optarg, "default")) { frirl->reduce_rb = 1; frirl->reduction_strategy = FRIRL_REDUCTION_STRATEGY_DEFAULT; }
                else { printf("Invalid reduction strategy!\n\n"); frirl_print_usage(); exit(-1); }
                break;
            case 'd': frirl->visualization = 1; break;
            case 'q': frirl->verbose = 0; break;
            default: frirl_print_usage(); exit(-1);
        }
    }
}

/* reference src/frirl/frirl_test_run.c:20-86: one greedy episode without updates */
void frirl_test_run(struct frirl_desc *frirl)
{
    if (frirl->reduce_rb) { printf("The 'test' mode does not perform reduction. Parameter omitted.\n\n"); frirl_print_usage(); exit(-1); }
    frirl->construct_rb = 0;
    frirl->reduction_state = 1;
    frirl_episode(frirl);
    if (frirl->verbose != 0) printf("Steps:\t%d\nRules:\t%d\nReward:\t%f\n", frirl->reward.ep_total_steps, frirl->fiverb->numofrules, frirl->reward.ep_total_value);
    if (frirl->reward.ep_total_value > frirl->reward_good_above) { printf("\nSuccess!\n"); exit(0); }
    printf("\nInvalid!\n");
    exit(-1);
}

/* reference src/frirl/frirl_app_helpers.c:50-109 */
void frirl_run(struct frirl_desc *frirl, int verbose)
{
    (void)verbose;
    if (frirl->rbfile != NULL) {
        if (frirl_load_rb_from_bin_file(frirl, frirl->rbfile) < 0) { printf("Error while loading the binary rule-base file: %s!\n", frirl->rbfile); exit(-1); }
        printf("Loaded %d rules from binary rule-base file: %s\n", frirl->fiverb->numofrules, frirl->rbfile);
        if (frirl->verbose > 1) frirl_show_rb(frirl);
    } else if (frirl->construct_rb == 0 && frirl->reduce_rb == 1) {
        printf("Incrementally constructed rule-base file is missing, \nplease run the construction process first, \nthen supply the constructed rule-base file! \n(example -f example.frirlrb.bin)\n");
        exit(-1);
    }
    switch (frirl->runmode) {
        case FRIRL_SEQ: frirl_sequential_run(frirl); break;
        case FRIRL_OMP: frirl_omp_run(frirl); break;
        case FRIRL_MPI: frirl_mpi_run(frirl); break;
        case FRIRL_TEST: frirl_test_run(frirl); break;
    }
}

/* visualisation (reference src/gui, GLUT) is outside the hot path: accepted and ignored */
void frirl_visualization_init(struct frirl_desc *frirl) { (void)frirl; }
void frirl_visualization_deinit() {}

/* reference src/frirl/frirl_imitation.c:22-89 is interactive keyboard teaching; without a terminal
 * protocol here every request resolves to "use the greedy action" (key 32, space). */
int getch(void) { return 32; }
void getActionFromTerminal(struct frirl_desc *frirl)
{
    frirl->keyaction = 32;
    frirl->valid_simulation = 1;
}
