/* frirl_test.h -- evaluation-only run mode (reference src/frirl/frirl_test.h:18). */
#ifndef FRIRL_TEST_H
#define FRIRL_TEST_H
struct frirl_desc;
void frirl_test_run(struct frirl_desc *frirl);
#endif
