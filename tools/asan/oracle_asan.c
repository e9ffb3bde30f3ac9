#include <stdio.h>
#include <stdlib.h>
#include "frirl_oracle.h"
int main(void){
  for (int env=0; env<3; env++) {
    uint64_t h; long st; int ep, R;
    int ok = orc_demo_run(env, 0, NULL, &h, &st, &ep, &R);
    printf("env %d ok %d steps %ld R %d\n", env, ok, st, R);
    orc_frirl *fr = orc_frirl_new(env, 1, 0);
    orc_sequential_run(fr, 0);
    int r = orc_reduce_run(fr, 1, 0.0);
    printf("  reduced %d\n", r);
    orc_frirl_delete(fr);
  }
  /* synthetic + batched */
  int nant=5,U=41,R=1000,E=7; double *u=malloc(8*nant*U),*ve=malloc(8*nant*U); orc_synth_tables(nant,U,3,u,ve);
  uint32_t *ui=malloc(4*nant*R); double *rc=malloc(8*R); orc_synth_rules(nant,U,R,3,5,ui,rc);
  double *rb=calloc((size_t)E*(nant+1)*R,8); int32_t *nr=malloc(4*E); double *x=malloc(8*E*nant); double *d=malloc(8*E*R); int32_t *hit=malloc(4*E);
  for(int e=0;e<E;e++){ nr[e]=R-e; for(int k=0;k<nant;k++) for(int r=0;r<R;r++) rb[((size_t)e*(nant+1)+k)*R+r]=ve[k*U+ui[k*R+r]]; for(int k=0;k<nant;k++) x[e*nant+k]=u[k*U+ui[k*R+e]]; }
  orc_batch_rule_distance(E,nant,U,R,u,ve,rb,nr,x,d,hit,2);
  printf("hits %d %d\n", hit[0], hit[6]);
  free(u);free(ve);free(ui);free(rc);free(rb);free(nr);free(x);free(d);free(hit);
  return 0; }
