#include <stdio.h>
#include <stdlib.h>
#include "frirl_demo.h"
int main(void){ const char *envs[]={"mountaincar","cartpole","acrobot"};
 for(int i=0;i<3;i++){ int ns,U,A,ms; frirl_demo_describe(envs[i],&ns,&U,&A,0,0,0,0,0,0,0,0,&ms); int n=ns+1;
  double *u=malloc(8*n*U),*ve=malloc(8*n*U),*grid=calloc(16*64,8),*av=calloc(32,8),gd[16],vd[16],hp[8]; int gl[16];
  int rc=frirl_demo_describe(envs[i],&ns,&U,&A,u,ve,grid,gl,gd,vd,av,hp,&ms); printf("%s rc %d ns %d U %d A %d ve_last %.6f av0 %.6f\n",envs[i],rc,ns,U,A,ve[n*U-1],av[0]);
  free(u);free(ve);free(grid);free(av);} return 0;}
