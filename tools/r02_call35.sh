#!/bin/bash
# two prefetch stages in sweep_gba_q (pd2) against one (libfrirl_hip.so)
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
for rep in 1 2; do
  for lib in libfrirl_hip.so libfrirl_hip_pd2.so; do
    echo "== $lib"
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1,0 2>&1 | grep -v amdgpu
  done
done
