#!/bin/bash
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
for rep in 1 2; do
  for lib in libfrirl_hip.so libfrirl_hip_exp.so; do
    echo "== $lib"
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 step_track 0 2>&1 | grep -v amdgpu
  done
done
