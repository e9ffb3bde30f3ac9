#!/bin/bash
# A/B of the Shepard-weight series + no-hit fast path (libfrirl_hip.so) against HEAD (base), the series alone (ser) and 5 waves (w5)
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 60 tools/exp/shepard_prec > gpurun_out/r02_shepard_prec.txt 2>&1; cat gpurun_out/r02_shepard_prec.txt
for rep in 1 2; do
  for lib in libfrirl_hip_base.so libfrirl_hip.so libfrirl_hip_ser.so libfrirl_hip_w5.so libfrirl_hip_norot.so; do
    echo "== $lib"
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 step_track 0 2>&1 | grep -v amdgpu
  done
done
