#!/bin/bash
# parity of the step / Q kernels after the Shepard-series + action-parallel changes, then A/B against HEAD (base)
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 700 python -m pytest tests/test_hip_train.py tests/test_hip_lanes.py tests/test_hip_shared.py tests/test_hip_merge.py -m gpu -x -q > gpurun_out/r02_step_suite31.log 2>&1 || { tail -40 gpurun_out/r02_step_suite31.log; exit 1; }
tail -n 2 gpurun_out/r02_step_suite31.log
for rep in 1 2; do
  for lib in libfrirl_hip_base.so libfrirl_hip.so; do
    echo "== $lib"
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 step_track 0 2>&1 | grep -v amdgpu
  done
done
