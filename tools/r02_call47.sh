#!/bin/bash
# v_mad_u32_u16 index decode (libfrirl_hip.so) vs the compiler sequence (nomad): parity, then the step shapes
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py tests/test_hip_cfg3.py tests/test_hip_q.py tests/test_hip_lanes.py tests/test_hip_shared.py tests/test_hip_merge.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_suite47.log 2>&1 || { tail -n 30 gpurun_out/r02_suite46.log; exit 1; }
tail -n 2 gpurun_out/r02_suite47.log
for rep in 1 2; do
  for lib in libfrirl_hip_nomad.so libfrirl_hip.so; do
    echo "== $lib"
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 2>&1 | grep -v amdgpu
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 0 2>&1 | grep -v amdgpu
  done
done
