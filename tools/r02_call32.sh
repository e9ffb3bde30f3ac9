#!/bin/bash
# all-actions-in-registers sweep (sweep_gba_many) for 9..24 actions: parity, then A/B against the action-parallel form (no_many = 1)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_hip_q.py tests/test_hip_sarsa.py tests/test_hip_cfg3.py tests/test_full_size.py tests/test_hip_train.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_step_suite32.log 2>&1 || { tail -40 gpurun_out/r02_step_suite32.log; exit 1; }
tail -n 2 gpurun_out/r02_step_suite32.log
for rep in 1 2; do
  timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 1,0 2>&1 | grep -v amdgpu
done
FRIRL_HIP_NO_UIDX=1 timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 1,0 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/step_ab.py cfg3_cartpole_32k_x_32k 32768 no_many 1,0 2>&1 | grep -v amdgpu
