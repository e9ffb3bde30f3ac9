#!/bin/bash
# decode before refill (no register hand-over copies): parity, then the step shapes (previous: cfg2 0.229, cfg4 1.96, cfg3/4096 1.707 ms)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py tests/test_hip_cfg3.py tests/test_hip_q.py -m gpu -x -q > gpurun_out/r02_suite48.log 2>&1 || { tail -n 30 gpurun_out/r02_suite48.log; exit 1; }
tail -n 2 gpurun_out/r02_suite48.log
for rep in 1 2 3; do
  timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
  timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 2>&1 | grep -v amdgpu
  timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 0 2>&1 | grep -v amdgpu
done
