#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py tests/test_multi.py tests/test_dropin.py -m gpu -x -q > gpurun_out/r02_pytest5.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02_pytest5.log
( python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track 0,1
  python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0,1
  python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 step_track 0,1 ) > gpurun_out/r02_step_ab1.txt 2>&1
cat gpurun_out/r02_step_ab1.txt | grep -v amdgpu.ids
