#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_wave 0,2
  python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 2048 step_wave 0,2 ) > gpurun_out/r02_step_ab2.txt 2>&1
cat gpurun_out/r02_step_ab2.txt | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest6.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02_pytest6.log
