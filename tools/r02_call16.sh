#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_uidx 0,1
  python tools/step_ab.py cfg3_cartpole_32k_x_32k 16384 no_uidx 0,1
  python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 no_uidx 0,1
  python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_wave 0,1 ) > gpurun_out/r02_step_ab3.txt 2>&1
cat gpurun_out/r02_step_ab3.txt | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_dropin.py -m gpu -x -q > gpurun_out/r02_pytest8.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_pytest8.log
