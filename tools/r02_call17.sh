#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_uidx 0,1
  python tools/step_ab.py cfg3_cartpole_32k_x_32k 32768 no_uidx 0 ) > gpurun_out/r02_step_ab4.txt 2>&1
cat gpurun_out/r02_step_ab4.txt | grep -v amdgpu.ids
