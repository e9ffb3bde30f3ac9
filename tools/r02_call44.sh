#!/bin/bash
# branch-free Q(s,a) terms (q_pair): parity of everything that sweeps, then the three step shapes
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_suite44.log 2>&1 || { tail -n 40 gpurun_out/r02_suite44.log; exit 1; }
tail -n 2 gpurun_out/r02_suite44.log
for rep in 1 2; do
  timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
  timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1,0 2>&1 | grep -v amdgpu
  timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 0 2>&1 | grep -v amdgpu
done
