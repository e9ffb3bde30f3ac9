#!/bin/bash
# A/B: -mllvm -amdgpu-sched-strategy=max-ilp (maxilp) against the default scheduler, learning + step + evaluation shapes
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
for rep in 1 2; do
for lib in libfrirl_hip.so libfrirl_hip_maxilp.so; do
  echo "== $lib"
  for spec in "acrobot 8192" "mountaincar 8192" "mountaincar 65536" "cartpole 8192"; do
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 300 python tools/learn_bench.py $spec 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['agents'], '%.3f s' % d['wall_s'], '%.3e env-steps/s' % d['env_steps_per_s'])"
  done
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 2>&1 | grep -v amdgpu
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 120 python tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 2>&1 | grep -v amdgpu
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 200 python tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 no_many 0 2>&1 | grep -v amdgpu
done
done
