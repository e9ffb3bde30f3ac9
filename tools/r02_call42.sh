#!/bin/bash
# A/B: lane-group kernel before (oldlanes) / after the batch-staged branch-free rule loop
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
for rep in 1 2; do
for lib in libfrirl_hip_oldlanes.so libfrirl_hip.so; do
  echo "== $lib"
  for spec in "acrobot 8192" "acrobot 65536" "mountaincar 8192" "mountaincar 65536" "cartpole 8192"; do
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 300 python tools/learn_bench.py $spec 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['agents'], '%.3f s' % d['wall_s'], '%.3e env-steps/s' % d['env_steps_per_s'])"
  done
done
done
