#!/bin/bash
# lane-group kernel with the v_mad_u32_u16 decode (libfrirl_hip.so) vs the compiler sequence (nomad): parity, learning throughput
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_lanes.py tests/test_hip_train.py tests/test_hip_merge.py -m gpu -x -q > gpurun_out/r02_suite49.log 2>&1 || { tail -n 30 gpurun_out/r02_suite49.log; exit 1; }
tail -n 2 gpurun_out/r02_suite49.log
for rep in 1 2; do
for lib in libfrirl_hip_nomad.so libfrirl_hip.so; do
  echo "== $lib"
  for spec in "acrobot 8192" "acrobot 65536" "mountaincar 8192" "mountaincar 65536" "cartpole 8192"; do
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 300 python tools/learn_bench.py $spec 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['env'], d['agents'], '%.3f s' % d['wall_s'], '%.3e env-steps/s' % d['env_steps_per_s'])"
  done
done
done
