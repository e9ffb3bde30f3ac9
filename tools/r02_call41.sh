#!/bin/bash
# lane-group kernel with the batch-staged, branch-free rule loop: parity, then learning throughput
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_hip_lanes.py tests/test_hip_train.py tests/test_hip_merge.py tests/test_multi.py -m gpu -x -q > gpurun_out/r02_suite41.log 2>&1 || { tail -n 30 gpurun_out/r02_suite41.log; exit 1; }
tail -n 2 gpurun_out/r02_suite41.log
for env in acrobot mountaincar cartpole; do
  for E in 8192 65536; do
    timeout -k 10 300 python tools/learn_bench.py $env $E 2>&1 | grep -v amdgpu
  done
done
