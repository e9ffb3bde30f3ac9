#!/bin/bash
# round 3, GPU call 1: resident roll-out kernel -- parity tests, then variants
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_shared.py -x -q 2>&1 | tail -5 > gpurun_out/c1_pytest.log || { cat gpurun_out/c1_pytest.log; exit 1; }
cat gpurun_out/c1_pytest.log
{
for v in "rollout_resident=0" "rollout_slices=4 rollout_wps=1" "rollout_slices=4 rollout_wps=2" "rollout_slices=4 rollout_wps=4" "rollout_slices=16 rollout_wps=2" "rollout_slices=16 rollout_wps=4" \
         "rollout_slices=4 rollout_wps=2 rollout_cap=128" "rollout_slices=4 rollout_wps=2 rollout_cap=256" "rollout_slices=4 rollout_wps=2 rollout_cap=1000" "rollout_slices=16 rollout_wps=2 rollout_cap=128"; do
  timeout -k 10 120 python tools/rollout_bench.py acrobot 65536 $v || exit 1
done
timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 rollout_resident=0 && timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 && \
timeout -k 10 120 python tools/rollout_bench.py cartpole 65536 rollout_resident=0 && timeout -k 10 120 python tools/rollout_bench.py cartpole 65536 && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 8192 rollout_resident=0 && timeout -k 10 120 python tools/rollout_bench.py acrobot 8192 && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 1048576 rollout_resident=0 && timeout -k 10 120 python tools/rollout_bench.py acrobot 1048576
} 2>&1 | tee gpurun_out/c1_rollout.log
python tools/reduce_bench.py 2>&1 | tail -8 | tee gpurun_out/c1_reduce.log
