set -e
timeout -k 10 400 python -m pytest tests/test_hip_lanes.py tests/test_abi.py -m gpu -x -q > gpurun_out/lanes_test.log 2>&1 || { tail -30 gpurun_out/lanes_test.log; exit 1; }
tail -1 gpurun_out/lanes_test.log
for e in mountaincar cartpole acrobot; do timeout -k 10 120 python tools/learn_bench.py $e 8192 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 120 python tools/learn_bench.py mountaincar 65536 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/learn_bench.py acrobot 65536 2>&1 | grep -v amdgpu.ids
