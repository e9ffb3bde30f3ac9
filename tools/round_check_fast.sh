set -e
timeout -k 10 600 python -m pytest tests/test_hip_lanes.py tests/test_hip_shared.py tests/test_hip_q.py -m gpu -x -q > gpurun_out/gpu_suite_fast.log 2>&1 || { tail -30 gpurun_out/gpu_suite_fast.log; exit 1; }
tail -2 gpurun_out/gpu_suite_fast.log
for e in mountaincar cartpole; do timeout -k 10 120 python tools/learn_bench.py $e 8192 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 120 python tools/learn_bench.py mountaincar 65536 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/learn_bench.py acrobot 65536 2>&1 | grep -v amdgpu.ids
LANES=1 timeout -k 10 120 python tools/learn_bench.py acrobot 8192 2>&1 | grep -v amdgpu.ids
python tools/shared_bench.py 2>&1 | grep -v amdgpu.ids
python tools/shared_bench.py --actions 21 --nant 7 --obs 262144 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/reduce_bench.py 2>&1 | grep -v amdgpu.ids
