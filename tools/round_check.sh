set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1 || { tail -30 gpurun_out/gpu_suite.log; exit 1; }
tail -3 gpurun_out/gpu_suite.log
for e in mountaincar cartpole acrobot; do timeout -k 10 120 python tools/learn_bench.py $e 8192 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 120 python tools/learn_bench.py mountaincar 65536 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/learn_bench.py cartpole 65536 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/learn_bench.py acrobot 65536 2>&1 | grep -v amdgpu.ids
LANES=0 timeout -k 10 120 python tools/learn_bench.py acrobot 65536 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
