set -e
timeout -k 10 300 python -m pytest tests/test_hip_lanes.py -m gpu -x -q > gpurun_out/lanes_test.log 2>&1; tail -3 gpurun_out/lanes_test.log
export LANES=1
for w in 2 3 4; do
  export FRIRL_HIP_LANES_WPE=$w
  for e in mountaincar cartpole acrobot; do echo "wpe=$w"; timeout -k 10 120 python tools/learn_bench.py $e 8192 2>&1 | grep -v amdgpu.ids; done
  echo "wpe=$w"; timeout -k 10 120 python tools/learn_bench.py mountaincar 65536 2>&1 | grep -v amdgpu.ids
  echo "wpe=$w"; timeout -k 10 120 python tools/learn_bench.py acrobot 65536 2>&1 | grep -v amdgpu.ids
done
