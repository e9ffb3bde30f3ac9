set -e
timeout -k 10 400 python -m pytest tests/test_hip_lanes.py -m gpu -x -q > gpurun_out/lanes_test.log 2>&1 || { tail -30 gpurun_out/lanes_test.log; exit 1; }
tail -3 gpurun_out/lanes_test.log
export LANES=1
for h in 1 2 4; do
  export FRIRL_HIP_LANES_SLICES=$h
  for e in mountaincar cartpole acrobot; do echo "slices=$h"; timeout -k 10 120 python tools/learn_bench.py $e 8192 2>&1 | grep -v amdgpu.ids; done
  echo "slices=$h"; timeout -k 10 120 python tools/learn_bench.py mountaincar 16384 2>&1 | grep -v amdgpu.ids
  echo "slices=$h"; timeout -k 10 120 python tools/learn_bench.py acrobot 32768 2>&1 | grep -v amdgpu.ids
done
