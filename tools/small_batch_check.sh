timeout -k 10 500 python -m pytest tests/test_hip_lanes.py -m gpu -x -q 2>&1 | tail -2
for e in acrobot cartpole mountaincar; do for n in 96 1024 2048; do LANES=1 timeout -k 10 200 python tools/learn_bench.py $e $n 2>&1 | grep -v amdgpu.ids | sed 's/"diversified_start": false, //; s/"converged.*//; s/"env_steps": [0-9]*, //'; done; done
