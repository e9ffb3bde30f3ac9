#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_rule_distance.py tests/test_hip_cfg3.py tests/test_full_size.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_pytest2.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02_pytest2.log
python bench.py --no-learn > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err; echo "bench rc=$?"
AB_F64=1 REPS=4 VARIANTS="-1,0,0,-1,0;-1,0,0,-1,1;-1,0,8192,-1,0;-1,0,8192,-1,1;-1,0,2048,-1,0;-1,0,1024,-1,0;-1,4,2048,-1,0" timeout -k 10 200 python tools/ab_rd.py cfg4_acrobot_64k_x_8k_per_gpu > gpurun_out/r02_ab_f64_cfg4.txt 2>&1; echo "ab rc=$?"
AB_F64=1 REPS=4 VARIANTS="-1,0,0,-1,0;-1,0,0,-1,1;-1,0,1024,-1,0;-1,0,2048,-1,0;-1,0,8192,-1,0" timeout -k 10 200 python tools/ab_rd.py cfg2_mountaincar_8k_x_8k > gpurun_out/r02_ab_f64_cfg2.txt 2>&1; echo "ab rc=$?"
