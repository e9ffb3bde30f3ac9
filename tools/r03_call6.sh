#!/bin/bash
# round 3, GPU call 6: whole GPU suite after the kernel trims, learner / roll-out timings, bench line, kernel trace + traffic PMC of the headline
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/c6_pytest.log; rc=$?
cat gpurun_out/c6_pytest.log
[ $rc -eq 0 ] || exit 1
{
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 rep 1024 400 2>/dev/null && \
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 rep 1024 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 8192 rep 1024 400 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 65536 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 1048576 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 2>/dev/null
} 2>&1 | tee gpurun_out/c6_kernels.log
timeout -k 10 600 python bench.py > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err; echo "bench rc=$?"
timeout -k 10 400 tools/profile_bench.sh r03_cfg4 --no-env-steps --no-other-configs --steps 30 > gpurun_out/c6_prof_cfg4.log 2>&1; echo "prof rc=$?"; tail -30 gpurun_out/c6_prof_cfg4.log
