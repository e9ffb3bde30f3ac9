#!/bin/bash
# round 3, GPU call 9: first-batch prefetch in the learner (main sweep before the env step, spread speculatively)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_hip_learn.py -m gpu -x -q > gpurun_out/c9_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/c9_pytest.log
[ $rc -eq 0 ] || exit 1
{
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 rep 1024 400 2>/dev/null && \
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 8192 rep 1024 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 rep 1024 400 2>/dev/null
} 2>&1 | tee gpurun_out/c9_learn.log
