#!/bin/bash
# round-2 first GPU call: new bench line, the GPU suite, cfg3 evidence (kernel trace + PMC)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r02_bench1.err
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02_pytest1.log
timeout -k 10 300 tools/profile_rd.sh r02_cfg3 cfg3_cartpole_32k_x_32k > gpurun_out/r02_prof_cfg3.log 2>&1; echo "prof rc=$?"
tail -5 gpurun_out/r02_prof_cfg3.log
