#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest3.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r02_pytest3.log
python bench.py --no-learn --no-cpu-baseline > gpurun_out/r02_bench3.json 2> gpurun_out/r02_bench3.err; echo "bench rc=$?"
