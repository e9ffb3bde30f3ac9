#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_merge.py -m gpu -x -q > gpurun_out/r02_pytest7.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r02_pytest7.log
