#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export RD_V2_ONLY=1
for spec in "cfg4 0 256" "cfg4 0 0" "cfg2 0 256" "cfg3 0 256" "cfg3 0 0" "cfg5 0 0"; do
  set -- $spec
  timeout -k 10 200 tools/exp/rd_bench $1 $2 $3 > gpurun_out/r02_rdbench4_$1_pad$3.txt 2>&1; echo "rd_bench $spec rc=$?"
done
