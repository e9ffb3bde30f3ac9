#!/bin/bash
# round-2 second GPU call: kernel-variant microbenchmark on the four BASELINE shapes + "before" PMC evidence for cfg4 / cfg5
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in cfg4 cfg2 cfg3 cfg5; do
  timeout -k 10 200 tools/exp/rd_bench $c > gpurun_out/r02_rdbench_$c.txt 2>&1; echo "rd_bench $c rc=$?"
done
timeout -k 10 200 tools/profile_rd.sh r02_cfg4_before cfg4_acrobot_64k_x_8k_per_gpu > gpurun_out/r02_prof_cfg4_before.log 2>&1; echo "prof cfg4 rc=$?"
timeout -k 10 200 tools/profile_rd.sh r02_cfg5_before cfg5_synth16_256k > gpurun_out/r02_prof_cfg5_before.log 2>&1; echo "prof cfg5 rc=$?"
