#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r02_sweep_cfg4.txt
for pad in 0 256 512 1024 2048 4096 6144; do
  timeout -k 10 120 tools/exp/rd_bench cfg4 0 $pad sweep | grep -v "^==" >> gpurun_out/r02_sweep_cfg4.txt 2>&1
done
: > gpurun_out/r02_sweep_cfg3.txt
for pad in 0 256 2048; do
  timeout -k 10 120 tools/exp/rd_bench cfg3 0 $pad sweep | grep -v "^==" >> gpurun_out/r02_sweep_cfg3.txt 2>&1
done
: > gpurun_out/r02_sweep_cfg2.txt
for pad in 0 256 2048; do
  timeout -k 10 120 tools/exp/rd_bench cfg2 0 $pad sweep | grep -v "^==" >> gpurun_out/r02_sweep_cfg2.txt 2>&1
done
echo done
