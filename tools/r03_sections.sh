#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
WARM_ALL=1 STAMP=1 timeout -k 10 120 python tools/learn2_bench.py acrobot 65536 div 1024 200 > gpurun_out/bal_div.json 2>gpurun_out/bal_err.log; echo rc=$?
WARM_ALL=1 STAMP=1 timeout -k 10 120 python tools/learn2_bench.py acrobot 65536 div 512 200 > gpurun_out/bal_div512.json 2>>gpurun_out/bal_err.log; echo rc=$?
WARM_ALL=1 STAMP=1 timeout -k 10 120 python tools/learn2_bench.py acrobot 65536 div 2048 200 > gpurun_out/bal_div2048.json 2>>gpurun_out/bal_err.log; echo rc=$?
