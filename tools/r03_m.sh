#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; cd gpurun_out
for n in 1024 8192 65536; do
( time timeout -k 10 300 ../fri-reinforcementlearning-c_amd/lib/frirl_demo --env acrobot --agents $n --merge ) > merge_$n.log 2>&1; tail -6 merge_$n.log | cut -c1-250
done
