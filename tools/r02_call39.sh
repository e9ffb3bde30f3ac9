#!/bin/bash
# single-agent drop-in path after the batched butterflies + the many-action step kernel: demo wall times (default, resident server, stream sync),
# then the suites touched
cd "$GRAFT_REPO_ROOT"
D=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib/frirl_demo
mkdir -p /tmp/d39 && cd /tmp/d39
for env in mountaincar cartpole acrobot; do
  for mode in "default" "FRIRL_HIP_MIRROR_SYNC=1" "default"; do
    s=$(date +%s%N); if [ "$mode" = default ]; then timeout -k 10 60 $D --env $env > out.txt 2>&1; else env $mode FRIRL_HIP_MIRROR_TIMES=1 timeout -k 10 60 $D --env $env > out.txt 2>&1; fi; rc=$?; e=$(date +%s%N)
    echo "$env [$mode] rc=$rc wall $(( (e - s) / 1000000 )) ms : $(grep -a '^demo' out.txt) $(grep -a -o 'step server.*' out.txt)"
    [ $rc -eq 0 ] || { tail -n 5 out.txt; exit 1; }
  done
done
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_dropin.py tests/test_hip_mirror.py tests/test_hip_q.py tests/test_hip_cfg3.py tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_hip_lanes.py tests/test_full_size.py -m gpu -x -q > gpurun_out/r02_suite39.log 2>&1 || { tail -n 30 gpurun_out/r02_suite39.log; exit 1; }
tail -n 2 gpurun_out/r02_suite39.log
