#!/bin/bash
# legacy single-agent drop-in path: completion-flag polling vs hipStreamSynchronize (FRIRL_HIP_MIRROR_SYNC=1), wall time of the three demos
cd "$GRAFT_REPO_ROOT"
D=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib/frirl_demo
timeout -k 10 600 python -m pytest tests/test_dropin.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_dropin37.log 2>&1 || { tail -n 30 gpurun_out/r02_dropin37.log; exit 1; }
tail -n 2 gpurun_out/r02_dropin37.log
mkdir -p /tmp/d37 && cd /tmp/d37
for env in mountaincar cartpole acrobot; do
  for sync in 1 0 1 0; do
    s=$(date +%s%N); FRIRL_HIP_MIRROR_SYNC=$sync timeout -k 10 120 $D --env $env > out.txt 2>&1; e=$(date +%s%N)
    echo "$env mirror_sync=$sync wall $(( (e - s) / 1000000 )) ms : $(tail -n 1 out.txt)"
  done
done
