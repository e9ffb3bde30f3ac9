#!/bin/bash
# resident step server of the single-agent drop-in path: demos with and without it (FRIRL_HIP_MIRROR_SERVER=0), then the drop-in suite
cd "$GRAFT_REPO_ROOT"
D=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib/frirl_demo
mkdir -p /tmp/d38 && cd /tmp/d38
for env in mountaincar cartpole acrobot; do
  for srv in 0 1 0 1; do
    s=$(date +%s%N); FRIRL_HIP_MIRROR_SERVER=$srv timeout -k 10 60 $D --env $env > out$srv.txt 2>&1; rc=$?; e=$(date +%s%N)
    echo "$env mirror_server=$srv rc=$rc wall $(( (e - s) / 1000000 )) ms : $(tail -n 1 out$srv.txt)"
    [ $rc -eq 0 ] || { tail -n 5 out$srv.txt; exit 1; }
    cp $env.frirlrb.txt rb$srv.txt
  done
  cmp rb0.txt rb1.txt && echo "$env: rule bases identical with and without the server"
done
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_dropin.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_dropin38.log 2>&1 || { tail -n 30 gpurun_out/r02_dropin38.log; exit 1; }
tail -n 2 gpurun_out/r02_dropin38.log
