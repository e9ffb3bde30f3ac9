#!/bin/bash
# shared-base kernels without the select around the weight (libfrirl_hip.so) vs HEAD~ (oldshared): parity, evaluation + reduction throughput
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_shared.py tests/test_hip_q.py tests/test_dropin.py -m gpu -x -q > gpurun_out/r02_suite51.log 2>&1 || { tail -n 30 gpurun_out/r02_suite51.log; exit 1; }
tail -n 2 gpurun_out/r02_suite51.log
for rep in 1 2; do
for lib in libfrirl_hip_oldshared.so libfrirl_hip.so; do
  echo "== $lib"
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 200 python tools/shared_bench.py 2>&1 | grep -v amdgpu | tail -n 3
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 200 python tools/shared_bench.py --actions 21 2>&1 | grep -v amdgpu | tail -n 3
  FRIRL_HIP_LIB_OVERRIDE=$L/$lib timeout -k 10 300 python tools/reduce_bench.py 2>&1 | grep -v amdgpu | tail -n 4
done
done
