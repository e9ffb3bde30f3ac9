#!/bin/bash
# rule-distance scans with the v_mad_u32_u16 decode (libfrirl_hip.so) vs the compiler sequence (nomad): parity, then the four shapes
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/fri-reinforcementlearning-c_amd/lib
timeout -k 10 900 python -m pytest tests/test_hip_rule_distance.py tests/test_hip_cfg3.py tests/test_full_size.py tests/test_hip_mirror.py -m gpu -x -q > gpurun_out/r02_suite50.log 2>&1 || { tail -n 30 gpurun_out/r02_suite50.log; exit 1; }
tail -n 2 gpurun_out/r02_suite50.log
for rep in 1 2; do
for lib in libfrirl_hip_nomad.so libfrirl_hip.so; do
  echo "== $lib"
  for w in cfg4_acrobot_64k_x_8k_per_gpu cfg2_mountaincar_8k_x_8k cfg3_cartpole_32k_x_32k cfg5_synth16_256k; do
    FRIRL_HIP_LIB_OVERRIDE=$L/$lib VARIANTS="-1,0,0,-1,0" REPS=8 timeout -k 10 200 python tools/ab_rd.py $w 2>&1 | grep -v amdgpu | tail -n 1
  done
done
done
