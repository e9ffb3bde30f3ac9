#!/bin/bash
# round 3, GPU call 5: staged roll-outs, rotating learner launches, reference round semantics of the merged training
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_shared.py tests/test_hip_merge.py tests/test_multi.py tests/test_hip_learn.py tests/test_dropin.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/c5_pytest.log; rc=$?
cat gpurun_out/c5_pytest.log
[ $rc -eq 0 ] || exit 1
{
for v in "" "rollout_wps=3" "rollout_cap=60" "rollout_cap=120" "rollout_slices=4"; do
  timeout -k 10 120 python tools/rollout_bench.py acrobot 65536 $v 2>/dev/null || exit 1
done
timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 2>/dev/null && timeout -k 10 120 python tools/rollout_bench.py cartpole 65536 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 1048576 2>/dev/null && timeout -k 10 120 python tools/rollout_bench.py acrobot 8192 2>/dev/null
} 2>&1 | tee gpurun_out/c5_rollout.log
python tools/reduce_bench.py 2>/dev/null | tail -8 | tee gpurun_out/c5_reduce.log
{
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 512 200 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 rep 1024 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 div 1024 200 2>/dev/null
} 2>&1 | tee gpurun_out/c5_learn.log
cd /tmp && export TMPDIR=/tmp
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5_trace -- python3 $GRAFT_REPO_ROOT/tools/rollout_bench.py acrobot 65536 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
rows=[]
for p in glob.glob("gpurun_out/c5_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rollout_resident" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3))
rows.sort()
for t,k,d in rows[len(rows)//2:]:
    if d > 20: print(k, "%.1f us" % d)
PY
