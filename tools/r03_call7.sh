#!/bin/bash
# round 3, GPU call 7: two-wave latency kernel (sweeper + speculative stepper): parity, roll-out and reduction timings, learner plan constant
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 60 python tools/rollout_bench.py acrobot 300 2>/dev/null | tee gpurun_out/c7_first.log || { echo "pair kernel: first small launch failed or hung"; exit 1; }
timeout -k 10 400 python -m pytest tests/test_hip_shared.py tests/test_dropin.py tests/test_hip_learn.py -m gpu -x -q > gpurun_out/c7_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/c7_pytest.log
[ $rc -eq 0 ] || exit 1
{
for v in "" "rollout_pair=0"; do
timeout -k 10 120 python tools/rollout_bench.py acrobot 65536 $v 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 $v 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 2000 $v 2>/dev/null || exit 1
done
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 2>/dev/null
} 2>&1 | tee gpurun_out/c7_kernels.log
python tools/reduce_bench.py 2>/dev/null | tail -8 | tee gpurun_out/c7_reduce.log
FRIRL_HIP_ROLLOUT_PAIR=0 python tools/reduce_bench.py 2>/dev/null | tail -8 | tee gpurun_out/c7_reduce_nopair.log
cd /tmp && export TMPDIR=/tmp
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c7_trace -- python3 $GRAFT_REPO_ROOT/tools/rollout_bench.py acrobot 65536 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob
rows=[]
for p in glob.glob("gpurun_out/c7_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rollout_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3))
rows.sort()
for t,k,d in rows[len(rows)//2:]:
    if d > 20: print(k, "%.1f us" % d)
PY
