#!/bin/bash
# round 3, GPU call 2: hash-hit roll-out + persistent learner: parity tests, then timings
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_shared.py tests/test_hip_learn.py -x -q 2>&1 | tail -15 > gpurun_out/c2_pytest.log; rc=$?
cat gpurun_out/c2_pytest.log
[ $rc -eq 0 ] || exit 1
{
for v in "rollout_resident=0" "" "rollout_wps=1" "rollout_wps=3" "rollout_slices=16 rollout_wps=3" "rollout_cap=128" "rollout_cap=250"; do
  timeout -k 10 120 python tools/rollout_bench.py acrobot 65536 $v 2>/dev/null || exit 1
done
timeout -k 10 120 python tools/rollout_bench.py mountaincar 65536 2>/dev/null && timeout -k 10 120 python tools/rollout_bench.py cartpole 65536 2>/dev/null && \
timeout -k 10 120 python tools/rollout_bench.py acrobot 1048576 2>/dev/null
} 2>&1 | tee gpurun_out/c2_rollout.log
python tools/reduce_bench.py 2>/dev/null | tail -8 | tee gpurun_out/c2_reduce.log
{
timeout -k 10 200 python tools/learn2_bench.py acrobot 8192 rep 4096 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 rep 4096 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 rep 30000 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 rep 4096 400 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 4096 200 2>/dev/null && \
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 div 4096 200 2>/dev/null && \
timeout -k 10 200 python tools/learn_bench.py acrobot 65536 2>/dev/null && \
timeout -k 10 300 python tools/learn_bench.py acrobot 65536 div 2>/dev/null
} 2>&1 | tee gpurun_out/c2_learn.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2_trace -- python3 $GRAFT_REPO_ROOT/tools/rollout_bench.py acrobot 65536 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
dur = collections.defaultdict(list)
for p in glob.glob("gpurun_out/c2_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "frirl" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:90s} calls {len(v):4d} avg {sum(v)/len(v)/1e3:9.1f} us  max {max(v)/1e3:9.1f}")
PY
