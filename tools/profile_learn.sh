#!/bin/bash
# kernel-trace of the real-learning run (tools/learn_bench.py): is the loop GPU-bound or launch-bound?
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_learn
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/learn_bench.py" mountaincar 8192 > "$OUT/learn.json" 2> "$OUT/trace.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(list)
for p in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(f"{k:70s} calls {len(v):6d} total {sum(v)/1e6:8.2f} ms avg {sum(v)/len(v)/1e3:7.2f} us")
print("sum of kernel time %.1f ms" % (tot / 1e6))
print(open(out + "/learn.json").read().strip())
PY
