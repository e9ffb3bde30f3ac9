#!/bin/bash
# SQ issue / stall counters (one PMC pass) for any python tool:  tools/pmc_cmd.sh <tag> <script> [args...]
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc" -- python3 "$SCRIPT" "$@" > "$OUT/run.json" 2> "$OUT/run.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for p in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("frirl::", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]:
    w = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k}: dispatches {calls[k]}")
    for n, v in sorted(c.items()):
        print(f"   {n:22s} {v:.4g}  ({100*v/w:.1f}% of wave cycles)  per dispatch {v/max(calls[k],1):.4g}")
PY
