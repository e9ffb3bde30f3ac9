#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one PMC pass each, never combined with trace domains) of any python tool, per kernel:
#   tools/pmc_traffic_cmd.sh <tag> <script> [args...]
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmct_$TAG
mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$SCRIPT" "$@" > "$OUT/run_fetch.json" 2> "$OUT/run_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$SCRIPT" "$@" > "$OUT/run_write.json" 2> "$OUT/run_write.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for kind in ("fetch", "write"):
    for p in glob.glob(out + f"/{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("frirl::", "").split("(")[0][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:3]:
    f = c.get("FETCH_SIZE", 0) / max(calls[k]["FETCH_SIZE"], 1) * 1024 * 2      # KiB units; x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md)
    w = c.get("WRITE_SIZE", 0) / max(calls[k]["WRITE_SIZE"], 1) * 1024
    print(f"{k}: per dispatch FETCH {f / 1e9:.3f} GB (corrected x2), WRITE {w / 1e9:.3f} GB  ({calls[k]['FETCH_SIZE']} / {calls[k]['WRITE_SIZE']} dispatches)")
PY
