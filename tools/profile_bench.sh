#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + separate PMC passes (never combined with
# trace domains), outputs under gpurun_out/prof_<tag>/.  Usage: tools/profile_bench.sh <tag> [bench args]
set -e -o pipefail
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-learn "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-learn --steps 5 --warmup 1 "$@" > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-learn --steps 5 --warmup 1 "$@" > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
find "$OUT" -name '*.csv' | head -50 > "$OUT/files.txt"
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
