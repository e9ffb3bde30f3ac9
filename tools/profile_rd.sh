#!/bin/bash
# Profiles the rule-distance leg of bench.py for ONE workload on the GPU box: kernel trace + three separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ/LDS counters; never combined with trace domains).  Output: gpurun_out/prof_<tag>/summary.md
#   tools/profile_rd.sh <tag> <workload> [extra bench.py args]
set -e -o pipefail
TAG=$1; WL=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--workload $WL --no-cpu-baseline --no-learn --no-env-steps --no-other-configs"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS --steps 20 --warmup 3 "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS --steps 4 --warmup 1 "$@" > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS --steps 4 --warmup 1 "$@" > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS --steps 4 --warmup 1 "$@" > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_lds" -- python3 "$REPO/bench.py" $ARGS --steps 4 --warmup 1 "$@" > "$OUT/bench_pmc_lds.json" 2> "$OUT/pmc_lds.err"
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.md"
tail -40 "$OUT/summary.md"
