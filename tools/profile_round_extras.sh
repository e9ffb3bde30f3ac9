#!/bin/bash
# kernel traces (rocprofv3 --kernel-trace --stats) of the learning-regime and evaluation-mode kernels:
#   lane-group learning (mountaincar x 65536, acrobot x 65536), shared-rule-base evaluation, roll-outs + speculative reduction
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_extras
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, script, args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
}
run learn_mc64k "$REPO/tools/learn_bench.py" mountaincar 65536
run learn_ac64k "$REPO/tools/learn_bench.py" acrobot 65536
run learn_cp8k "$REPO/tools/learn_bench.py" cartpole 8192
run shared_eval "$REPO/tools/shared_bench.py"
run reduce "$REPO/tools/reduce_bench.py"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
print("# rocprofv3 --kernel-trace summaries: learning-regime and evaluation-mode kernels (round 1)\n")
for name in ("learn_mc64k", "learn_ac64k", "learn_cp8k", "shared_eval", "reduce"):
    dur = collections.defaultdict(list); meta = {}
    for p in glob.glob(f"{out}/{name}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"]
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = (r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
    tot = sum(sum(v) for v in dur.values()) or 1
    print(f"## {name}\n")
    print("```\n" + "\n".join(l for l in open(f"{out}/{name}.json").read().strip().splitlines() if l.startswith("{")) + "\n```\n")
    print("| kernel | calls | total ms | share | avg us | max us | VGPR | LDS | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:7]:
        m = meta[k]
        short = k.replace("frirl::", "").split("(")[0][:90]
        print(f"| `{short}` | {len(v)} | {sum(v)/1e6:.2f} | {100*sum(v)/tot:.1f}% | {sum(v)/len(v)/1e3:.1f} | {max(v)/1e3:.1f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} |")
    print(f"\nsum of kernel time {tot/1e6:.1f} ms\n")
PY
