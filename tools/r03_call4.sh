#!/bin/bash
# round 3, GPU call 4: where does the diversified learning run spend its time; PMC of the learner and of the roll-out kernel
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 2>/dev/null | tee gpurun_out/c4_div_stamp.json
STAMP=1 timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 4096 200 2>/dev/null | tee -a gpurun_out/c4_div_stamp.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4_trace_div -- python3 $R/tools/learn2_bench.py acrobot 65536 div 1024 200 > $R/gpurun_out/c4_trace_div.json 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/c4_pmc_learn -- python3 $R/tools/learn2_bench.py acrobot 65536 rep 30000 400 > $R/gpurun_out/c4_pmc_learn.json 2>/dev/null
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/c4_pmc_learn2 -- python3 $R/tools/learn2_bench.py acrobot 65536 rep 30000 400 > $R/gpurun_out/c4_pmc_learn2.json 2>/dev/null
REPS=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/c4_pmc_roll -- python3 $R/tools/rollout_bench.py acrobot 65536 > $R/gpurun_out/c4_pmc_roll.json 2>/dev/null
REPS=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/c4_pmc_roll2 -- python3 $R/tools/rollout_bench.py acrobot 65536 > $R/gpurun_out/c4_pmc_roll2.json 2>/dev/null
cd $R && python3 - <<'PY'
import csv, glob, collections
dur = collections.defaultdict(list)
for p in glob.glob("gpurun_out/c4_trace_div/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "frirl" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:80s} calls {len(v):4d} total {sum(v)/1e3:9.2f} ms  each {[round(x/1e3,2) for x in v[:24]]}")
for tag in ("c4_pmc_learn", "c4_pmc_learn2", "c4_pmc_roll", "c4_pmc_roll2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for p in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "learn_kernel" in r["Kernel_Name"] or "rollout_resident" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][:70]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, c in acc.items():
        print(tag, k, {a: f"{b:.4g}" for a, b in sorted(c.items())}, "dispatches", max(v for (kk, _), v in n.items() if kk == k))
PY
