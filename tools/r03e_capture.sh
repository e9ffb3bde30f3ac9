#!/bin/bash
# round-3 (final) evidence capture on the GPU box: smoke, the whole GPU suite, the bench line (with parity gates), the N = 2 launcher rehearsal
# (gloo, both ranks on the one GPU), kernel trace + PMC passes of the learner and the roll-out kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r03e_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r03e_smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03e_pytest_gpu.log
( time timeout -k 10 600 python bench.py ) > gpurun_out/r03e_bench_final.json 2> gpurun_out/r03e_bench_final.err; echo "bench rc=$?"; tail -4 gpurun_out/r03e_bench_final.err
FRIRL_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --no-other-configs --steps 30 --envs 4096 > gpurun_out/r03e_bench_gpus2_gloo.json 2> gpurun_out/r03e_bench_gpus2_gloo.err; echo "bench --gpus 2 rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03e_trace_learn -- python3 $R/tools/learn2_bench.py acrobot 65536 rep 512 1000 > $R/gpurun_out/r03e_trace_learn.json 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r03e_pmc_learn -- python3 $R/tools/learn2_bench.py acrobot 65536 rep 512 1000 > $R/gpurun_out/r03e_pmc_learn.json 2>/dev/null
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/r03e_pmc_learn2 -- python3 $R/tools/learn2_bench.py acrobot 65536 rep 512 1000 > $R/gpurun_out/r03e_pmc_learn2.json 2>/dev/null
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03e_trace_roll -- python3 $R/tools/rollout_bench.py acrobot 65536 > $R/gpurun_out/r03e_trace_roll.json 2>/dev/null
REPS=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r03e_pmc_roll -- python3 $R/tools/rollout_bench.py acrobot 65536 > $R/gpurun_out/r03e_pmc_roll.json 2>/dev/null
REPS=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/r03e_pmc_roll2 -- python3 $R/tools/rollout_bench.py acrobot 65536 > $R/gpurun_out/r03e_pmc_roll2.json 2>/dev/null
cd $R && python3 - > gpurun_out/r03e_kernels_summary.md <<'PY'
import csv, glob, collections, json
print("# round 3, final capture: learn_kernel and the roll-out kernels (tools/r03e_capture.sh)\n")
for tag, what in (("r03e_trace_learn", "tools/learn2_bench.py acrobot 65536 rep 512 1000 (the bench's `learning` leg: 43 launches)"), ("r03e_trace_roll", "tools/rollout_bench.py acrobot 65536 (REPS=1: one warm-up + one timed call)")):
    dur = collections.defaultdict(list)
    for p in glob.glob(f"gpurun_out/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "frirl" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"## kernel trace: {what}\n\n| kernel | calls | total ms | avg us | max us |\n|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) > 30: print(f"| `{k}` | {len(v)} | {sum(v)/1e3:.3f} | {sum(v)/len(v):.1f} | {max(v):.1f} |")
    try:
        print("\nrun: `" + open(f"gpurun_out/{tag}.json").read().strip().splitlines()[-1][:700] + "`\n")
    except Exception: pass
print("## PMC (two passes per command; counters summed over the dispatches of a kernel)\n")
for tag in ("r03e_pmc_learn", "r03e_pmc_learn2", "r03e_pmc_roll", "r03e_pmc_roll2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for p in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "learn_kernel" in r["Kernel_Name"] or "rollout_" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][:80]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, c in acc.items():
        if c.get("SQ_WAVE_CYCLES", 1e9) < 1e7 and c.get("SQ_WAVES", 1e9) < 100: continue
        print(f"* `{k}` ({tag}, dispatches {max(v for (kk, _), v in n.items() if kk == k)}): " + ", ".join(f"{a} {b:.4g}" for a, b in sorted(c.items())))
PY
cat gpurun_out/r03e_kernels_summary.md | head -60
