#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout -k 10 600 python bench.py ) > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err; echo "bench rc=$?"; tail -4 gpurun_out/r03e_bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["frac"], "step ms", d["env_steps"]["ms_per_step"])
for k in ("learning","learning_diversified","learning_diversified_4x","evaluation"):
    x=d[k]; print(k, "%.3g"%x["value"], round(x["wall_s"],4), round(x["fp64_issue"]["frac"],3))
for n,o in d["other_configs"].items():
    print(n, "%.3g"%o["value"], round(o["roofline"]["frac"],3), "step", round(o.get("env_steps",{}).get("ms_per_step",0),3), {k:("%.3g"%v["value"], round(v["fp64_issue_frac_counted_work"],3), v["agents_converged"]) for k,v in (o.get("learning") or {}).items()})
PY
