#!/bin/bash
# round 3, GPU call 3: whole GPU suite (new: full shapes, logical shards, learner lane groups, NaN tails), then the bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/c3_pytest.log; rc=$?
cat gpurun_out/c3_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; rc=$?
tail -5 gpurun_out/c3_bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/c3_bench.json") if l.startswith("{")][-1])
print("value %.4g frac %.3f gate %s" % (d["value"], d["roofline"]["frac"], d["parity_gate"]))
for k in ("env_steps","learning","learning_diversified","evaluation"):
    v=d.get(k)
    if v: print(k, "%.4g"%v["value"], "frac", (v.get("fp64_issue") or {}).get("frac"), v.get("parity_gate",{}).get("checked"), v.get("wall_s"), v.get("launches"), v.get("vs_replicas"))
for n,o in (d.get("other_configs") or {}).items():
    print(n, "%.4g"%o["value"], "frac %.3f"%o["roofline"]["frac"], o.get("env_steps",{}).get("ms_per_step"))
PY
exit $rc
