#!/bin/bash
# round 3, GPU call 8: learner launch plan (cost of a lone wave), smoke, pipelined evaluation
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/c8_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/c8_smoke.log
{
for a in 10 13 16 20 30; do
  timeout -k 10 200 python tools/learn2_bench.py acrobot 65536 div 1024 200 learn_alone=$a 2>/dev/null || exit 1
done
timeout -k 10 200 python tools/learn2_bench.py acrobot 8192 rep 1024 400 learn_alone=13 2>/dev/null
timeout -k 10 200 python tools/learn2_bench.py acrobot 8192 rep 1024 400 learn_alone=20 2>/dev/null
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 div 1024 200 learn_alone=13 2>/dev/null
timeout -k 10 200 python tools/learn2_bench.py mountaincar 65536 div 1024 200 learn_alone=20 2>/dev/null
} 2>&1 | tee gpurun_out/c8_learn.log
timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline --steps 50 > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c8_bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/c8_bench.json") if l.startswith("{")][-1])
for k in ("learning","learning_diversified","evaluation"):
    v=d[k]; print(k, "%.4g"%v["value"], "frac %.3f"%v["fp64_issue"]["frac"], v.get("vs_replicas"), v.get("pipelined"))
PY
