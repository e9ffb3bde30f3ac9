#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py tests/test_hip_cfg3.py tests/test_dropin.py -m gpu -x -q > gpurun_out/r02_pytest4.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02_pytest4.log
python bench.py --no-learn --no-cpu-baseline > gpurun_out/r02_bench4.json 2> gpurun_out/r02_bench4.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02_bench4.json') if l.startswith('{')][-1])
e=d['env_steps']; print("cfg4 env ms", e['ms_per_step'])
for k,v in d['other_configs'].items():
    es=v.get('env_steps',{}); print(k, 'env ms', es.get('ms_per_step'))
PY
