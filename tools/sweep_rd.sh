#!/bin/bash
# distance-kernel tuning sweep (experiments): env-var hooks in five_rule_distance.hip
cd ${GRAFT_REPO_ROOT:-.}
for nt in 0 1; do for un in 1 2 4 8; do for ch in 1024 2048 4096 8192; do
  r=$(FRIRL_HIP_RD_NT=$nt FRIRL_HIP_RD_UNROLL=$un FRIRL_HIP_RD_CHUNK=$ch python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-env-steps "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['achieved']), round(d['roofline']['avg_launch_ms'],4))")
  echo "nt=$nt unroll=$un chunk=$ch -> $r"
done; done; done
