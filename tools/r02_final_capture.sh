#!/bin/bash
# round-2 evidence capture on the GPU box: smoke, the bench line, the N = 2 launcher rehearsal (gloo, both ranks on the one GPU),
# kernel traces + PMC passes of the shipped rule-distance kernels at the four BASELINE shapes, the step-kernel trace, the GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r02_smoke.log
( time python bench.py ) > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "bench rc=$?"; tail -4 gpurun_out/r02_bench_final.err
FRIRL_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-learn --no-cpu-baseline --steps 30 --envs 4096 > gpurun_out/r02_bench_gpus2_gloo.json 2> gpurun_out/r02_bench_gpus2_gloo.err; echo "bench --gpus 2 rc=$?"
for spec in "cfg4 cfg4_acrobot_64k_x_8k_per_gpu" "cfg3 cfg3_cartpole_32k_x_32k" "cfg5 cfg5_synth16_256k" "cfg2 cfg2_mountaincar_8k_x_8k"; do
  set -- $spec
  timeout -k 10 300 tools/profile_rd.sh r02_$1_after $2 > gpurun_out/r02_prof_$1_after.log 2>&1; echo "prof $1 rc=$?"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_step/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-learn --steps 10 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_step/bench_trace.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_r02_step/trace.err"; echo "step trace rc=$?"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_pytest_final.log
