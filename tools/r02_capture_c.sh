#!/bin/bash
# round-2 final evidence capture: smoke, the bench
# line, step-kernel trace + SQ counters at the cartpole shape, kernel trace + PMC of the headline scan, the GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/r02c_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/r02c_smoke.log
( time python bench.py ) > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; echo "bench rc=$?"; tail -n 4 gpurun_out/r02c_bench.err
timeout -k 10 300 tools/profile_rd.sh r02c_cfg4 cfg4_acrobot_64k_x_8k_per_gpu > gpurun_out/r02c_prof_cfg4.log 2>&1; echo "prof cfg4 rc=$?"
timeout -k 10 300 tools/pmc_cmd.sh r02c_step_cfg3 tools/step_ab.py cfg3_cartpole_32k_x_32k 4096 step_track 0 > gpurun_out/r02c_pmc_step_cfg3.txt 2>&1; echo "pmc step cfg3 rc=$?"
timeout -k 10 300 tools/pmc_cmd.sh r02c_step_cfg4 tools/step_ab.py cfg4_acrobot_64k_x_8k_per_gpu 0 step_track -1 > gpurun_out/r02c_pmc_step_cfg4.txt 2>&1; echo "pmc step cfg4 rc=$?"
timeout -k 10 300 tools/pmc_cmd.sh r02c_step_cfg2 tools/step_ab.py cfg2_mountaincar_8k_x_8k 0 step_track 0 > gpurun_out/r02c_pmc_step_cfg2.txt 2>&1; echo "pmc step cfg2 rc=$?"
mkdir -p gpurun_out/prof_r02c_step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r02c_step/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-learn --steps 10 --warmup 2 > "$GRAFT_REPO_ROOT/gpurun_out/prof_r02c_step/bench_trace.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_r02c_step/trace.err"; echo "step trace rc=$?"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r02c_pytest.log
FRIRL_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-learn --no-cpu-baseline --steps 30 --envs 4096 > gpurun_out/r02c_bench_gpus2_gloo.json 2> gpurun_out/r02c_bench_gpus2_gloo.err; echo "bench --gpus 2 rc=$?"
