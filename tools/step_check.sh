set -e
timeout -k 10 400 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py -m gpu -x -q > gpurun_out/step_suite.log 2>&1 || { tail -30 gpurun_out/step_suite.log; exit 1; }
tail -2 gpurun_out/step_suite.log
for sw in 0 1; do
export FRIRL_HIP_STEP_WAVE=$sw
for w in cfg2_mountaincar_8k_x_8k cfg4_acrobot_64k_x_8k_per_gpu; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-learn --workload $w > gpurun_out/bench_$w.json 2>/dev/null
  python -c "
import json,sys; d=json.load(open('gpurun_out/bench_$w.json')); print('wave=$sw', '$w', 'evals/s %.3e' % d['value'], 'env-steps/s %.3e' % d['env_steps']['value'], 'ms/step %.4f' % d['env_steps']['ms_per_step'])"
done
done
