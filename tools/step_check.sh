set -e
timeout -k 10 400 python -m pytest tests/test_hip_sarsa.py tests/test_hip_train.py tests/test_full_size.py -m gpu -x -q > gpurun_out/step_suite.log 2>&1 || { tail -30 gpurun_out/step_suite.log; exit 1; }
tail -1 gpurun_out/step_suite.log
for w in cfg2_mountaincar_8k_x_8k cfg4_acrobot_64k_x_8k_per_gpu; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-learn --workload $w > gpurun_out/bench_$w.json 2>/dev/null
  python -c "
import json,sys; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', 'evals/s %.3e' % d['value'], 'env-steps/s %.3e' % d['env_steps']['value'], 'ms/step %.4f' % d['env_steps']['ms_per_step'])"
done
timeout -k 10 500 python bench.py --no-cpu-baseline --no-learn --workload cfg3_cartpole_32k_x_32k --envs 4096 --steps 10 --warmup 2 --env-steps 10 > gpurun_out/bench_cfg3.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/bench_cfg3.json')); print('cfg3/4096 envs evals/s %.3e env-steps/s %.3e ms/step %.3f' % (d['value'], d['env_steps']['value'], d['env_steps']['ms_per_step']))"
