set -e
timeout -k 10 300 python bench.py > gpurun_out/final_cfg2.json 2> gpurun_out/final_cfg2.err
timeout -k 10 500 python bench.py --no-cpu-baseline --no-learn --workload cfg3_cartpole_32k_x_32k --envs 4096 --steps 10 --warmup 2 --env-steps 10 > gpurun_out/final_cfg3.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-learn --workload cfg4_acrobot_64k_x_8k_per_gpu > gpurun_out/final_cfg4.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --workload cfg5_synth16_256k > gpurun_out/final_cfg5.json 2>/dev/null
for c in cfg2 cfg3 cfg4 cfg5; do python -c "
import json; d=json.load(open('gpurun_out/final_$c.json')); r=d['roofline']; e=d.get('env_steps',{})
print('$c', 'evals/s %.3e' % d['value'], 'frac %.2f moved %.2f' % (r['frac'], r.get('moved_frac', r['frac'])), 'f64 %.3e frac %.2f' % (r.get('f64_layout',{}).get('evals_per_s',0), r.get('f64_layout',{}).get('frac',0)), 'env-steps/s %.3e (%.3f ms)' % (e.get('value',0), e.get('ms_per_step',0)))"; done
bash tools/profile_bench.sh r01_final > gpurun_out/prof_final.md 2> gpurun_out/prof_final.err
echo profiled
