#!/usr/bin/env python3
"""Where a step of the persistent learner spends its cycles: needs a library built with -DLEARN_TIMING
(tools/exp/build_learn_variant.sh timing -DLEARN_TIMING; FRIRL_HIP_LIB_OVERRIDE=.../libfrirl_hip_timing.so).
   python tools/learn_sections.py [env] [E] [rep|div] [budget] [max_episodes] [launches] [opt=value ...]
Prints, per launch, the share of wave cycles per section of the step."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frirl_amd

env = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
diversify = len(sys.argv) > 3 and sys.argv[3] == "div"
budget = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
max_episodes = int(sys.argv[5]) if len(sys.argv) > 5 else 200
for a in sys.argv[6:]:
    k, v = a.split("="); frirl_amd.set_option(k, int(v))
dev = torch.device("cuda", 0)
d = frirl_amd.demo_describe(env)
start = None
if diversify:
    g = torch.Generator(device=dev).manual_seed(1)
    start = torch.stack([torch.from_numpy(d["grids"][k]).to(dev)[torch.randint(0, len(d["grids"][k]), (E,), generator=g, device=dev)] for k in range(d["nstates"])], 1).contiguous()
prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 1024, dev, start_states=start)
L = frirl_amd.lib()
names = ["env+observe", "park", "sweep", "butterfly+unpark", "greedy+Q", "boundary/snap/append", "write/spread", "episode end", "rest (begin, loop)"]
buf = (C.c_ulonglong * 16)()
L.frirl_hip_learn_timing(buf)
last = [0]
tlast = [None]
def on_chunk(i, live, conv):
    torch.cuda.synchronize()
    now = time.perf_counter()
    wall = now - tlast[0] if tlast[0] is not None else float("nan")
    tlast[0] = now
    L.frirl_hip_learn_timing(buf)
    t = [buf[k] for k in range(9)]
    tot = sum(t) or 1
    st = int(prob._learn_progress[0].sum())
    n = int(live.numel())
    H = 2
    while H < 64 and n * 2 * H <= 131072: H *= 2
    wave_steps = (st - last[0]) / (64 // H)
    last[0] = st
    waves = (n * H + 63) // 64
    print(json.dumps({"launch": i, "live": n, "wall_ms": round(wall * 1e3, 2), "waves": waves,
                      "mean_wave_busy_ms_at_1p9GHz": round(tot / max(waves, 1) / 1.9e9 * 1e3, 2), "mean_rules": round(float(prob.nrules[live.long()].float().mean()), 1), "cycles_per_wave_step": round(tot / max(wave_steps, 1)),
                      "share": {nm: round(x / tot, 3) for nm, x in zip(names, t)}}))
torch.cuda.synchronize(); tlast[0] = time.perf_counter()
frirl_amd.train_persistent(prob, agent, envs, max_episodes=max_episodes, budget=budget, on_chunk=on_chunk)
