#!/usr/bin/env python3
"""Persistent construct loop (frirl_hip_learn_run): E agents learn the demo from the corner rule base until every one has
converged (or has run max_episodes - 1 episodes).  env-steps/s over the whole run + the counted FP64-issue fraction.
   python tools/learn2_bench.py [env] [E] [rep|div] [budget] [max_episodes] [opt=value ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frirl_amd

env = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
diversify = len(sys.argv) > 3 and sys.argv[3] == "div"
budget = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
max_episodes = int(sys.argv[5]) if len(sys.argv) > 5 else 400
optv = dict(a.split("=") for a in sys.argv[6:])
for k, v in optv.items():
    frirl_amd.set_option(k, int(v))
dev = torch.device("cuda", 0)
d = frirl_amd.demo_describe(env)
start = None
if diversify:          # per-agent start state on the state grid (reference gen_def_states, frirl_agent.c:121-139)
    g = torch.Generator(device=dev).manual_seed(1)
    cols = []
    for k in range(d["nstates"]):
        vals = torch.from_numpy(d["grids"][k]).to(dev)
        cols.append(vals[torch.randint(0, len(vals), (E,), generator=g, device=dev)])
    start = torch.stack(cols, 1).contiguous()
# warm-up (code objects, allocator)
for Hw in ((2, 4, 8, 16, 32, 64) if os.environ.get("WARM_ALL") else (0,)):     # first use of a kernel variant costs tens of ms
    wp, wa, we = frirl_amd.demo_fresh_batch(env, 64, 1024, dev)
    frirl_amd.set_option("learn_slices", Hw)
    frirl_amd.train_persistent(wp, wa, we, max_episodes=3, budget=64)
    del wp, wa, we
frirl_amd.set_option("learn_slices", int(optv.get("learn_slices", 0)))
prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 1024, dev, start_states=start)
chunks = []
progress = []
stamps = []
def on_chunk(i, live, conv):
    chunks.append(E if live is None else int(live.numel()))
    if os.environ.get("STAMP"):
        torch.cuda.synchronize(); stamps.append(time.perf_counter())
        st, wk = prob._learn_progress
        progress.append((int(st.sum()), int(wk[:, 0].sum()), int(wk[:, 1].sum()), float(prob.nrules[live.long()].float().mean()) if live is not None else 0.0,
                         int(prob.nrules[live.long()].max()) if live is not None else 0))
torch.cuda.synchronize(); t0 = time.perf_counter()
run = frirl_amd.train_persistent(prob, agent, envs, max_episodes=max_episodes, budget=budget, on_chunk=on_chunk)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
total = int(run.steps_total.sum())
w = run.work.sum(0).tolist()
nant, A = prob.nant, agent.A
slots = w[0] * (14.4 * (A + 1) + 4.0 * (nant - 1)) + w[1] * (2.0 * nant + 10.4)
print(json.dumps({"kernel": "learn_run", "env": env, "agents": E, "diversified_start": diversify, "budget": budget, "max_episodes": max_episodes, "opts": optv,
                  "wall_s": dt, "env_steps": total, "env_steps_per_s": total / dt, "launches": run.launches, "live_per_launch": chunks[:40],
                  "converged": int(run.conv.converged.sum()), "episodes_max": int(run.conv.episodes.max()), "rules_min": int(prob.nrules.min()), "rules_max": int(prob.nrules.max()),
                  "launch_ms": [round((b - a) * 1e3, 2) for a, b in zip([t0] + stamps[:-1], stamps)][:40],
                  "per_launch": [dict(live=c, ms=round((b - a) * 1e3, 2), steps=p1[0] - p0[0], visits=p1[1] - p0[1], extra=p1[2] - p0[2], mean_rules=round(p1[3], 1), max_rules=p1[4],
                                      slots_frac=round(((p1[1] - p0[1]) * (14.4 * (A + 1) + 4.0 * (nant - 1)) + (p1[2] - p0[2]) * (2.0 * nant + 10.4)) / (b - a) / 3.93e13, 3))
                                 for c, a, b, p0, p1 in zip(chunks, [t0] + stamps[:-1], stamps, [(0, 0, 0, 0, 0)] + progress[:-1], progress)],
                  "visits_main": w[0], "visits_extra": w[1], "fp64_issue_frac": slots / dt / 3.93e13}))
