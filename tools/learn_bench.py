#!/usr/bin/env python3
"""Real learning throughput: E agents of a demo learn from the initial corner rule base until convergence
(batched construct run on the device).  Reports env-steps/s over the whole run."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frirl_amd

env = sys.argv[1] if len(sys.argv) > 1 else "mountaincar"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
diversify = len(sys.argv) > 3 and sys.argv[3] == "div"
lanes = {"1": True, "0": False}.get(os.environ.get("LANES", ""), None)      # force / forbid the lane-group kernel; default: library heuristic
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
prob, agent, envs = frirl_amd.demo_fresh_batch(env, E, 1024, dev, start_states=start)
total = torch.zeros((), dtype=torch.int64, device=dev)
def on_ep(ep, conv):
    total.add_((envs.ep_steps.long() * (conv.episodes == ep).long()).sum())
torch.cuda.synchronize(); t0 = time.perf_counter()
conv = frirl_amd.train(prob, agent, envs, on_episode=on_ep, max_episodes=400, lanes=lanes)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"lanes": lanes, "env": env, "agents": E, "diversified_start": diversify, "wall_s": dt, "env_steps": int(total), "env_steps_per_s": int(total) / dt,
                  "converged": int(conv.converged.sum()), "episodes_max": int(conv.episodes.max()), "rules_min": int(prob.nrules.min()),
                  "rules_max": int(prob.nrules.max())}))
