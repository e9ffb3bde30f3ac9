#!/usr/bin/env python3
"""Greedy roll-outs of Q environments on one trained rule base (the bench's `evaluation` leg in isolation):
   python tools/rollout_bench.py [env] [Q] [opt=value ...]      e.g. acrobot 65536 rollout_slices=4 rollout_wps=2
Prints one JSON line: wall ms (events), env-steps/s, counted FP64-issue fraction."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import frirl_amd
from oracle import binding as ob

env = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
optv = dict(a.split("=") for a in sys.argv[3:])
for k, v in optv.items():
    frirl_amd.set_option(k, int(v))
dev = torch.device("cuda", 0)
fr = ob.Frirl(env, trig_mode=1)
assert fr.run() == 1
f = fr.five
R, nant = f.R, f.nant
maxR = 1024
rb = np.zeros((1, nant + 1, maxR)); rb[0, :nant, :R] = f.veval[:, :R]; rb[0, nant, :R] = f.rconc[:R]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
prob = frirl_amd.Problem(t(np.array(f.u)), t(np.array(f.ve)), t(rb), t(np.array([R], dtype=np.int32)))
dd = frirl_amd.demo_describe(env)
agent = frirl_amd.demo_agent(dd, dev)
ns = nant - 1
g = torch.Generator(device=dev); g.manual_seed(7)
lo = torch.tensor([dd["grids"][k].min() for k in range(ns)], dtype=torch.float64, device=dev)
hi = torch.tensor([dd["grids"][k].max() for k in range(ns)], dtype=torch.float64, device=dev)
vd = torch.tensor([dd["values_def"][k] for k in range(ns)], dtype=torch.float64, device=dev)
ss = (vd + (torch.rand((Q, ns), dtype=torch.float64, device=dev, generator=g) - 0.5) * 0.2 * (hi - lo)).clamp(lo, hi).contiguous()
prob.rollout_shared(agent, Q, start_states=ss)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "3"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    steps, rew, succ, _ = prob.rollout_shared(agent, Q, start_states=ss)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
A = agent.A
tot = int(steps.sum().item())
sweeps = tot + Q                                   # one greedy sweep per step + the first action's sweep
slots = sweeps * R * (14.4 * A + 2.0 * (nant - 1))
print(json.dumps({"env": env, "Q": Q, "rules": R, "A": A, "opts": optv, "ms": ms, "env_steps": tot, "env_steps_per_s": tot / ms * 1e3,
                  "steps_max": int(steps.max().item()), "succeeded": int((succ == 1).sum().item()),
                  "checksum": [int(steps.long().sum().item()), float(rew.sum().item())],
                  "fp64_issue_frac": slots / (ms * 1e-3) / 3.93e13}))
