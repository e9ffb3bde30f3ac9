#!/usr/bin/env python3
"""Fused step at the cfg4 / cfg2 shapes with different rule-base capacities (maxR = R + pad): the column stride decides how the
~7700 concurrent streams of the one-workgroup-per-environment step fall on the HBM channels (profiles/r02_rule_distance_order.md:
reads alone 5.9-6.8 TB/s environment-fastest depending on the pad).   python tools/step_pad_ab.py [cfg4|cfg2] pad ..."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import frirl_amd

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
pads = [int(x) for x in sys.argv[2:]] or [256, 1024, 2048]
env, R, E = ("acrobot", 65536, 8192) if which == "cfg4" else ("mountaincar", 8192, 8192)
dev = torch.device("cuda", 0)
for pad in pads:
    prob, agent, envs = frirl_amd.demo_batch(env, E, R, R + pad, dev, seed=0, keep_rant=False)
    frirl_amd.episode_begin(prob, agent, envs)
    for _ in range(3):
        frirl_amd.episode_step(prob, agent, envs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        frirl_amd.episode_step(prob, agent, envs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    moved = float(E) * R * (2.0 * prob.nant + 8.0)
    print(json.dumps({"shape": which, "pad": pad, "maxR": R + pad, "ms_per_step": ms, "moved_frac": moved / (ms * 1e-3) / 8e12}), flush=True)
    del prob, agent, envs
    torch.cuda.empty_cache()
