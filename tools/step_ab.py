#!/usr/bin/env python3
"""A/B of the fused episode step under option values:  tools/step_ab.py <workload> <E> <option> v1,v2,...  (interleaved, median)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frirl_amd
import bench

w = dict(bench.WORKLOADS[sys.argv[1]])
if int(sys.argv[2]) > 0:
    w["E"] = int(sys.argv[2])
opt, vals = sys.argv[3], [int(v) for v in sys.argv[4].split(",")]
dev = torch.device("cuda", 0)
prob, agent, envs = frirl_amd.demo_batch(w["env"], w["E"], w["R"], w["R"] + 256, dev, seed=0, keep_rant=False)
res = {v: [] for v in vals}
for rep in range(5):
    for v in vals:
        frirl_amd.set_option(opt, v)
        frirl_amd.episode_begin(prob, agent, envs)
        for _ in range(2):
            frirl_amd.episode_step(prob, agent, envs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            frirl_amd.episode_step(prob, agent, envs)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 8)
st = torch.bincount(envs.status.long(), minlength=6).tolist()
for v in vals:
    t = sorted(res[v])
    print(f"{sys.argv[1]} E={w['E']} {opt}={v}: median {t[len(t) // 2]:.4f} ms/step  min {t[0]:.4f}   (last outcomes {st})")
