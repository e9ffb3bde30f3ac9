#!/usr/bin/env python3
"""Interleaved A/B of distance-kernel variants in ONE process (frirl_hip_set_option between launches).
VARIANTS="nt,unroll,chunk[,persist[,order]];..."  (-1 / 0 = shipped value)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frirl_amd
import bench

w = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2_mountaincar_8k_x_8k"])
if len(sys.argv) > 2:
    w["E"] = int(sys.argv[2])
dev = torch.device("cuda", 0)
if w["env"]:
    prob, agent, envs = frirl_amd.demo_batch(w["env"], w["E"], w["R"], w["R"] + 256, dev, seed=0)
    del envs
else:
    prob, _, _ = bench.synth_problem(w, dev, 0)
x = bench.make_queries(prob, dev, 0)
dists = torch.empty((prob.E, prob.maxR), dtype=torch.float64, device=dev)
hit = torch.empty((prob.E,), dtype=torch.int32, device=dev)
variants = [tuple(map(int, v.split(","))) for v in os.environ.get("VARIANTS", "-1,0,0,-1,0;-1,0,0,-1,1;-1,0,0,0,0;-1,0,0,1,0;-1,0,4096,0,0;-1,0,1024,0,0").split(";")]
variants = [v + (-1, 0)[len(v) - 3:] if len(v) < 5 else v for v in variants]
compressed = prob.uidx is not None and not os.environ.get("AB_F64")
if not compressed:
    prob = frirl_amd.Problem(prob.u, prob.ve, prob.rb, prob.nrules)
alg = ((2.0 if compressed else 8.0) * prob.nant + 8.0) * prob.E * w["R"]      # bytes the kernel moves
res = {v: [] for v in variants}
for rep in range(int(os.environ.get("REPS", "6"))):
    for v in variants:
        for name, val in zip(("rd_nt", "rd_unroll", "rd_chunk", "rd_persist", "rd_order"), v):
            frirl_amd.set_option(name, val)
        for _ in range(3):
            prob.rule_distance(x, ruledists=dists, hit=hit)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            prob.rule_distance(x, ruledists=dists, hit=hit)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20)
for v in variants:
    t = sorted(res[v])
    med = t[len(t) // 2]
    print(f"nt={v[0]} unroll={v[1]} chunk={v[2]} persist={v[3]} order={v[4]}: median {med:.4f} ms  min {t[0]:.4f}  -> {alg / med / 1e6:.0f} GB/s (median)")
