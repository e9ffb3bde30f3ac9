#!/usr/bin/env python3
"""Throughput of the shared-rule-base evaluation kernels (one read-only rule base, Q observations):
   python tools/shared_bench.py [--rules 8192] [--obs 1048576] [--actions 3] [--nant 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import frirl_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rules", type=int, default=8192)
    ap.add_argument("--obs", type=int, default=1 << 20)
    ap.add_argument("--actions", type=int, default=3)
    ap.add_argument("--nant", type=int, default=5)
    ap.add_argument("--U", type=int, default=1001)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    nant, U, R = a.nant, a.U, a.rules
    u = torch.linspace(0, 1, U, dtype=torch.float64, device=dev).repeat(nant, 1).contiguous()
    ve = (u * 3.0).contiguous()
    idx = torch.randint(0, U, (nant, R), device=dev, generator=g)
    rb = torch.empty((1, nant + 1, R), dtype=torch.float64, device=dev)
    rb[0, :nant] = torch.gather(ve, 1, idx)
    rb[0, nant] = torch.rand((R,), dtype=torch.float64, device=dev, generator=g)
    nr = torch.full((1,), R, dtype=torch.int32, device=dev)
    prob = frirl_amd.Problem(u, ve, rb, nr)
    states = torch.rand((a.obs, nant - 1), dtype=torch.float64, device=dev, generator=g)
    ave = ve[nant - 1, torch.linspace(0, U - 1, a.actions).long()].contiguous()
    prob.get_best_action_shared(states, ave)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        prob.get_best_action_shared(states, ave)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(json.dumps({"kernel": "frirl_hip_get_best_action_shared", "rules": R, "obs": a.obs, "actions": a.actions, "ms": ms,
                      "obs_per_s": a.obs / ms * 1e3, "rule_action_evals_per_s": a.obs * R * a.actions / ms * 1e3}))


if __name__ == "__main__":
    main()
