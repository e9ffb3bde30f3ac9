#!/usr/bin/env python3
"""Evaluation-mode numbers on the three demo rule bases (learned here on the GPU, E = 1):
  * wall time of the speculative try-remove reduction (frirl_hip_reduce_shared), strategies 1 and 2
  * throughput of greedy roll-outs on the shared rule base (frirl_hip_rollout_shared), Q environments
python tools/reduce_bench.py [--depth 10] [--envs 65536]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import frirl_amd


def learned(env, dev):
    prob, agent, envs = frirl_amd.demo_fresh_batch(env, 1, 1024, dev)
    conv = frirl_amd.train(prob, agent, envs)
    torch.cuda.synchronize()
    assert int(conv.converged[0]) == 1
    return prob, agent, envs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--envs", type=int, default=65536)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for env in ("mountaincar", "cartpole", "acrobot"):
        prob, agent, envs = learned(env, dev)
        R = int(prob.nrules[0])
        rb0, nr0 = prob.rb.clone(), prob.nrules.clone()
        ui0 = prob.uidx.clone() if prob.uidx is not None else None
        # roll-outs from perturbed start states
        ns = prob.nant - 1
        g = torch.Generator(device=dev); g.manual_seed(3)
        d = frirl_amd.demo_describe(env)
        lo = torch.tensor([d["grids"][k].min() for k in range(ns)], dtype=torch.float64, device=dev)
        hi = torch.tensor([d["grids"][k].max() for k in range(ns)], dtype=torch.float64, device=dev)
        vd = torch.tensor([d["values_def"][k] for k in range(ns)], dtype=torch.float64, device=dev)
        ss = (vd + (torch.rand((a.envs, ns), dtype=torch.float64, device=dev, generator=g) - 0.5) * 0.2 * (hi - lo)).clamp(lo, hi).contiguous()
        prob.rollout_shared(agent, a.envs, start_states=ss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps, reward, success, _ = prob.rollout_shared(agent, a.envs, start_states=ss)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot = int(steps.sum())
        print(json.dumps({"env": env, "rules": R, "rollouts": a.envs, "env_steps": tot, "s": round(dt, 4), "env_steps_per_s": tot / dt,
                          "success_frac": float((success == 1).double().mean())}))
        for strategy in (1, 2):
            best = None
            for rep in range(3):
                prob.rb.copy_(rb0); prob.nrules.copy_(nr0)
                if ui0 is not None:
                    prob.uidx.copy_(ui0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                kept, res = prob.reduce_shared(agent, strategy, 0.0, a.depth)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(json.dumps({"env": env, "strategy": strategy, "rules_before": R, "rules_after": res.rules_after, "launches": res.rounds,
                              "rollouts": res.rollouts, "steps_incremental": res.steps_incremental, "gpu_s": round(best, 4)}))


if __name__ == "__main__":
    main()
