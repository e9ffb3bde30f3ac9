#!/usr/bin/env python3
"""Wall time of the speculative try-remove reduction (frirl_hip_reduce_shared) on the three demo rule bases, next to the
oracle's sequential loop on one host core:  python tools/reduce_bench.py [--depth 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import frirl_amd
from oracle import binding as ob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for env in ("mountaincar", "cartpole", "acrobot"):
        for strategy in (1, 2):
            fr = ob.Frirl(env, trig_mode=1)
            fr.run()
            f = fr.five
            R, nant = f.R, f.nant
            maxR = R + 8 + (R & 1)
            rb = np.zeros((1, nant + 1, maxR))
            rb[0, :nant, :R] = f.veval[:, :R]
            rb[0, nant, :R] = f.rconc[:R]
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            agent = frirl_amd.demo_agent(frirl_amd.demo_describe(env), dev)
            best = None
            for rep in range(3):
                prob = frirl_amd.Problem(t(np.array(f.u)), t(np.array(f.ve)), t(rb), t(np.array([R], dtype=np.int32)))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                kept, res = prob.reduce_shared(agent, strategy, 0.0, a.depth)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            t0 = time.perf_counter()
            fr.reduce(strategy, 0.0)
            cpu = time.perf_counter() - t0
            print(json.dumps({"env": env, "strategy": strategy, "rules_before": R, "rules_after": res.rules_after, "oracle_rules_after": f.R,
                              "launches": res.rounds, "rollouts": res.rollouts, "gpu_s": round(best, 4), "oracle_1core_s": round(cpu, 4)}))


if __name__ == "__main__":
    main()
