#!/usr/bin/env python3
"""bench.py -- throughput of the FRIRL hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: five_hip_rule_distance over E independent
rule bases of R rules each (E*R rule-distance evaluations, distances materialised as the reference's
five_rule_distance does) -- BASELINE.json's metric "rule-distance evals/sec (rules x envs)".  The second
half of that metric, env-steps/sec, is measured by a second timed leg (fused do_action + reward +
quantise + greedy sweep + SARSA update per environment) and reported under "env_steps".  Two more legs report
the learning regime ("learning": E agents learn the demo from the reference's initial rule base to convergence,
per-episode reward statistics all-reduced) and evaluation mode ("evaluation": greedy roll-outs of 65 536
environments on one shared rule base); "cpu_baseline" times the genuine reference on the host cores (rank 0, N = 1).
Default workload = BASELINE.json configs[1]: mountaincar (nant 3, U 41, real tables and dynamics),
8192 rules x 8192 environments per GPU.  Environments are sharded over ranks by env id with no
data-path collective (weak scaling); only the per-episode reward statistics are all-reduced (RCCL).
value = all ranks' evaluations / max-over-ranks time.  Inputs are synthetic, generated on the device
and resident in HBM before the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {   # name -> demo env (tables / grids / dynamics), nant, U, R (rules per env), E (envs per GPU), A
    "cfg2_mountaincar_8k_x_8k": dict(env="mountaincar", nant=3, U=41, R=8192, E=8192, A=3),
    "cfg3_cartpole_32k_x_32k": dict(env="cartpole", nant=5, U=1001, R=32768, E=32768, A=21),
    "cfg4_acrobot_64k_x_8k_per_gpu": dict(env="acrobot", nant=5, U=41, R=65536, E=8192, A=3),
    "cfg5_synth16_256k": dict(env=None, nant=16, U=1001, R=262144, E=64, A=0),
}


def synth_problem(w, device, seed):
    """cfg5 (no environment): synthetic universes / VE tables and on-grid rule bases generated on the device."""
    import numpy as np
    import torch
    import frirl_amd
    nant, U, R, E = w["nant"], w["U"], w["R"], w["E"]
    g = torch.Generator(device=device).manual_seed(0x5EED0000 + seed)
    rng = np.random.default_rng(1234)
    u = np.zeros((nant, U))
    ve = np.zeros((nant, U))
    for k in range(nant):
        div = 2.0 * (k + 1) / (U - 1)
        half = [-(U - 1) * div / 2 + div * i for i in range(U // 2 + 1)]
        u[k] = half + [-half[U - 1 - i] for i in range(U // 2 + 1, U)]
        scf = 0.5 + rng.random(U)
        ve[k, 1:] = np.cumsum((u[k, 1:] - u[k, :-1]) * (scf[:-1] + scf[1:]) * 0.5)
    u_d, ve_d = torch.from_numpy(u).to(device), torch.from_numpy(ve).to(device)
    rb = torch.empty((E, nant + 1, R), dtype=torch.float64, device=device)
    uidx = torch.empty((E, nant, R), dtype=torch.int16, device=device)       # 16-bit universe-index mirror (FIVERB.rseqant_uindex)
    for k in range(nant):
        ik = torch.randint(0, U, (E, R), generator=g, device=device)
        rb[:, k, :] = ve_d[k][ik]
        uidx[:, k, :] = ik.to(torch.int16)
    rb[:, nant, :] = torch.rand((E, R), generator=g, device=device, dtype=torch.float64) * 3000.0 - 1500.0
    nrules = torch.full((E,), R, dtype=torch.int32, device=device)
    return frirl_amd.Problem(u_d, ve_d, rb, nrules, uidx), None, None


def make_queries(prob, device, seed):
    """Observations for the distance sweep: continuous, uniform over each universe (snapped by the kernel)."""
    import torch
    g = torch.Generator(device=device).manual_seed(77 + seed)
    lo, hi = prob.u[:, 0], prob.u[:, prob.U - 2]
    return (lo + (hi - lo) * torch.rand((prob.E, prob.nant), generator=g, device=device, dtype=torch.float64)).contiguous()


def cpu_baseline(w):
    """CPU baseline on this host, bounded to ~10-20 s: the genuine reference (oracle/_ref, AVX2 asm path,
    one core) when it shipped with the snapshot, else the oracle port."""
    nant, U, R = w["nant"], w["U"], w["R"]
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness) and nant <= 8:
        nq = max(64, int(2.0e10 / R))
        t0 = time.time()
        out = subprocess.run([harness, "bench", str(nant), str(U), str(R), str(nq)], check=True, capture_output=True, text=True).stdout
        rec = json.loads(out.strip().splitlines()[-1])
        # the reference's own parallel model is one private rule base per agent/core (frirl_agent.c:309-325): run one
        # harness process per host core at the same time (shorter sample) and add the rates up
        ncores = max(1, min(len(os.sched_getaffinity(0)), 16))    # the GPU box gives one GPU a 16-core share
        procs = [subprocess.Popen([harness, "bench", str(nant), str(U), str(R), str(max(64, nq // 4))], stdout=subprocess.PIPE, text=True)
                 for _ in range(ncores)]
        allc = 0.0
        for pr in procs:
            o, _ = pr.communicate()
            try:
                allc += json.loads(o.strip().splitlines()[-1])["rule_distance_evals_per_s"]
            except (ValueError, IndexError):
                pass
        # second half of the metric: the reference's own demo application, whole construct run on one core (its real regime:
        # a growing rule base of <= 367 rules), timed as a process (traced by the harness: FNV hash per step)
        demo = None
        if w["env"]:
            import tempfile
            best = None
            with tempfile.TemporaryDirectory() as td:
                for _ in range(3):
                    t1 = time.time()
                    subprocess.run([harness, "demo", w["env"], td], check=True, capture_output=True)
                    dt = time.time() - t1
                    best = dt if best is None else min(best, dt)
                end = json.loads(open(os.path.join(td, w["env"] + ".trace.jsonl")).read().strip().splitlines()[-1])
            demo = {"value": end["total_steps"] / best, "unit": "env-steps/s", "cores": 1,
                    "sample": f"genuine reference examples/{w['env']} (construct run, {end['total_steps']} steps, {end['episodes']} episodes, "
                              f"{end['R']} rules at the end) in {best * 1e3:.0f} ms"}
        return dict(value=rec["rule_distance_evals_per_s"], unit="rule-distance evals/s", cores=1, kind="reference", learning_env_steps=demo,
                    all_cores={"value": allc, "cores": ncores, "how": "one independent reference process per host core, run concurrently"},
                    sample=f"genuine reference five_rule_distance (AVX2 inline-asm path), one rule base nant={nant} R={R}, {nq} random queries "
                           f"({nq * R:.3g} evals, {rec['rule_distance_s']:.1f} s); vag_concl {rec['vag_concl_evals_per_s']:.3g} evals/s",
                    wall_s=round(time.time() - t0, 1))
    import numpy as np
    from tests.problems import Batch
    cores = os.cpu_count() or 1
    E = max(cores * 4, 32)
    b = Batch(nant, U, min(R, 65536), E, A=0, seed=5, ragged=False)
    x = b.queries(hit_fraction=0.01)
    b.oracle_rule_distance(x, nthreads=cores)
    reps, t0 = 0, time.time()
    while time.time() - t0 < 10.0:
        b.oracle_rule_distance(x, nthreads=cores)
        reps += 1
    dt = time.time() - t0
    return dict(value=reps * E * b.maxR / dt, unit="rule-distance evals/s", cores=cores, kind="port",
                sample=f"oracle port, OpenMP over {E} rule bases nant={nant} R={b.maxR}, {reps} sweeps in {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2_mountaincar_8k_x_8k", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--env-steps", type=int, default=20, help="timed environment steps of the fused SARSA leg")
    ap.add_argument("--no-env-steps", action="store_true")
    ap.add_argument("--no-learn", action="store_true", help="skip the real-learning leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import frirl_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FRIRL_DIST_BACKEND", "nccl")     # "gloo" only to rehearse N > 1 on a one-GPU box
    assert backend == "gloo" or local_rank < ndev, f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible"
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    if not (os.path.exists(frirl_amd.HIP_LIB_PATH) and os.path.exists(frirl_amd.DROPIN_LIB_PATH)):
        frirl_amd.build()          # normally prebuilt by __graft_entry__.build(); never rebuilt per rank
    D = frirl_amd.dist()
    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w["E"] = args.envs
    E, R, nant = w["E"], w["R"], w["nant"]
    env_start, _ = D.shard(E * world, world, rank)            # this rank's slice of the global env ids
    if w["env"]:
        maxR = R + 256                                        # head-room: the SARSA leg appends rules
        prob, agent, envs = frirl_amd.demo_batch(w["env"], E, R, maxR, device, seed=env_start)
    else:
        prob, agent, envs = synth_problem(w, device, seed=env_start)
    x = make_queries(prob, device, seed=env_start)
    dists = torch.empty((E, prob.maxR), dtype=torch.float64, device=device)
    hit = torch.empty((E,), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # on the launch stream
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            fn()
        ev1.record(stream)
        sync_all()
        dt = time.perf_counter() - t0
        return D.max_over_ranks(dt, device), ev0.elapsed_time(ev1) / steps

    # ---- leg 1 (the metric): rule-distance sweep, distances materialised ---------------------------------
    dt, kern_ms = timed(lambda: prob.rule_distance(x, ruledists=dists, hit=hit, stream=stream), args.steps, args.warmup)
    f64_ms = None
    if prob.uidx is not None and not os.environ.get("FRIRL_HIP_NO_UIDX"):
        # the same sweep on the reference's f64 SoA columns (no index mirror), timed beside the default path
        os.environ["FRIRL_HIP_NO_UIDX"] = "1"
        _, f64_ms = timed(lambda: prob.rule_distance(x, ruledists=dists, hit=hit, stream=stream), max(args.steps // 2, 5), 3)
        del os.environ["FRIRL_HIP_NO_UIDX"]
    nhits = torch.tensor([float((hit >= 0).sum().item())], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(nhits)

    # ---- leg 2: whole environment steps (do_action, reward, quantise, greedy sweep, SARSA update) ---------
    env_leg = None
    if agent is not None and not args.no_env_steps:
        frirl_amd.episode_begin(prob, agent, envs, stream=stream)
        edt, ems = timed(lambda: frirl_amd.episode_step(prob, agent, envs, stream=stream), args.env_steps, min(args.warmup, 3))
        # per-episode reward statistics: the ONLY cross-rank exchange (RCCL all-reduce over xGMI when N > 1)
        st = D.allreduce_stats(envs.ep_reward, envs.ep_steps, envs.done, prob.nrules)
        status = torch.bincount(envs.status.long(), minlength=6).tolist()
        env_leg = {"value": float(E) * world * args.env_steps / edt, "unit": "env-steps/s", "steps": args.env_steps,
                   "ms_per_step": edt / args.env_steps * 1e3, "avg_launch_ms": ems,
                   "algorithmic_bytes_per_step": float(E) * (2.0 * R * (nant + 1) * 8 + R * 8),
                   "stats_allreduce": {"envs": st.envs, "mean_reward": st.mean_reward, "mean_rules": st.mean_rules, "steps_sum": st.steps_sum,
                                       "episodes_done": st.success, "reward_min": st.reward_min, "reward_max": st.reward_max},
                   "last_step_outcomes_rank0": dict(zip(["inactive", "exact", "spread", "inserted", "skipped", "full"], status))}
        env_leg["effective_GBps"] = env_leg["algorithmic_bytes_per_step"] / (ems * 1e-3) / 1e9

    # ---- leg 3: real learning -- E agents learn the demo from the reference's initial 2^nant rule base until each
    # rule base is "considered complete" (batched frirl_sequential_run construct loop, all on the device) --------
    learn_leg = None
    if agent is not None and not args.no_learn:
        del dists
        lE = min(E, 8192)
        lprob, lagent, lenvs = frirl_amd.demo_fresh_batch(w["env"], lE, 1024, device)
        lsteps = torch.zeros((), dtype=torch.int64, device=device)

        ep_log = []

        def on_ep(ep, conv):
            lsteps.add_((lenvs.ep_steps.long() * (conv.episodes == ep).long()).sum())
            # the reference's per-episode report (frirl_sequential_run.c:74-77) for the whole job: reward statistics only
            # cross the GPUs -- one tiny all-reduce per episode (RCCL over xGMI; latency-bound, overlaps the next episode)
            st = torch.stack([lenvs.ep_reward.sum(), lenvs.ep_steps.sum().double(), lprob.nrules.sum().double(), conv.converged.sum().double()])
            if world > 1:
                dist.all_reduce(st)
            ep_log.append(st)
        # untimed warm-up: one short episode of a 64-agent batch loads the code objects of the episode kernels ...
        wprob, wagent, wenvs = frirl_amd.demo_fresh_batch(w["env"], 64, 1024, device)
        frirl_amd.train(wprob, wagent, wenvs, max_episodes=3)
        del wprob, wagent, wenvs
        # ... and one pass over the bookkeeping ops (first use of a torch kernel loads its code object: milliseconds each)
        _w = torch.stack([lenvs.ep_reward.sum(), lenvs.ep_steps.sum().double(), lprob.nrules.sum().double(), torch.zeros((), device=device, dtype=torch.float64)])
        _w2 = (lenvs.ep_steps.long() * (lprob.nrules == 1).long()).sum()
        if world > 1:
            dist.all_reduce(_w)
        del _w, _w2
        if frirl_amd.lib().frirl_hip_lanes_preferred(lprob.nant, lE, lagent.A):
            frirl_amd.episode_run_lanes(lprob, lagent, lenvs, 0)      # allocates the transposed-rule-base workspace outside the timed region
        sync_all()
        t0 = time.perf_counter()
        conv = frirl_amd.train(lprob, lagent, lenvs, on_episode=on_ep)
        sync_all()
        ldt = D.max_over_ranks(time.perf_counter() - t0, device)
        tot = torch.tensor([float(lsteps.item()), float(conv.converged.sum().item()), float(lprob.nrules.sum().item())], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tot)
        learn_leg = {"value": tot[0].item() / ldt, "unit": "env-steps/s", "agents": lE * world, "wall_s": ldt, "env_steps": tot[0].item(),
                     "agents_converged": tot[1].item(), "episodes_to_converge": int(conv.episodes.max().item()),
                     "mean_final_rules": tot[2].item() / (lE * world),
                     "per_episode_stats_allreduce": {"episodes": len(ep_log), "last": dict(zip(["reward_sum", "steps_sum", "rules_sum", "converged"],
                                                                                          [float(v) for v in ep_log[-1].tolist()])) if ep_log else None},
                     "kernel": "episode_run_lanes (lane groups)" if frirl_amd.lib().frirl_hip_lanes_preferred(lprob.nant, lE, lagent.A)
                               else "episode_run / episode_step (one wave per environment)",
                     "note": "whole construct run from the 2^nant corner rules (reference: 15548 / 33002 / 21207 steps per agent for "
                             "mountaincar / cartpole / acrobot); rule bases stay small (<= 367 rules): latency / occupancy bound at "
                             "8192 agents, 1.2-1.5e9 env-steps/s at 65536 mountaincar agents (tools/learn_bench.py)"}

    # ---- leg 4: evaluation mode -- greedy roll-outs of many environments on ONE shared, read-only rule base (the one
    # agent 0 just learned): frirl_test_run's episode, lane group per environment -------------------------------
    eval_leg = None
    if learn_leg is not None:
        ns = lprob.nant - 1
        dd = frirl_amd.demo_describe(w["env"])
        one = frirl_amd.Problem(lprob.u, lprob.ve, lprob.rb[0:1].clone(), lprob.nrules[0:1].clone())
        Qn = 65536
        g = torch.Generator(device=device)
        g.manual_seed(7 + rank)
        lo = torch.tensor([dd["grids"][k].min() for k in range(ns)], dtype=torch.float64, device=device)
        hi = torch.tensor([dd["grids"][k].max() for k in range(ns)], dtype=torch.float64, device=device)
        vd = torch.tensor([dd["values_def"][k] for k in range(ns)], dtype=torch.float64, device=device)
        ss = (vd + (torch.rand((Qn, ns), dtype=torch.float64, device=device, generator=g) - 0.5) * 0.2 * (hi - lo)).clamp(lo, hi).contiguous()
        one.rollout_shared(lagent, Qn, start_states=ss)
        sync_all()
        t0 = time.perf_counter()
        rsteps, rrew, rsucc, _ = one.rollout_shared(lagent, Qn, start_states=ss)
        sync_all()
        edt = D.max_over_ranks(time.perf_counter() - t0, device)
        et = torch.tensor([float(rsteps.sum().item()), float((rsucc == 1).sum().item())], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(et)
        eval_leg = {"value": et[0].item() / edt, "unit": "env-steps/s", "environments": Qn * world, "rules": int(one.nrules[0].item()),
                    "wall_s": edt, "env_steps": et[0].item(), "episodes_succeeded": et[1].item(),
                    "note": "frirl_hip_rollout_shared: whole greedy episodes from perturbed start states on one shared rule base, no updates"}

    if rank == 0:
        evals = float(E) * R * args.steps * world
        alg_bytes = 8.0 * (nant + 1) * E * R                 # SURVEY 8d U1 contract figure (f64 SoA layout): 8*nant read + 8 written per eval
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        tabb = 8 * nant * w["U"]
        compressed = prob.uidx is not None and tabb <= 150 * 1024 and not os.environ.get("FRIRL_HIP_NO_UIDX")
        moved_bytes = (2.0 * nant + 8.0) * E * R if compressed else alg_bytes     # what the kernel actually streams
        moved = moved_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "rule-distance evals/sec (rules x envs)", "value": evals / dt, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "device": torch.cuda.get_device_name(device),
            "config": {"workload": args.workload, "nant": nant, "universe_len": w["U"], "rules_per_env": R, "envs_per_gpu": E,
                       "sharding": f"env ids split over {world} rank(s), no data-path collective; reward statistics all-reduced",
                       "exact_hits": int(nhits.item())},
            "roofline": {"bound": "hbm", "kernel": "rule_distance_idx_kernel" if compressed else "rule_distance_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms,
                         "layout": "u16 universe-index mirror + LDS VE tables (bit-identical to the f64 columns)" if compressed else "f64 SoA columns",
                         "moved_bytes_per_launch": moved_bytes, "moved_GBps": moved, "moved_frac": moved / HBM_PEAK_GBS,
                         "note": ("achieved/frac use the CONTRACT bytes of the reference's f64 layout (SURVEY 8d); achieved > peak is a compression "
                                  "effect, not bandwidth: the kernel streams moved_bytes_per_launch (2*nant B read + 8 B written per eval), "
                                  "moved_GBps / moved_frac is its real HBM rate") if compressed else "f64 layout: contract bytes == moved bytes"},
        }
        if f64_ms:
            out["roofline"]["f64_layout"] = {"kernel": "rule_distance_kernel", "avg_launch_ms": f64_ms, "evals_per_s": float(E) * R / (f64_ms * 1e-3),
                                             "achieved": alg_bytes / (f64_ms * 1e-3) / 1e9, "frac": alg_bytes / (f64_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "note": "same sweep streaming the reference's f64 columns (contract bytes == moved bytes)"}
        if env_leg:
            out["env_steps"] = env_leg
        if learn_leg:
            out["learning"] = learn_leg
        if eval_leg:
            out["evaluation"] = eval_leg
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
            if tr and not args.envs:
                out["roofline"]["traffic"] = tr["fetch_bytes"] + tr["write_bytes"]
                out["roofline"]["traffic_source"] = tr["source"]
        except (OSError, ValueError):
            pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
