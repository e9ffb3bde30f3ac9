#!/usr/bin/env python3
"""bench.py -- throughput of the FRIRL hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before anything touches the
GPU it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`
as a child, relays rank 0's JSON line and exits with the child's code.  Under a launcher (RANK / LOCAL_RANK /
WORLD_SIZE set) it is one rank of the job.

One "step" = one pass of the hot path over one batch: five_hip_rule_distance over E independent rule bases of R rules
each (E*R rule-distance evaluations, distances materialised as the reference's five_rule_distance does) --
BASELINE.json's metric "rule-distance evals/sec (rules x envs)".  Default workload = the north-star shape on one GPU
(BASELINE.json configs[3] per GPU): acrobot tables (nant 5, U 41), 65 536 rules x 8 192 environments.  The second half
of the metric, env-steps/sec, is the "env_steps" leg (fused do_action + reward + quantise + greedy sweep + SARSA update
per environment); "learning" / "learning_diversified" = 65 536 agents per GPU learn the demo from the reference's initial
rule base until every rule base is complete (replicas of the demo / one start state per agent; "learning_diversified_4x" =
the same with 262 144 agents per GPU, four times what the chip keeps resident: launches stay full until the end of the run;
the job's report crosses the GPUs once per leg), with a COUNTED FP64-issue roofline (rule visits accumulated by the kernel); "evaluation" = greedy roll-outs of
65 536 environments on one shared rule base, counted the same way ("evaluation_16x": a million environments per GPU, the queue-fed
throughput regime of the same call); "other_configs" = the first two legs on BASELINE's other
configurations (cfg2, cfg3, cfg5; N = 1 only), and for their demos (mountaincar, cartpole) the many-agent learning run as "learning"; "cpu_baseline" = the genuine reference on the host cores (rank 0, N = 1,
time-boxed).  Every timed leg is followed, outside the timed region, by a PARITY GATE (tests/gates.py): sampled environments
against the oracle -- the run exits non-zero on a mismatch.

roofline.achieved / frac are PHYSICAL: the bytes the timed kernel moves (compressed layout: 2*nant B of 16-bit
universe indices read + 8 B distance written per evaluation) / the average launch time from
events on the launch stream / 8 TB/s.  The reference-layout figure (8*(nant+1) B per evaluation) is reported separately
as contract_equiv_GBps / contract_frac and may exceed 1 -- it is a compression effect, not a bandwidth.

Environments are sharded over ranks by env id with no data-path collective (weak scaling); only the per-episode reward
statistics are all-reduced (RCCL).  value = all ranks' evaluations / max-over-ranks time.  Inputs are synthetic,
generated on the device and resident in HBM before the timed region.
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
DEFAULT_WORKLOAD = "cfg4_acrobot_64k_x_8k_per_gpu"


# ---- launcher (parent process; nothing here may touch torch.cuda / HIP) -------------------------------------------------
def launch_ranks(gpus, argv):
    """Starts `gpus` ranks of this script under torch.distributed.run as a CHILD process and relays its output.
    Never exec: a process that has initialised the GPU must not be replaced, and the parent has not initialised it."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def rank_probe():
    """--rank-probe: the launcher / rendezvous path without a GPU (CPU test): every rank joins a gloo group, the ranks
    are counted by an all-reduce and rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    n = torch.ones(1, dtype=torch.float64)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.all_reduce(n)
        backend = dist.get_backend()
        world = dist.get_world_size()
    if rank == 0:
        print(json.dumps({"metric": "rank-probe", "value": float(n.item()), "n_gpus": world, "config": {"dist_backend": backend, "world_size": world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---- synthetic inputs ---------------------------------------------------------------------------------------------------
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


# ---- CPU baseline (rank 0, N = 1; time-boxed) ---------------------------------------------------------------------------
def _harness_bench(harness, nant, U, R, nq):
    out = subprocess.run([harness, "bench", str(nant), str(U), str(R), str(nq)], check=True, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def cpu_baseline(w, budget_s=18.0):
    """CPU baseline on this host, bounded to ~budget_s seconds in total: the genuine reference (oracle/_ref, AVX2 asm
    path) when it shipped with the snapshot, else the oracle port.  The number of queries comes from a calibration call
    (the reference's rate drops ~15x between a cache-resident 8k-rule base and a 64k-rule one)."""
    nant, U, R = w["nant"], w["U"], w["R"]
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    t0 = time.time()
    if os.path.exists(harness) and nant <= 8:
        ncal = max(32, int(2.0e7 / R))                                             # calibration: ~0.1-0.3 s, past the first-touch effects
        cal = _harness_bench(harness, nant, U, R, ncal)
        rate_q = ncal / max(cal["rule_distance_s"] + cal["vag_concl_s"], 1e-6)     # queries per second incl. the Shepard leg (1/8 of the queries)
        nq = max(64, int(0.45 * budget_s * rate_q))                                # one-core leg: ~45 % of the budget
        rec = _harness_bench(harness, nant, U, R, nq)
        # the reference's own parallel model is one private rule base per agent/core (frirl_agent.c:309-325): one harness
        # process per host core at the same time (~25 % of the budget) and the rates added up
        ncores = max(1, min(len(os.sched_getaffinity(0)), 16))    # the GPU box gives one GPU a 16-core share
        nq_all = max(64, int(0.30 * budget_s * rate_q))
        procs = [subprocess.Popen([harness, "bench", str(nant), str(U), str(R), str(nq_all)], stdout=subprocess.PIPE, text=True) for _ in range(ncores)]
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
                for _ in range(2):
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
                    sample=f"genuine reference five_rule_distance (AVX2 inline-asm path) on ONE rule base of nant={nant} R={R} that stays resident in the "
                           f"core's caches ({(nant + 1) * 8 * R / 1e6:.1f} MB of columns + its scratch) -- the reference's real regime; the GPU figure streams E "
                           f"distinct rule bases from HBM.  {nq} random queries ({nq * R:.3g} evals, {rec['rule_distance_s']:.1f} s); "
                           f"vag_concl {rec['vag_concl_evals_per_s']:.3g} evals/s",
                    wall_s=round(time.time() - t0, 1))
    import numpy as np  # noqa: F401
    from tests.problems import Batch
    cores = os.cpu_count() or 1
    E = max(cores * 4, 32)
    b = Batch(min(nant, 16), U, min(R, 65536), E, A=0, seed=5, ragged=False)
    x = b.queries(hit_fraction=0.01)
    b.oracle_rule_distance(x, nthreads=cores)
    reps, t1 = 0, time.time()
    while time.time() - t1 < 0.5 * budget_s:
        b.oracle_rule_distance(x, nthreads=cores)
        reps += 1
    dt = time.time() - t1
    return dict(value=reps * E * b.maxR / dt, unit="rule-distance evals/s", cores=cores, kind="port",
                sample=f"oracle port, OpenMP over {E} rule bases nant={nant} R={b.maxR}, {reps} sweeps in {dt:.1f} s", wall_s=round(time.time() - t0, 1))


# ---- timed legs ---------------------------------------------------------------------------------------------------------
class Bench:
    def __init__(self, device, world, rank, D):
        import torch
        self.torch, self.device, self.world, self.rank, self.D = torch, device, world, rank, D
        self.stream = torch.cuda.current_stream()

    def sync_all(self):
        import torch.distributed as dist
        self.torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        """W untimed + EXACTLY `steps` timed calls bracketed by barrier + synchronize; returns (max-over-ranks wall s,
        average ms per call from HIP events recorded on the launch stream)."""
        torch = self.torch
        for _ in range(warmup):
            fn()
        self.sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(self.stream)
        for _ in range(steps):
            fn()
        ev1.record(self.stream)
        self.sync_all()
        dt = time.perf_counter() - t0
        return self.D.max_over_ranks(dt, self.device), ev0.elapsed_time(ev1) / steps

    def build(self, w, env_start, keep_rant=True):
        import frirl_amd
        if w["env"]:
            maxR = w["R"] + 256                                        # head-room: the SARSA leg appends rules
            return frirl_amd.demo_batch(w["env"], w["E"], w["R"], maxR, self.device, seed=env_start, keep_rant=keep_rant)
        return synth_problem(w, self.device, seed=env_start)

    def rule_distance_leg(self, name, w, prob, x, steps, warmup):
        """The metric's first half.  Returns (value record, roofline record)."""
        import frirl_amd
        torch = self.torch
        E, R, nant = w["E"], w["R"], w["nant"]
        dists = torch.empty((E, prob.maxR), dtype=torch.float64, device=self.device)
        hit = torch.empty((E,), dtype=torch.int32, device=self.device)
        dt, kern_ms = self.timed(lambda: prob.rule_distance(x, ruledists=dists, hit=hit, stream=self.stream), steps, warmup)
        compressed = prob.uidx is not None and bool(frirl_amd.lib().five_hip_rule_distance_uses_uidx(prob.nant, prob.U))
        f64_ms = None
        if compressed:
            # the same sweep on the reference's f64 SoA columns (no index mirror), timed beside the shipped path
            plain = frirl_amd.Problem(prob.u, prob.ve, prob.rb, prob.nrules)
            _, f64_ms = self.timed(lambda: plain.rule_distance(x, ruledists=dists, hit=hit, stream=self.stream), max(steps // 3, 5), 2)
        nhits = float((hit >= 0).sum().item())
        # parity gate, outside the timed region: sampled environments (first, last, middle, either side of work-item batch edges)
        # against the oracle -- distances bit-exact and the exact-hit index, for every layout that was timed
        from tests import gates
        sample = gates.spread_sample(E, 8, boundaries=(256, E // 4, E - 256))
        gate = gates.gate_rule_distance(prob, x, dists, hit, sample)
        gate["layouts"] = ["u16 index mirror" if compressed else "f64 columns"]
        if compressed:
            d2 = torch.empty_like(dists)
            h2 = torch.empty_like(hit)
            plain.rule_distance(x, ruledists=d2, hit=h2, stream=self.stream)
            torch.cuda.synchronize()
            gates.gate_rule_distance(plain, x, d2, h2, sample)
            assert (h2 == hit).all(), "the two layouts disagree on an exact-hit index"
            gate["layouts"].append("f64 columns")
            del d2, h2
        del dists
        evals = float(E) * R
        contract_bytes = 8.0 * (nant + 1) * evals              # the reference's f64 SoA layout, 8*nant read + 8 written per eval
        moved_bytes = (2.0 * nant + 8.0) * evals if compressed else contract_bytes
        achieved = moved_bytes / (kern_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "rule_distance_idx_kernel" if compressed else "rule_distance_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": moved_bytes, "bytes_per_eval": moved_bytes / evals, "avg_launch_ms": kern_ms,
                "layout": "u16 universe-index mirror + LDS VE tables (bit-identical to the f64 columns)" if compressed else "f64 SoA columns",
                "contract_bytes_per_launch": contract_bytes, "contract_equiv_GBps": contract_bytes / (kern_ms * 1e-3) / 1e9,
                "contract_frac": contract_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": ("achieved / frac = bytes this kernel moves (2*nant B of 16-bit indices read + 8 B written per eval) / avg launch time; "
                         "contract_* prices the same launch at the reference's f64 layout (8*(nant+1) B per eval) and may exceed the peak: "
                         "a compression effect, not a bandwidth") if compressed else "f64 layout: contract bytes == moved bytes"}
        if f64_ms:
            roof["f64_layout"] = {"kernel": "rule_distance_kernel", "avg_launch_ms": f64_ms, "evals_per_s": evals / (f64_ms * 1e-3),
                                  "achieved": contract_bytes / (f64_ms * 1e-3) / 1e9, "frac": contract_bytes / (f64_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "same sweep streaming the reference's f64 columns (contract bytes == moved bytes)"}
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(name)
            if tr and w["E"] == WORKLOADS[name]["E"]:
                roof["traffic"] = tr["fetch_bytes"] + tr["write_bytes"]
                roof["traffic_source"] = tr["source"]
                roof["traffic_kernel"] = tr.get("kernel")
        except (OSError, ValueError):
            pass
        return {"evals_per_s_rank": evals * steps / dt, "wall_s": dt, "exact_hits_rank": nhits, "parity_gate": gate}, roof

    def env_steps_leg(self, w, prob, agent, envs, nsteps, warmup, name=None):
        """The metric's second half: whole environment steps (do_action, reward, quantise, greedy sweep, SARSA update)."""
        import frirl_amd
        torch = self.torch
        E, R, nant = w["E"], w["R"], w["nant"]
        frirl_amd.episode_begin(prob, agent, envs, stream=self.stream)
        edt, ems = self.timed(lambda: frirl_amd.episode_step(prob, agent, envs, stream=self.stream), nsteps, warmup)
        # parity gate, outside the timed region: ONE more step of the whole batch; sampled environments replayed by the oracle from a
        # snapshot (state, chosen action, rule count and appended antecedents exact, consequents <= 1e-9)
        from tests import gates
        ns = nant - 1
        cell_before = envs.q_ant[:, :ns].clone()
        running = envs.done == 0
        gate = gates.gate_env_step(prob, agent, envs, w["env"], gates.spread_sample(E, 8, boundaries=(E // 4,)),
                                   lambda: frirl_amd.episode_step(prob, agent, envs, stream=self.stream))
        # environments that did not leave their quantisation cell in that step: the fused sweep of the 3-action shapes then takes the pending
        # conclusion from the greedy sweep's conclusion of the pending action (sweeps.h: SAMES); the slot model counts only what is computed
        same_cell = float(((cell_before == envs.q_ant[:, :ns]).all(1) & running).double().sum() / running.double().sum().clamp(min=1.0)) if w["A"] <= 8 else 0.0
        # per-episode reward statistics: the ONLY cross-rank exchange (RCCL all-reduce over xGMI when N > 1)
        st = self.D.allreduce_stats(envs.ep_reward, envs.ep_steps, envs.done, prob.nrules)
        status = torch.bincount(envs.status.long(), minlength=6).tolist()
        compressed = prob.uidx is not None and bool(frirl_amd.lib().frirl_hip_step_uses_uidx(prob.nant, prob.U, prob.maxR, prob.E))
        moved = float(E) * R * ((2.0 * nant if compressed else 8.0 * nant) + 8.0)         # one pass over the antecedents + consequents
        leg = {"value": float(E) * self.world * nsteps / edt, "unit": "env-steps/s", "steps": nsteps,
               "ms_per_step": edt / nsteps * 1e3, "avg_launch_ms": ems,
               "moved_bytes_per_step": moved, "moved_GBps": moved / (ems * 1e-3) / 1e9, "moved_frac": moved / (ems * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "contract_bytes_per_step": float(E) * (2.0 * R * (nant + 1) * 8 + R * 8),    # SURVEY 8d U2 (the reference's three sweeps, f64)
               "rule_action_evals_per_s": float(E) * R * (w["A"] + 1) / (ems * 1e-3),
               "fp64_issue": fp64_issue(E, R, nant, w["A"], ems, name, same_cell),
               "stats_allreduce": {"envs": st.envs, "mean_reward": st.mean_reward, "mean_rules": st.mean_rules, "steps_sum": st.steps_sum,
                                   "episodes_done": st.success, "reward_min": st.reward_min, "reward_max": st.reward_max},
               "last_step_outcomes_rank0": dict(zip(["inactive", "exact", "spread", "inserted", "skipped", "full"], status)),
               "parity_gate": gate,
               "note": "moved_* = one pass over the rule-base slabs per step (the fused sweep; compressed antecedents where the step kernel uses "
                       "them); fp64_issue = the step's second roofline: algorithmic FP64 vector-instruction slots of the fused sweep / the "
                       "chip's FP64 vector issue rate (many actions: issue-bound; 3 actions at 65 536 rules: between the two)"}
        return leg


FP64_VECTOR_PEAK_TFLOPS = 78.6        # MI355X FP64 vector peak (FMA = 2 flop): 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x 2


def fp64_issue(E, R, nant, A, ms, name=None, same_cell=0.0):
    """FP64-issue roofline of the fused step.  Algorithmic slots per rule (one slot = one FP64 vector instruction of one lane;
    v_rsq_f64 issues in 3.4 slots, profiles/r02_valu_cost.txt): per conclusion (A greedy + 1 pending) 2 for the squared distance,
    3.4 + 7 for the Shepard weight (rsq + series + power, sweeps.h: shepard_series), 2 for the two sums = 14.4; plus the state part
    (sub + fma per state dimension) and the rest of the pending update's full distance.  For the `same_cell` share of the environments
    (measured on the gate step: they did not leave their quantisation cell) the kernel does not compute the pending conclusion at all --
    it IS the greedy conclusion of the pending action -- so neither its 14.4 slots nor its state part are counted."""
    slots_per_rule = 14.4 * (A + 1) + 4.0 * (nant - 1) - same_cell * (14.4 + 2.0 * (nant - 1))
    peak = FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.0          # lane-instructions per second
    achieved = float(E) * R * slots_per_rule / (ms * 1e-3)
    counted = None
    try:        # the counter-derived figure beside the slot model: SQ_INSTS_VALU of this kernel from a committed PMC pass
        cv = json.load(open(os.path.join(ROOT, "profiles", "valu.json"))).get(name)
        if cv:
            lane_instr = cv["insts_valu"] * 64.0 * (float(E) / cv["envs"])
            counted = {"lane_instr_per_launch": lane_instr, "lane_instr_per_s": lane_instr / (ms * 1e-3), "frac": lane_instr / (ms * 1e-3) / peak,
                       "kernel": cv["kernel"], "source": cv["source"],
                       "note": "SQ_INSTS_VALU x 64 from the committed PMC pass / THIS run's launch time: every VALU instruction (FP64, integer, address) counts once"}
    except (OSError, ValueError, KeyError):
        pass
    return {"slots_per_rule": slots_per_rule, "same_cell_share": same_cell, "achieved_lane_instr_per_s": achieved, "peak_lane_instr_per_s": peak, "frac": achieved / peak, "counted": counted,
            "how": "frac = slot MODEL (algorithmic FP64 slots of the fused sweep / launch time / peak); counted = the same launch priced by the SQ_INSTS_VALU counter",
            "peak_source": "FP64 vector 78.6 TFLOP/s (half the FP32 vector peak of MI355X_MICROARCH.md: 16 lanes per clock and SIMD at 2.4 GHz) = 3.93e13 lane-instructions/s; a v_fma_f64 stream alone reaches 0.82 of it (profiles/r02_valu_cost.txt)"}


LEARN_AGENTS = 65536          # agents per GPU of the learning legs (the many-agent regime; weak scaling)


def issue_record(slots, seconds, extra):
    peak = FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.0
    rec = {"bound": "fp64-issue", "achieved_lane_instr_per_s": slots / seconds, "peak_lane_instr_per_s": peak, "frac": slots / seconds / peak,
           "how": "COUNTED work: rule visits accumulated by the kernel itself x algorithmic FP64 vector-instruction slots per visit (14.4 per conclusion: 2 squared "
                  "distance + 3.4 rsq + 7 series / power + 2 sums; + 2 per state dimension and sweep) / wall time / 3.93e13 lane-instructions/s"}
    rec.update(extra)
    return rec


def grid_start_states(d, E, device, seed):
    """One start state per agent on the state grid (the reference's many-agent modes give every agent its own start state:
    gen_def_states takes them from the master's rule antecedents, which lie on this grid; frirl_agent.c:121-139)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    cols = []
    for k in range(d["nstates"]):
        vals = torch.from_numpy(d["grids"][k]).to(device)
        cols.append(vals[torch.randint(0, len(vals), (E,), generator=g, device=device)])
    return torch.stack(cols, 1).contiguous()


def learning_and_evaluation(B, w, world, rank):
    """Legs 3-5: real learning -- E agents from the reference's initial 2^nant corner rule base until every rule base is "considered
    complete" (frirl_sequential_run's construct loop, all on the device) -- as REPLICAS of the demo (every agent the same start state:
    identical trajectories, the best case) and with DIVERSIFIED start states (the reference's many-agent regime: agents never in step,
    converging after anything between a few hundred and tens of thousands of steps, some not within max_episodes); and evaluation:
    greedy roll-outs of 65 536 environments on ONE shared rule base (the one agent 0 of the replica run learned)."""
    import torch
    import torch.distributed as dist
    import frirl_amd
    device, D = B.device, B.D
    env = w["env"]
    dd = frirl_amd.demo_describe(env)
    nant, A = dd["nant"], len(dd["action_ve"])
    wprob, wagent, wenvs = frirl_amd.demo_fresh_batch(env, 64, 1024, device)
    persistent = frirl_amd.learn_supported(wprob, wagent)
    legs = {}
    one = None
    if persistent:
        frirl_amd.train_persistent(wprob, wagent, wenvs, max_episodes=3, budget=64)      # untimed: loads the code objects
        del wprob, wenvs
        # the third leg holds four times as many agents as the chip keeps resident at two lanes each (HBM has room for far more): launches stay
        # full until the very end of the run, where the 65 536-agent leg spends 15 % of its time on its last few thousand agents
        for name, diversify, max_episodes, NAG in (("learning", False, 1000, LEARN_AGENTS), ("learning_diversified", True, 200, LEARN_AGENTS),
                                                   ("learning_diversified_4x", True, 200, 4 * LEARN_AGENTS)):
            start = grid_start_states(dd, NAG, device, 1 + rank) if diversify else None
            lprob, lagent, lenvs = frirl_amd.demo_fresh_batch(env, NAG, 1024 if NAG == LEARN_AGENTS else 512, device, start_states=start)
            lagent.desc.env_id_base = rank * NAG
            chunks = []

            def on_chunk(i, live, conv):
                # the report of the many-agent job (reward statistics only).  Ranks with diversified agents make DIFFERENT numbers of
                # launches, so nothing collective may happen per launch: the statistics are kept locally and cross the GPUs once, below
                st = torch.stack([lenvs.ep_reward.sum(), lenvs.ep_steps.sum().double(), lprob.nrules.sum().double(), conv.converged.sum().double()])
                chunks.append((NAG if live is None else int(live.numel()), st))
            B.sync_all()
            t0 = time.perf_counter()
            run = frirl_amd.train_persistent(lprob, lagent, lenvs, max_episodes=max_episodes, budget=512, on_chunk=on_chunk)
            B.sync_all()
            ldt = D.max_over_ranks(time.perf_counter() - t0, device)
            wk = run.work.sum(0).double()
            tot = torch.cat([torch.stack([run.steps_total.sum().double(), run.conv.converged.sum().double(), lprob.nrules.sum().double(),
                                          torch.tensor(float(run.conv.full_envs), device=device, dtype=torch.float64), wk[0], wk[1]]), chunks[-1][1]])
            if world > 1:
                dist.all_reduce(tot)                  # the ONE exchange of the leg: totals + the final report (RCCL over xGMI; latency-bound)
            tot = tot.tolist()
            slots = tot[4] * (14.4 * (A + 1) + 4.0 * (nant - 1)) + tot[5] * (2.0 * nant + 10.4)
            eps = run.conv.episodes
            legs[name] = {"value": tot[0] / ldt, "unit": "env-steps/s", "agents": NAG * world, "start_states": "per-agent, on the state grid" if diversify else "identical (replicas of the demo)",
                          "wall_s": ldt, "env_steps": tot[0], "agents_converged": tot[1], "max_episodes": max_episodes,
                          "episodes_min_max_rank0": [int(eps.min().item()), int(eps.max().item())], "mean_final_rules": tot[2] / (NAG * world),
                          "agents_with_refused_appends": tot[3], "launches": run.launches, "live_agents_per_launch_rank0": [c[0] for c in chunks][:64],
                          "report_allreduce": {"launches_rank0": len(chunks), "final": dict(zip(["reward_sum", "steps_sum", "rules_sum", "converged"], tot[6:10])),
                                               "note": "one all-reduce per leg: ranks make different numbers of launches"},
                          "kernel": "learn_kernel (persistent construct loop, csrc/learn_kernel.h)",
                          "fp64_issue": issue_record(slots, ldt, {"rule_visits_fused_sweeps": tot[4], "rule_visits_extra_sweeps": tot[5],
                                                                  "slots_per_fused_visit": 14.4 * (A + 1) + 4.0 * (nant - 1), "slots_per_extra_visit": 2.0 * nant + 10.4})}
            if not diversify:
                assert bool((run.conv.converged == 1).all()), "the replicas of the demo must all converge"
                one = frirl_amd.Problem(lprob.u, lprob.ve, lprob.rb[0:1].clone(), lprob.nrules[0:1].clone())
                eagent = lagent
            del lprob, lenvs, run
            torch.cuda.empty_cache()
        for nm in ("learning_diversified", "learning_diversified_4x"):
            legs[nm]["vs_replicas"] = legs[nm]["fp64_issue"]["frac"] / legs["learning"]["fp64_issue"]["frac"]      # counted work per second against the replica leg's
            legs[nm]["vs_replicas_env_steps"] = legs[nm]["value"] / legs["learning"]["value"]
    else:
        # shapes the persistent learner does not cover: one episode per launch through the lane groups of lanes.hip; no work counters in that kernel
        lE = 8192
        for name, diversify in (("learning", False), ("learning_diversified", True)):
            start = grid_start_states(dd, lE, device, 1 + rank) if diversify else None
            lprob, lagent, lenvs = frirl_amd.demo_fresh_batch(env, lE, 1024, device, start_states=start)
            lsteps = torch.zeros((), dtype=torch.int64, device=device)

            def on_ep(ep, conv):
                lsteps.add_((lenvs.ep_steps.long() * (conv.episodes == ep).long()).sum())
            frirl_amd.episode_run_lanes(lprob, lagent, lenvs, 0)
            B.sync_all()
            t0 = time.perf_counter()
            conv = frirl_amd.train(lprob, lagent, lenvs, on_episode=on_ep, max_episodes=200 if diversify else 1000)
            B.sync_all()
            ldt = D.max_over_ranks(time.perf_counter() - t0, device)
            tot = torch.tensor([float(lsteps.item()), float(conv.converged.sum().item()), float(lprob.nrules.sum().item())], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(tot)
            legs[name] = {"value": tot[0].item() / ldt, "unit": "env-steps/s", "agents": lE * world, "wall_s": ldt, "env_steps": tot[0].item(),
                          "agents_converged": tot[1].item(), "mean_final_rules": tot[2].item() / (lE * world),
                          "start_states": "per-agent, on the state grid" if diversify else "identical (replicas of the demo)",
                          "kernel": "episode_run_lanes (lane groups, one episode per launch)", "fp64_issue": None}
            if not diversify:
                one = frirl_amd.Problem(lprob.u, lprob.ve, lprob.rb[0:1].clone(), lprob.nrules[0:1].clone())
                eagent = lagent
            del lprob, lenvs
            torch.cuda.empty_cache()

    ns = nant - 1
    Qn = 65536
    g = torch.Generator(device=device)
    g.manual_seed(7 + rank)
    lo = torch.tensor([dd["grids"][k].min() for k in range(ns)], dtype=torch.float64, device=device)
    hi = torch.tensor([dd["grids"][k].max() for k in range(ns)], dtype=torch.float64, device=device)
    vd = torch.tensor([dd["values_def"][k] for k in range(ns)], dtype=torch.float64, device=device)
    ss = (vd + (torch.rand((Qn, ns), dtype=torch.float64, device=device, generator=g) - 0.5) * 0.2 * (hi - lo)).clamp(lo, hi).contiguous()
    one.rollout_shared(eagent, Qn, start_states=ss)
    B.sync_all()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        rsteps, rrew, rsucc, _ = one.rollout_shared(eagent, Qn, start_states=ss)
    B.sync_all()
    edt = D.max_over_ranks(time.perf_counter() - t0, device) / reps
    # a STREAM of evaluation calls: calls alternate between two HIP streams, so that one call's straggler tail (a few long episodes on a
    # handful of waves) overlaps the next call's bulk -- the call itself is unchanged (stream-ordered, no host round trip)
    s2 = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
    for st_ in s2:
        with torch.cuda.stream(st_):
            one.rollout_shared(eagent, Qn, start_states=ss, stream=st_)
    B.sync_all()
    t0 = time.perf_counter()
    for i in range(2 * reps):
        with torch.cuda.stream(s2[i & 1]):
            one.rollout_shared(eagent, Qn, start_states=ss, stream=s2[i & 1])
    B.sync_all()
    pdt = D.max_over_ranks(time.perf_counter() - t0, device) / (2 * reps)
    Rone = int(one.nrules[0].item())
    et = torch.tensor([float(rsteps.sum().item()), float((rsucc == 1).sum().item())], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(et)
    sweeps = et[0].item() + Qn * world                      # one greedy sweep per step + the first action's
    # parity gate: sampled roll-outs replayed by the oracle's frirl_test_run episode on the same rule base (steps and reward exact)
    gate = rollout_gate(one, eagent, env, ss, rsteps, rrew, [0, 1, Qn // 2, Qn - 1] + [int(i) for i in torch.topk(rsteps, 4).indices.tolist()])
    legs["evaluation"] = {"value": et[0].item() / edt, "unit": "env-steps/s", "environments": Qn * world, "rules": Rone, "wall_s": edt, "env_steps": et[0].item(),
                          "episodes_succeeded": et[1].item(), "longest_episode_rank0": int(rsteps.max().item()), "parity_gate": gate,
                          "kernel": "rollout_resident_kernel (csrc/rollout.hip)" if frirl_amd.lib().frirl_hip_rollout_resident_rules(nant, A, 0, eagent.desc.env_kind) >= Rone else "rollout_shared_kernel",
                          "fp64_issue": issue_record(sweeps * Rone * (14.4 * A + 2.0 * (nant - 1)), edt,
                                                     {"rule_visits": sweeps * Rone, "slots_per_visit": 14.4 * A + 2.0 * (nant - 1),
                                                      "counted_from": "steps[] returned by the kernel: one greedy sweep over the rule base per step + one per episode start"}),
                          "pipelined": {"value": et[0].item() / pdt, "unit": "env-steps/s", "wall_s_per_call": pdt,
                                        "fp64_issue_frac": sweeps * Rone * (14.4 * A + 2.0 * (nant - 1)) / pdt / (FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.0),
                                        "how": "the same call 10 times, alternating between two HIP streams: one call's straggler tail overlaps the next call's bulk"},
                          "note": "frirl_hip_rollout_shared: whole greedy episodes from perturbed start states on one shared rule base, no updates; value = average of 5 back-to-back calls on one stream"}
    # the same call with sixteen times the environments: the launch is queue-fed (four waves per SIMD), the few never-ending episodes no longer
    # set the call's length -- the throughput regime of the roll-out kernels, next to the 65 536-environment call above
    Qb = 16 * Qn
    sb = (vd + (torch.rand((Qb, ns), dtype=torch.float64, device=device, generator=g) - 0.5) * 0.2 * (hi - lo)).clamp(lo, hi).contiguous()
    one.rollout_shared(eagent, Qb, start_states=sb)
    B.sync_all()
    t0 = time.perf_counter()
    for _ in range(3):
        bsteps, brew, bsucc, _ = one.rollout_shared(eagent, Qb, start_states=sb)
    B.sync_all()
    bdt = D.max_over_ranks(time.perf_counter() - t0, device) / 3
    bt = torch.tensor([float(bsteps.sum().item())], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(bt)
    bsweeps = bt[0].item() + Qb * world
    bgate = rollout_gate(one, eagent, env, sb, bsteps, brew, [0, Qb // 3, Qb - 1] + [int(i) for i in torch.topk(bsteps, 2).indices.tolist()])
    legs["evaluation_16x"] = {"value": bt[0].item() / bdt, "unit": "env-steps/s", "environments": Qb * world, "rules": Rone, "wall_s": bdt, "env_steps": bt[0].item(),
                              "parity_gate": bgate,
                              "fp64_issue": issue_record(bsweeps * Rone * (14.4 * A + 2.0 * (nant - 1)), bdt,
                                                         {"rule_visits": bsweeps * Rone, "slots_per_visit": 14.4 * A + 2.0 * (nant - 1)})}
    return legs


def learning_other_env(B, env, agents=LEARN_AGENTS):
    """The learning legs on another demo (N = 1 only, beside the other configurations): `agents` replicas to convergence and the same
    number of agents with one start state each (200 episodes), through frirl_hip_learn_train; counted FP64-issue fraction as above."""
    import torch
    import frirl_amd
    device = B.device
    dd = frirl_amd.demo_describe(env)
    nant, A = dd["nant"], len(dd["action_ve"])
    wprob, wagent, wenvs = frirl_amd.demo_fresh_batch(env, 64, 1024, device)
    if not frirl_amd.learn_supported(wprob, wagent):
        return None
    frirl_amd.train_persistent(wprob, wagent, wenvs, max_episodes=3, budget=64)          # untimed: loads the code objects
    del wprob, wenvs
    out = {}
    for name, diversify, max_episodes in (("replicas", False, 1000), ("diversified", True, 200)):
        start = grid_start_states(dd, agents, device, 1) if diversify else None
        lprob, lagent, lenvs = frirl_amd.demo_fresh_batch(env, agents, 1024, device, start_states=start)
        B.sync_all()
        t0 = time.perf_counter()
        run = frirl_amd.train_persistent(lprob, lagent, lenvs, max_episodes=max_episodes, budget=512)
        B.sync_all()
        dt = time.perf_counter() - t0
        wk = run.work.sum(0).double().tolist()
        steps = float(run.steps_total.sum().item())
        slots = wk[0] * (14.4 * (A + 1) + 4.0 * (nant - 1)) + wk[1] * (2.0 * nant + 10.4)
        out[name] = {"value": steps / dt, "unit": "env-steps/s", "agents": agents, "wall_s": dt, "env_steps": steps, "launches": run.launches,
                     "agents_converged": float(run.conv.converged.sum().item()), "mean_final_rules": float(lprob.nrules.double().mean().item()),
                     "agents_with_refused_appends": float(run.conv.full_envs),
                     "fp64_issue_frac_counted_work": slots / dt / (FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.0)}
        del lprob, lenvs, run
        torch.cuda.empty_cache()
    return out


def rollout_gate(one, agent, env, ss, rsteps, rrew, sample):
    """Sampled roll-outs against the oracle's frirl_test_run episode (portable trig) on the same rule base."""
    import numpy as np
    from oracle import binding as ob
    nant = one.nant
    R = int(one.nrules[0].item())
    fr = ob.Frirl(env, trig_mode=1, maxR=max(1024, R + 8))
    f = fr.five
    while f.R:
        assert ob.lib().orc_remove_rule(f.h, 0) == 0
    u = one.u.cpu().numpy()
    rb = one.rb[0].cpu().numpy()
    # raw antecedents from the VE columns: every stored antecedent is a universe point (five_add_rule.c:76-81)
    ve = one.ve.cpu().numpy()
    for r in range(R):
        rant = [u[k][int(np.nonzero(ve[k] == rb[k, r])[0][0])] for k in range(nant)]
        assert f.add_rule(np.array(rant), rb[nant, r]) == 0
    assert (f.veval[:, :R] == rb[:nant, :R]).all()
    st, rw, s0 = rsteps.cpu().numpy(), rrew.cpu().numpy(), ss.cpu().numpy()
    sample = sorted(set(sample))
    for i in sample:
        fr.set_start_state(s0[i])
        fr.episode_eval()
        assert st[i] == fr.ep_steps, f"roll-out {i}: {st[i]} steps, oracle {fr.ep_steps}"
        assert abs(rw[i] - fr.ep_reward) <= 1e-9 * max(1.0, abs(fr.ep_reward)), f"roll-out {i}: reward {rw[i]}, oracle {fr.ep_reward}"
    return {"checked": len(sample), "ok": True, "environments": sample, "what": "steps and total reward of whole roll-outs vs the oracle's frirl_test_run episode (longest episodes included)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--env-steps", type=int, default=20, help="timed environment steps of the fused SARSA leg")
    ap.add_argument("--no-env-steps", action="store_true")
    ap.add_argument("--no-learn", action="store_true", help="skip the real-learning and evaluation legs")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg2 / cfg3 / cfg5 sub-records")
    ap.add_argument("--rank-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # parent: launch only, never touches the GPU
    if args.rank_probe:
        return rank_probe()

    import torch
    import torch.distributed as dist
    import frirl_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch through `python bench.py --gpus N` or torch.distributed.run with N ranks")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FRIRL_DIST_BACKEND", "nccl")     # "gloo" only to rehearse N > 1 on a one-GPU box
    assert backend == "gloo" or local_rank < ndev, f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible"
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", local_rank % ndev)
    dist_info = {"dist_backend": "none", "world_size": 1}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        dist_info = {"dist_backend": dist.get_backend() + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size()}

    if not (os.path.exists(frirl_amd.HIP_LIB_PATH) and os.path.exists(frirl_amd.DROPIN_LIB_PATH)):
        frirl_amd.build()          # normally prebuilt by __graft_entry__.build(); never rebuilt per rank
    D = frirl_amd.dist()
    B = Bench(device, world, rank, D)
    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w["E"] = args.envs
    E, R, nant = w["E"], w["R"], w["nant"]
    env_start, _ = D.shard(E * world, world, rank)            # this rank's slice of the global env ids

    # ---- legs 1 + 2 on the headline workload -------------------------------------------------------------------------
    prob, agent, envs = B.build(w, env_start)
    x = make_queries(prob, device, seed=env_start)
    val, roof = B.rule_distance_leg(args.workload, w, prob, x, args.steps, args.warmup)
    nhits = torch.tensor([val["exact_hits_rank"]], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(nhits)
    env_leg = None
    if agent is not None and not args.no_env_steps:
        env_leg = B.env_steps_leg(w, prob, agent, envs, args.env_steps, min(args.warmup, 3), args.workload)
    del prob, agent, envs, x
    torch.cuda.empty_cache()

    # ---- legs 3 + 4 ----------------------------------------------------------------------------------------------
    legs = None
    if w["env"] and not args.no_learn:
        legs = learning_and_evaluation(B, w, world, rank)
        torch.cuda.empty_cache()

    # ---- the other BASELINE configurations, same two legs, shorter (N = 1 only: they are per-GPU workloads) -----------
    others = None
    if world == 1 and not args.no_other_configs and not args.envs:
        others = {}
        for name, ow in WORKLOADS.items():
            if name == args.workload:
                continue
            ow = dict(ow)
            t0 = time.perf_counter()
            oprob, oagent, oenvs = B.build(ow, 0, keep_rant=False)
            ox = make_queries(oprob, device, seed=0)
            oval, oroof = B.rule_distance_leg(name, ow, oprob, ox, max(args.steps // 2, 10), 3)
            rec = {"config": {"nant": ow["nant"], "universe_len": ow["U"], "rules_per_env": ow["R"], "envs_per_gpu": ow["E"]},
                   "value": oval["evals_per_s_rank"], "unit": "evals/s", "roofline": oroof, "parity_gate": oval["parity_gate"]}
            if oagent is not None and not args.no_env_steps:
                rec["env_steps"] = B.env_steps_leg(ow, oprob, oagent, oenvs, max(args.env_steps // 2, 5), 2, name)
                rec["env_steps"].pop("stats_allreduce", None)
            rec["wall_s_incl_setup"] = round(time.perf_counter() - t0, 1)
            del oprob, oagent, oenvs, ox
            torch.cuda.empty_cache()
            if ow.get("env") and ow["env"] != w.get("env") and not args.no_learn:
                rec["learning"] = learning_other_env(B, ow["env"])          # the demo's many-agent learning run (replicas / one start state each)
            others[name] = rec

    if rank == 0:
        evals = float(E) * R * args.steps * world
        dt = val["wall_s"]
        out = {
            "metric": "rule-distance evals/sec (rules x envs)", "value": evals / dt, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "device": torch.cuda.get_device_name(device),
            "config": {"workload": args.workload, "nant": nant, "universe_len": w["U"], "rules_per_env": R, "envs_per_gpu": E,
                       "sharding": f"env ids split over {world} rank(s), no data-path collective; reward statistics all-reduced",
                       "exact_hits": int(nhits.item()), **dist_info},
            "roofline": roof,
        }
        if env_leg:
            out["env_steps"] = env_leg
        out["parity_gate"] = val["parity_gate"]
        if legs:
            out.update(legs)
        if others:
            out["other_configs"] = others
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
