#!/usr/bin/env python3
"""bench.py -- throughput of the FRIRL hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: five_hip_rule_distance over E independent
rule bases of R rules each (E*R rule-distance evaluations, distances materialised as the reference's
five_rule_distance does).  Default workload = BASELINE.json configs[1]: mountaincar-shaped
(nant 3, U 41), 8192 rules x 8192 environments per GPU.  Environments are sharded over ranks with
no data-path collective (weak scaling); value = all ranks' evaluations / max-over-ranks time.
Inputs are synthetic, generated on the device and resident in HBM before the timed region.
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

WORKLOADS = {   # name -> nant, U, R, E(per GPU), A
    "cfg2_mountaincar_8k_x_8k": dict(nant=3, U=41, R=8192, E=8192, A=3),
    "cfg3_cartpole_32k_x_32k": dict(nant=5, U=1001, R=32768, E=32768, A=21),
    "cfg4_acrobot_64k_x_8k_per_gpu": dict(nant=5, U=41, R=65536, E=8192, A=3),
    "cfg5_synth16_256k": dict(nant=16, U=1001, R=262144, E=64, A=0),
}


def synth_problem(w, device, seed):
    """SURVEY 8d synthetic inputs, generated on the device: shared universes / VE tables, one private rule
    base per environment with on-grid antecedents (uniform universe indices; the action column uses the A
    action grid points when A > 0) and Q ~ U(-1500, 1500).  Duplicates are not removed (they only matter
    for exact-hit tie-breaking, which is defined: lowest index)."""
    import numpy as np
    import torch
    import frirl_amd
    nant, U, R, E, A = w["nant"], w["U"], w["R"], w["E"], w["A"]
    g = torch.Generator(device=device).manual_seed(0x5EED0000 + seed)
    rng = np.random.default_rng(1234)
    u = np.zeros((nant, U))
    ve = np.zeros((nant, U))
    for k in range(nant):
        div = 2.0 * (k + 1) / (U - 1)
        half = [-(U - 1) * div / 2 + div * i for i in range(U // 2 + 1)]
        row = half + [-half[U - 1 - i] for i in range(U // 2 + 1, U)]
        u[k] = row
        scf = 0.5 + rng.random(U)
        ve[k, 1:] = np.cumsum((u[k, 1:] - u[k, :-1]) * (scf[:-1] + scf[1:]) * 0.5)
    u_d, ve_d = torch.from_numpy(u).to(device), torch.from_numpy(ve).to(device)
    rb = torch.empty((E, nant + 1, R), dtype=torch.float64, device=device)
    for k in range(nant):
        if A > 0 and k == nant - 1:
            idx = (torch.randint(0, A, (E, R), generator=g, device=device) * (U - 1)) // max(A - 1, 1)
        else:
            idx = torch.randint(0, U, (E, R), generator=g, device=device)
        rb[:, k, :] = ve_d[k][idx]
        del idx
    rb[:, nant, :] = torch.rand((E, R), generator=g, device=device, dtype=torch.float64) * 3000.0 - 1500.0
    nrules = torch.full((E,), R, dtype=torch.int32, device=device)
    prob = frirl_amd.Problem(u_d, ve_d, rb, nrules)
    lo = u_d[:, 0]
    hi = u_d[:, U - 2]
    x = (lo + (hi - lo) * torch.rand((E, nant), generator=g, device=device, dtype=torch.float64)).contiguous()
    # 1 % exact grid hits to exercise the index path
    nh = max(1, E // 100)
    pick = torch.randint(0, R, (nh,), generator=g, device=device)
    for j in range(nh):
        vals = rb[j, :nant, pick[j]]
        # invert ve -> u exactly: the VE tables are strictly increasing
        x[j] = torch.stack([u_d[k][torch.searchsorted(ve_d[k], vals[k])] for k in range(nant)])
    return prob, x


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
        return dict(value=rec["rule_distance_evals_per_s"], unit="rule-distance evals/s", cores=1, kind="reference",
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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import frirl_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)      # RCCL over xGMI
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    frirl_amd.build()
    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w["E"] = args.envs
    prob, x = synth_problem(w, device, seed=rank)
    E, R, nant = w["E"], w["R"], w["nant"]
    dists = torch.empty((E, prob.maxR), dtype=torch.float64, device=device)
    hit = torch.empty((E,), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream()

    def step():
        prob.rule_distance(x, ruledists=dists, hit=hit, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # same stream as the launches
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps            # average launch duration incl. the 4*E-byte hit memset
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    stats = torch.tensor([float((hit >= 0).sum().item()), float(E)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)        # statistics only; no data-path collective
    dt = float(tmax.item())

    if rank == 0:
        evals = float(E) * R * args.steps * world
        alg_bytes = 8.0 * (nant + 1) * E * R                 # SURVEY 8d U1, materialised form: 8*nant read + 8 written per eval
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "rule-distance evals/sec (rules x envs)", "value": evals / dt, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "nant": nant, "universe_len": w["U"], "rules_per_env": R, "envs_per_gpu": E,
                       "sharding": f"envs x{world}, no data-path collective", "exact_hits": int(stats[0].item())},
            "roofline": {"bound": "hbm", "kernel": "rule_distance_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": kern_ms},
        }
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
