#!/usr/bin/env python3
"""Per-workgroup timeline of one fused step (experiment build: tools/build_variant.sh timing -DFRIRL_STEP_TIMING, loaded through
FRIRL_HIP_LIB_OVERRIDE):  tools/step_timing.py <workload> [E]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, frirl_amd, bench

w = dict(bench.WORKLOADS[sys.argv[1]])
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    w["E"] = int(sys.argv[2])
dev = torch.device("cuda", 0)
prob, agent, envs = frirl_amd.demo_batch(w["env"], w["E"], w["R"], w["R"] + 256, dev, seed=0, keep_rant=False)
frirl_amd.episode_begin(prob, agent, envs)
for _ in range(4):
    frirl_amd.episode_step(prob, agent, envs)
torch.cuda.synchronize()
n = min(w["E"], 65536)
buf = np.zeros(4 * n, dtype=np.int64)
lib = C.CDLL(frirl_amd.HIP_LIB_PATH)
assert lib.frirl_hip_debug_step_timing(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.reshape(n, 4).astype(np.float64)
t0 = t[:, 0].min()
start, sweep, end, st = (t[:, 0] - t0) * 0.01, (t[:, 1] - t[:, 0]) * 0.01, (t[:, 2] - t[:, 0]) * 0.01, t[:, 3].astype(int)
print(f"{sys.argv[1]} E={n}: kernel span {((t[:, 2].max() - t0) * 0.01):.1f} us; workgroup start times: median {np.median(start):.1f} us, 90% {np.percentile(start, 90):.1f}, max {start.max():.1f}")
print(f"  fused sweep per workgroup: median {np.median(sweep):.1f} us, 90% {np.percentile(sweep, 90):.1f}, max {sweep.max():.1f}")
print(f"  whole workgroup: median {np.median(end):.1f} us, 90% {np.percentile(end, 90):.1f}, 99% {np.percentile(end, 99):.1f}, max {end.max():.1f}")
for code, name in enumerate(["inactive", "exact", "spread", "inserted", "skipped", "full"]):
    m = st == code
    if m.any():
        print(f"  outcome {name:9s}: {int(m.sum()):6d} workgroups, lifetime median {np.median(end[m]):.1f} us, max {end[m].max():.1f}; update part median {np.median((end - sweep)[m]):.1f} us")
# how many workgroups are alive over time
ev = np.concatenate([np.stack([start, np.ones(n)], 1), np.stack([start + end, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
span = ev[-1, 0]
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    i = np.searchsorted(ev[:, 0], frac * span)
    print(f"  alive at {frac:.0%} of the span: {int(alive[min(i, len(alive) - 1)])} workgroups")
