#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into a small markdown summary (profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ONLY_OURS = "--all" not in sys.argv      # torch's input-generation kernels are noise in these summaries


def ours(name):
    return (not ONLY_OURS) or (name is not None and "frirl::" in name)


def rows(pattern):
    for p in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                yield p, r


print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
bj = os.path.join(out, "bench_trace.json")
if os.path.exists(bj):
    try:
        import json
        rec = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][-1])
        rf = rec["roofline"]
        print(f"bench line of the profiled run: workload {rec['config']['workload']} (E {rec['config']['envs_per_gpu']}, R {rec['config']['rules_per_env']}), "
              f"{rec['value']:.4g} evals/s, kernel {rf['kernel']}: avg launch {rf['avg_launch_ms']:.4f} ms (HIP events), moved {rf['algorithmic_bytes_per_launch']:.4g} B "
              f"-> {rf['achieved']:.0f} GB/s = {rf['frac']:.3f} of 8 TB/s" + (f"; f64 layout {rf['f64_layout']['avg_launch_ms']:.4f} ms = {rf['f64_layout']['frac']:.3f}" if 'f64_layout' in rf else "") + "\n")
    except Exception as ex:      # keep the raw line if the format changes
        print("bench line (profiled run):\n```\n" + open(bj).read().strip()[:3000] + "\n```\n")

# kernel stats from the kernel trace
dur = defaultdict(list)
meta = {}
for p, r in rows("trace/**/*kernel_trace.csv"):
    name = r.get("Kernel_Name") or r.get("Name")
    if not ours(name):
        continue
    try:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    except (KeyError, ValueError):
        continue
    dur[name].append(d)
    meta[name] = (r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size") or r.get("Grid_Size_X"),
                  r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
print("## kernel trace (--kernel-trace --stats)\n")
print("| kernel | calls | total ms | avg us | min us | max us | VGPR | SGPR | LDS | grid | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot = sum(sum(v) for v in dur.values()) or 1
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    m = meta[name]
    short = name if len(name) < 90 else name[:87] + "..."
    print(f"| `{short}` | {len(v)} | {sum(v) / 1e6:.3f} ({100 * sum(v) / tot:.1f}%) | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} |")
print()

# PMC passes
for ctr, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    acc = defaultdict(list)
    for p, r in rows(pat):
        if r.get("Counter_Name") != ctr or not ours(r.get("Kernel_Name")):
            continue
        acc[r.get("Kernel_Name")].append(float(r["Counter_Value"]))
    if not acc:
        continue
    print(f"## PMC {ctr} (own pass; raw counter is in KiB)\n")
    print("| kernel | dispatches | avg raw | avg bytes (raw*1024) | gfx950-corrected bytes |")
    print("|---|---|---|---|---|")
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        corr = avg * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {len(v)} | {avg:.1f} | {avg * 1024:.4g} | {corr:.4g} |")
    print("\nFETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE is exact for 16-B/lane streaming stores.\n")

# SQ / LDS counter passes (sums over all dispatches of a kernel; percentages relative to SQ_WAVE_CYCLES of the same pass)
for sub in ("pmc_sq", "pmc_lds"):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    first = None
    for p, r in rows(sub + "/**/*counter_collection.csv"):
        k = r.get("Kernel_Name")
        if not ours(k):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        first = first or r["Counter_Name"]
        if r["Counter_Name"] == first:
            calls[k] += 1
    if not acc:
        continue
    print(f"## PMC {sub} (per dispatch averages)\n")
    key = "SQ_WAVE_CYCLES" if sub == "pmc_sq" else "SQ_LDS_IDX_ACTIVE"
    for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:4]:
        n = max(calls[k], 1)
        short = k if len(k) < 90 else k[:87] + "..."
        print(f"`{short}` ({n} dispatches)\n")
        print("| counter | per dispatch | vs " + key + " |")
        print("|---|---|---|")
        base = c.get(key, 0.0) or 1.0
        for name, v in sorted(c.items()):
            print(f"| {name} | {v / n:.4g} | {100 * v / base:.1f}% |")
        print()
