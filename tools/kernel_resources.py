#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in a -Rpass-analysis=kernel-resource-usage log:
   hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> log ; tools/kernel_resources.py log [name-filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: (?:\S+ )?Function Name: ", t)[1:]:
    mangled = b.split()[0]
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt not in name:
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    print(f"{name[:110]:110s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} spilled {g('VGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
