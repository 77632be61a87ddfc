#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / LDS / occupancy of every gfx950 kernel (hipcc -Rpass-analysis)."""
import os, re, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "gaussian-splatting-wgpu_amd", "csrc")
for f in ("k_preprocess", "k_binning", "k_sort", "k_blend"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                          os.path.join(CSRC, f + ".hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            if cur: print(cur)
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
            cur = "%-46s" % name[:46]
        else:
            cur += " %s=%s" % (k.split(" ")[0], v)
    if cur: print(cur)
