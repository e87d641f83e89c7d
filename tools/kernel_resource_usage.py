#!/usr/bin/env python3
"""One line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks (runs here, no GPU):
python tools/kernel_resource_usage.py > profiles/r0N_kernel_resource_usage.txt"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "circkit_amd", "csrc", "circkit_hip.hip")
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Wno-inline-asm", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"],
                   capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.split("\n"):
    m = re.search(r"remark: (.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", c["name"])
    print("%-80s sgpr %s vgpr %s spillS %s spillV %s scratch %s occ %s lds %s" % (name[:80], c.get("TotalSGPRs"), c.get("VGPRs"), c.get("SGPRs Spill"), c.get("VGPRs Spill"),
          c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
