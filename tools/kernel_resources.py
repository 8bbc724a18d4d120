#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os
import re
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run(
    ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
     "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/capi_res.o",
     os.path.join(root, "dietgpu_amd/csrc/capi.hip")],
    capture_output=True, text=True)
rows, cur = [], None
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = re.sub(r"^_ZN4dgpu\d+", "", r["name"])[:36]
    g = lambda k: r.get(k, "?")
    print("%-36s vgpr %4s agpr %3s sgpr %4s scratch %4s occ %2s lds %6s" % (
        n, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
        g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
