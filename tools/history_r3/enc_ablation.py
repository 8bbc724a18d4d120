#!/usr/bin/env python3
"""Encode-only timing of a library build (no verification: DGPU_ENC_ABLATE builds produce wrong archives)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, dietgpu_amd as dg
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "u8"
data, ft, _, P, desc = bench.make_workload(wl, 256, 1234, dev)
c = bench.Codec(dg, data, ft, P)
for _ in range(30): c.encode()
torch.cuda.synchronize()
prof = bench.kernel_profile(c, 200, lambda i: c.encode())
print(os.path.basename(os.environ.get("DGPU_LIB") or "base"), wl, {n[6:]: round(r["total_ms"] / max(r["launches"], 1) * 1e3, 1) for n, r in prof.items()})
