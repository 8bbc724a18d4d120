#!/usr/bin/env python3
"""Debug tool: per-phase tick counts inside k_ans_encode_fused (needs a -DDGPU_PHASE_TIMING build).
Usage (GPU box): python tools/phase_timing_fused.py [bf16|u8|fp16|fp32]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = "/tmp/libdietgpu_amd_dbg.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-DDGPU_PHASE_TIMING"] + os.environ.get("DGPU_EXTRA_FLAGS", "").split() + ["-o", DBG, os.path.join(ROOT, "dietgpu_amd/csrc/capi.hip")])
import dietgpu_amd.build as b
b.LIB_PATH = DBG
import dietgpu_amd._lib as L
L.LIB_PATH = DBG
import dietgpu_amd as dg
import bench

lib = C.CDLL(DBG)
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(os.environ.get("DGPU_PT_BATCH", "256"))
N = int(os.environ.get("DGPU_PT_ELEMS", str(512 * 1024)))
data, ft, _, P, desc = bench.make_workload(wl, B, 1234, dev, N)
codec = bench.Codec(dg, data, ft, P)
for _ in range(3):
    codec.step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ntiles = B * max(1, data.shape[1] // 32768) + 64
buf = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
assert lib.dgpu_debug_set_phase_buffer(C.c_void_p(buf.data_ptr())) == 0
ev[0].record()
codec.encode()
ev[1].record()
torch.cuda.synchronize()
print("encode call:", ev[0].elapsed_time(ev[1]) * 1e3, "us")
t = buf.cpu().numpy()
t = t[t[:, 0] != 0]
names = ["wait table + fetch", "draw + chunk loop (rows + next tile load/hist)", "publish histogram (+normalise)", "states + look-back", "copy-out"]
print(f"{wl}: {t.shape[0]} tiles; ticks of the constant-rate counter (100 MHz => 1 tick = 10 ns)")
for k, n in enumerate(names):
    d = (t[:, k + 1] - t[:, k]).astype(np.float64)
    print(f"  {n:44s} mean {d.mean():8.0f}  p50 {np.median(d):8.0f}  p95 {np.percentile(d, 95):8.0f}")
tot = (t[:, 5] - t[:, 0]).astype(np.float64)
print(f"  {'total per tile':44s} mean {tot.mean():8.0f}")
wg = t[:, 7]
xcd = wg % 8
for x in range(8):
    r = t[xcd == x]
    if len(r):
        print(f"    xcd {x}: tiles {len(r)} span {int(r[:, 5].max() - r[:, 0].min())}")
counts = [int((wg == w).sum()) for w in np.unique(wg)]
print("  workgroups", len(counts), "tiles/WG min", min(counts), "max", max(counts))
