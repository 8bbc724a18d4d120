#!/usr/bin/env python3
"""Debug tool: per-phase cycle counts inside k_ans_encode (needs a -DDGPU_PHASE_TIMING build).

Builds a debug library next to the product one, runs the bf16 bench workload
once and prints the mean s_memtime delta of each phase per tile."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.environ.get("DGPU_PT_LIB", "/tmp/libdietgpu_amd_dbg.so")  # DGPU_PT_LIB: a prebuilt -DDGPU_PHASE_TIMING library
if "DGPU_PT_LIB" not in os.environ:
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
PT_B = int(os.environ.get("DGPU_PT_BATCH", "256"))
PT_N = int(os.environ.get("DGPU_PT_ELEMS", str(512 * 1024)))
data, ft, _, P, desc = bench.make_workload(wl, PT_B, 1234, dev, PT_N)
codec = bench.Codec(dg, data, ft, P)
codec.lib = dg.lib()
for _ in range(3):
    codec.step()
torch.cuda.synchronize()
ntiles = max(256 * 64, PT_B * ((PT_N + 32767) // 32768) + 64)
buf = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
assert lib.dgpu_debug_set_phase_buffer(C.c_void_p(buf.data_ptr())) == 0
codec.encode()
torch.cuda.synchronize()
t = buf.cpu().numpy()
slot = np.arange(t.shape[0])
keep = t[:, 0] != 0
elem = (slot % PT_B)[keep]
t = t[keep]
print("tiles whose element b has b % 8 == XCD of the workgroup (blockIdx % 8):",
      int(((elem % 8) == (t[:, 7] % 8)).sum()), "of", t.shape[0])
names = ["ticket->table", "rows", "states+sync", "lookback", "copy"]
print(f"{wl}: {t.shape[0]} tiles")
for k, n in enumerate(names):
    d = (t[:, k + 1] - t[:, k]).astype(np.float64)
    print(f"  {n:16s} mean {d.mean():9.0f}  p50 {np.median(d):9.0f}  p95 {np.percentile(d, 95):9.0f} cycles (s_memtime @100MHz? ticks)")
tot = (t[:, 5] - t[:, 0]).astype(np.float64)
print(f"  {'total':16s} mean {tot.mean():9.0f}")
# the cycle counters of different XCDs are not aligned: spans are per XCD
xcd = t[:, 7] % 8
spans = [int(t[xcd == x][:, 5].max() - t[xcd == x][:, 0].min()) for x in range(8) if (xcd == x).any()]
print("  span per XCD (max end - min start):", spans)
for x in range(8):
    r = t[xcd == x]
    if len(r):
        t0x = r[:, 0].min()
        starts = np.sort(r[:, 0] - t0x)
        print(f"    xcd {x}: tiles {len(r)}  tile starts p10/p50/p90 {np.percentile(starts,10):.0f}/{np.percentile(starts,50):.0f}/{np.percentile(starts,90):.0f}")
# per-workgroup timelines (slot 7 = blockIdx.x): idle time between consecutive tiles, start skew
wg = t[:, 7]
t0 = t[:, 0].min()
gaps, firsts, lasts, counts = [], [], [], []
for w in np.unique(wg):
    r = t[wg == w]
    r = r[np.argsort(r[:, 0])]
    firsts.append(r[0, 0] - t0)
    lasts.append(r[-1, 5] - t0)
    counts.append(len(r))
    gaps += list(r[1:, 0] - r[:-1, 5])
# rows phase and total by the tile's ordinal within its workgroup (0 = the tile a workgroup starts with)
by_ord = {}
for w in np.unique(wg):
    r = t[wg == w]
    r = r[np.argsort(r[:, 0])]
    for k in range(len(r)):
        by_ord.setdefault(k, []).append((r[k, 2] - r[k, 1], r[k, 4] - r[k, 3], r[k, 5] - r[k, 0]))
for k in sorted(by_ord):
    a = np.array(by_ord[k], dtype=np.float64)
    print(f"    tile #{k} of its workgroup: n {len(a)}  rows mean {a[:,0].mean():.0f} p95 {np.percentile(a[:,0],95):.0f}  lookback mean {a[:,1].mean():.0f}  total mean {a[:,2].mean():.0f}")
wgspan = np.array(lasts) - np.array(firsts)
print(f"  per-WG span (first tile start -> last tile end): mean {wgspan.mean():.0f} p50 {np.median(wgspan):.0f} p95 {np.percentile(wgspan,95):.0f} max {wgspan.max():.0f}")
for ntiles in sorted(set(counts)):
    sel = wgspan[np.array(counts) == ntiles]
    print(f"    WGs with {ntiles} tiles: {len(sel)}  span mean {sel.mean():.0f} max {sel.max():.0f}")
print(f"  workgroups {len(firsts)}  tiles/WG mean {np.mean(counts):.2f} min {min(counts)} max {max(counts)}")
print(f"  first tile start: mean {np.mean(firsts):.0f} max {np.max(firsts):.0f};  last tile end: mean {np.mean(lasts):.0f} min {np.min(lasts):.0f} max {np.max(lasts):.0f}")
if gaps:
    print(f"  gap between tiles of a WG: mean {np.mean(gaps):.0f} p95 {np.percentile(gaps, 95):.0f}")
