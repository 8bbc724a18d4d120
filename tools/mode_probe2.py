#!/usr/bin/env python3
"""Follow-up of tools/mode_probe.py: re-allocate ONE of {temp, archive rows, input, output} at a time (fresh torch
allocations, the others stay where they are) and watch the encoder's / decoder's duration."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dietgpu_amd as dg
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "bf16"
data, ft, _, P, desc = bench.make_workload(wl, 256, 1234, dev)
c = bench.Codec(dg, data, ft, P)
keep = []

def measure(tag):
    for _ in range(20):
        c.encode(); c.decode()
    torch.cuda.synchronize()
    prof = bench.kernel_profile(c, 80, lambda i: (c.encode(), c.decode()))
    t = {n[6:]: round(r["total_ms"] / max(r["launches"], 1) * 1e3, 1) for n, r in prof.items()}
    print("%-14s in %x comp %x out %x temp %x  %s" % (tag, c.data.data_ptr(), c.comp.data_ptr(), c.out.data_ptr(), c.temp.data_ptr(), t), flush=True)

for _ in range(3):
    measure("warm")
B = c.B
row_in = data.stride(0) * data.element_size()
for rnd in range(5):
    keep.append(c.temp); c.temp = torch.empty_like(c.temp); measure("new temp")
for rnd in range(5):
    keep.append(c.comp); c.comp = torch.empty_like(c.comp)
    c.comp_ptrs = (C.c_void_p * B)(*[c.comp.data_ptr() + i * c.row_cap for i in range(B)]); measure("new comp")
for rnd in range(5):
    keep.append(c.data); c.data = c.data.clone()
    c.in_ptrs = (C.c_void_p * B)(*[c.data.data_ptr() + i * row_in for i in range(B)]); measure("new in")
for rnd in range(5):
    keep.append(c.out); c.out = torch.empty_like(c.out)
    c.out_ptrs = (C.c_void_p * B)(*[c.out.data_ptr() + i * row_in for i in range(B)]); measure("new out")
