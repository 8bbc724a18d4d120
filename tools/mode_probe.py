#!/usr/bin/env python3
"""The bf16 encoder runs at ~80 or ~86 us depending on the process (same build, same command).  This probe prints the
encode kernel's duration together with the device addresses of the buffers, then re-allocates everything a few times
inside the SAME process (with a pad allocation in between, so the addresses move) and measures again: a mode that
follows the addresses inside one process is a placement effect, one that only changes between processes is not."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dietgpu_amd as dg
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pads = []
for rnd in range(rounds):
    data, ft, _, P, desc = bench.make_workload(wl, 256, 1234, dev)
    c = bench.Codec(dg, data, ft, P)
    for _ in range(30):
        c.encode(); c.decode()
    torch.cuda.synchronize()
    prof = bench.kernel_profile(c, 100, lambda i: (c.encode(), c.decode()))
    t = {n[6:]: round(r["total_ms"] / max(r["launches"], 1) * 1e3, 1) for n, r in prof.items()}
    print("pid %d round %d  in %x comp %x out %x temp %x  %s" % (os.getpid(), rnd, data.data_ptr(), c.comp.data_ptr(),
          c.out.data_ptr(), c.temp.data_ptr(), t), flush=True)
    pads.append(torch.empty(((rnd + 1) * 3 * 1048576 + 4096 * (rnd + 1),), dtype=torch.uint8, device=dev))
    if rnd % 2 == 1:
        del c, data
        torch.cuda.empty_cache()
