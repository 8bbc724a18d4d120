#!/usr/bin/env python3
"""How long after the start of load does this box reach its steady rate?  The sysfs clock files of the GPU boxes report
the sleep state whatever runs (tools/clock_sampler.py --probe), so the clocks themselves cannot be logged by an ordinary
user; what can be logged is the duration of identical pieces of work over time since load began.  Two kinds of work,
each from an idle GPU (1 s without work), each piece between two events recorded back to back on the stream:

  * copy : a 512 MiB device-to-device copy (a vendor kernel, nothing of this library)
  * codec: one compress + decompress step of bench.py's headline loop (4 rotating buffer sets of BASELINE config 3)

Output: mean duration of the pieces by index bucket and by time since the first launch.  If the copy shows the same
transient as the codec step, the slow first milliseconds are the box's, not the library's.  Usage: python tools/transient_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import dietgpu_amd as dg

dev = torch.device("cuda:0")
BUCKETS = [(0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 400), (400, 800), (800, 1600)]


def run(label, fn, n, unit_bytes=None):
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]  # us
    t = [sum(d[:i]) / 1e3 for i in range(n)]                      # ms since the first launch
    last = sum(d[-n // 4:]) / (n // 4)
    print(f"== {label}: {n} pieces after 1 s idle; mean of the last quarter {last:.1f} us")
    print("   pieces        since start    mean us    vs last quarter")
    for lo, hi in BUCKETS:
        if lo >= n:
            break
        hi = min(hi, n)
        m = sum(d[lo:hi]) / (hi - lo)
        print(f"   {lo:5d}-{hi:<5d}  {t[lo]:8.2f} ms   {m:9.1f}    {m / last:6.3f}")
    return d


src = torch.empty(512 << 20, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty_like(src)
run("copy 512 MiB (vendor kernel)", lambda i: dst.copy_(src), 1600)
run("copy 512 MiB (vendor kernel), again", lambda i: dst.copy_(src), 1600)

sets = []
for r in range(4):
    d, ft, _, P, _ = bench.make_workload("bf16", 256, 1234 + 1000 * r, dev, 512 * 1024)
    c = bench.Codec(dg, d, ft, P)
    if sets:
        c.temp = sets[0].temp
    c.step()
    c.verify()
    sets.append(c)
run("codec step (rotating buffer sets, BASELINE config 3)", lambda i: sets[i % 4].step(), 1600)
run("codec step, again", lambda i: sets[i % 4].step(), 1600)
for c in sets:
    c.verify()
