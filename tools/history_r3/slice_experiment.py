#!/usr/bin/env python3
"""Experiment: does encoding the batch in slices of <= 128 MiB (histogram slice k, encode slice k) let the
encoder's read hit the 256 MiB memory-side cache?  Times compress of 256 x 1 MiB bf16 as 1, 2, 4 calls."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dietgpu_amd as dg
import bench

dev = torch.device("cuda:0")
data, ft, _, P, _ = bench.make_workload("bf16", 256, 1234, dev)
for slices in (1, 2, 4, 8):
    per = 256 // slices
    codecs = [bench.Codec(dg, data[i * per:(i + 1) * per], ft, P) for i in range(slices)]
    dec = bench.Codec(dg, data, ft, P)
    def enc():
        for c in codecs:
            c.encode()
    for _ in range(5):
        enc(); dec.encode(); dec.decode()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(20):
        dec.decode()          # the step's other half, so the cache state is the bench's
        s.record(); enc(); e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    print(f"{os.environ.get('DGPU_LIB','default').split('/')[-1]:24s} slices {slices}: encode {tot / 20 * 1e3:7.1f} us")
