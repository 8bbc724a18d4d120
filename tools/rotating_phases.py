#!/usr/bin/env python3
"""Compress-only and decompress-only loops over rotating buffer sets (the two sides of a compressed collective, a
checkpoint writer / reader): per-call time on one buffer set and on R distinct sets, per-kernel durations.
  python tools/rotating_phases.py [--workload bf16] [--sets 4] [--steps 300]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import dietgpu_amd as dg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="bf16")
ap.add_argument("--sets", type=int, default=4)
ap.add_argument("--steps", type=int, default=300)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sets = []
for r in range(args.sets):
    data, ft, _, P, desc = bench.make_workload(args.workload, 256, 1234 + 1000 * r, dev)
    c = bench.Codec(dg, data, ft, P)
    if sets:
        c.temp = sets[0].temp
    c.step()
    c.verify()
    sets.append(c)


def run(name, fn, nsets):
    for i in range(50):
        fn(sets[i % nsets])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fn(sets[i % nsets])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = bench.kernel_profile(sets[0], args.steps, lambda i: fn(sets[i % nsets]))
    k = {n[6:]: round(rec["total_ms"] / max(rec["launches"], 1) * 1e3, 1) for n, rec in prof.items()}
    print(f"{args.workload} {name:28s} sets {nsets}: {dt * 1e6:7.1f} us per call  kernels {k}")


for nsets in (1, args.sets):
    run("compress only", lambda c: c.encode(), nsets)
    run("decompress only", lambda c: c.decode(), nsets)
    run("compress + decompress", lambda c: c.step(), nsets)

# The two sides of a real exchange: the tensor to compress was just WRITTEN by a producer kernel (here a copy from a
# cold source: its lines are dirty in the memory-side cache), the archive to decompress just ARRIVED (a copy, as RCCL
# delivers it).  The copies are torch kernels: they do not appear in the library's per-kernel durations.
for c in sets:
    c.data_src = c.data.clone()
    c.comp_src = c.comp.clone()


def produce_then_compress(c):
    c.data.copy_(c.data_src)
    c.encode()


def receive_then_decompress(c):
    c.comp.copy_(c.comp_src)
    c.decode()


run("producer copy -> compress", produce_then_compress, args.sets)
run("arrival copy -> decompress", receive_then_decompress, args.sets)
