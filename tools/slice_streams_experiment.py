#!/usr/bin/env python3
"""Does a WINDOWED compress pipeline pay?  (tested with the kernels that ship.)

The encode call reads its input twice: k_float_histogram, then k_ans_encode.  With the whole 256 MiB batch between
the two reads the second one comes from HBM again.  Here the batch is cut into S slices that go through the same C-ABI
call on T streams round robin, so that histogram(slice k+1) runs WHILE encode(slice k) does and the encoder re-reads
bytes that were streamed in a few tens of microseconds earlier -- from the 256 MiB memory-side cache when the histogram
pass loads allocate there (dgpu_set_histogram_load_policy(1)).  Cold (rotating buffer sets) compress-only time per batch.

  python tools/slice_streams_experiment.py [--steps 100]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import dietgpu_amd as dg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sets", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = dg.lib()
    B, n, ft, P = 256, 512 * 1024, 2, 10
    sets = []
    for r in range(args.sets):
        data, _, _, _, _ = bench.make_workload("bf16", B, 1234 + 1000 * r, dev, n)
        sets.append(bench.Codec(dg, data, ft, P))
    for c in sets:
        c.step()
        c.verify()
    ref_sizes = [c.sizes.clone() for c in sets]
    ref_comp = [c.comp.clone() for c in sets[:1]]
    max_streams = 8
    streams = [torch.cuda.Stream(dev) for _ in range(max_streams)]
    tb = int(L.dgpu_float_compress_temp_bytes(ft, B, n))
    temps = [torch.empty((tb,), dtype=torch.uint8, device=dev) for _ in range(max_streams)]
    main_stream = torch.cuda.current_stream(dev)
    row_in = n * 2

    arg_cache = {}
    fork = torch.cuda.Event()
    joins = [torch.cuda.Event() for _ in range(max_streams)]

    def slice_args(c, S, j):
        key = (id(c), S, j)
        if key not in arg_cache:  # built once: the timed loop must not be bound by Python
            per = B // S
            lo = j * per
            arg_cache[key] = (per,
                              (C.c_void_p * per)(*[c.data.data_ptr() + (lo + i) * row_in for i in range(per)]),
                              (C.c_uint32 * per)(*([n] * per)),
                              (C.c_void_p * per)(*[c.comp.data_ptr() + (lo + i) * c.row_cap for i in range(per)]),
                              C.c_void_p(c.sizes.data_ptr() + 4 * lo))
        return arg_cache[key]

    def compress_sliced(c, S, T):
        if S == 1 and T == 1:
            c.encode()
            return
        fork.record(main_stream)
        for k in range(min(S, T)):
            streams[k].wait_event(fork)
        for j in range(S):
            per, in_ptrs, sizes, out_ptrs, osz = slice_args(c, S, j)
            rc = L.dgpu_float_compress(C.c_void_p(temps[j % T].data_ptr()), tb, None, ft, P, 0, per, in_ptrs, sizes, out_ptrs, osz,
                                       C.c_void_p(streams[j % T].cuda_stream))
            assert rc == 0, L.dgpu_last_error().decode()
        for k in range(min(S, T)):
            joins[k].record(streams[k])
            main_stream.wait_event(joins[k])

    def timed(S, T):
        for i in range(args.warmup):
            compress_sliced(sets[i % len(sets)], S, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            compress_sliced(sets[i % len(sets)], S, T)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        # archives unchanged
        for c, rs in zip(sets, ref_sizes):
            assert torch.equal(c.sizes, rs)
        k = int(ref_sizes[0][0])
        assert torch.equal(sets[0].comp[0, :k], ref_comp[0][0, :k])
        return dt * 1e6

    rows = []
    for policy in (0, 1):
        L.dgpu_set_histogram_load_policy(policy)
        for S, T in ((1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (16, 2), (16, 4), (16, 8)):
            us = timed(S, T)
            rows.append({"hist_loads": "allocating" if policy else "non-temporal", "slices": S, "streams": T, "us_per_batch": round(us, 1)})
            print(rows[-1], flush=True)
    L.dgpu_set_histogram_load_policy(-1)
    print(json.dumps({"experiment": "sliced multi-stream compress, cold (rotating sets), 256 x 512 Ki bf16", "rows": rows}))


if __name__ == "__main__":
    main()
