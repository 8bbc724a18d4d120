#!/usr/bin/env python3
"""Raw bytes (BASELINE config 2): does the histogram pass hide behind the encoder?  

The raw-byte encoder is compute-bound with HBM half idle (455 MB in ~142 us) and k_histogram reads its 268 MB at the
memory system's rate (~50 us): they contend for little.  Here the batch is cut into slices that go through the same
C-ABI call (dgpu_ans_encode_batch_pointer) on two streams, so that -- once the two streams are out of phase --
histogram(slice k+1) runs while encode(slice k) does.  `pattern` = rows per slice, slices alternate between the streams;
unequal first / last slices put the streams out of phase from the start.  Cold (rotating buffer sets) compress-only
time per batch, archives checked against the unsliced call.

  python tools/slice_streams_u8_experiment.py [--steps 60]
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

PATTERNS = {
    "unsliced": [256],
    "2 equal": [128, 128],
    "4 equal": [64] * 4,
    "8 equal": [32] * 8,
    "3 staggered (64,128,64)": [64, 128, 64],
    "5 staggered (32,64,64,64,32)": [32, 64, 64, 64, 32],
    "9 staggered (16,32x7,16)": [16] + [32] * 7 + [16],
    "5 staggered (16,64,64,64,48)": [16, 64, 64, 64, 48],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--sets", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = dg.lib()
    B, n, P = 256, 1 << 20, 10
    sets = []
    for r in range(args.sets):
        data, _, _, _, _ = bench.make_workload("u8", B, 1234 + 1000 * r, dev)
        sets.append(bench.Codec(dg, data, 0, P))
    for c in sets:
        c.step()
        c.verify()
    ref_sizes = [c.sizes.clone() for c in sets]
    ref_comp = sets[0].comp.clone()
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    tb = int(L.dgpu_ans_encode_temp_bytes(B, n))
    temps = [torch.empty((tb,), dtype=torch.uint8, device=dev) for _ in range(2)]
    main_stream = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event()
    joins = [torch.cuda.Event() for _ in range(2)]
    arg_cache = {}

    def slice_args(c, lo, per):
        key = (id(c), lo, per)
        if key not in arg_cache:
            arg_cache[key] = ((C.c_void_p * per)(*[c.data.data_ptr() + (lo + i) * n for i in range(per)]),
                              (C.c_uint32 * per)(*([n] * per)),
                              (C.c_void_p * per)(*[c.comp.data_ptr() + (lo + i) * c.row_cap for i in range(per)]),
                              C.c_void_p(c.sizes.data_ptr() + 4 * lo))
        return arg_cache[key]

    def compress(c, pattern):
        if len(pattern) == 1:
            c.encode()
            return
        fork.record(main_stream)
        for s in streams:
            s.wait_event(fork)
        lo = 0
        for j, per in enumerate(pattern):
            in_ptrs, sizes, out_ptrs, osz = slice_args(c, lo, per)
            k = j % 2
            rc = L.dgpu_ans_encode_batch_pointer(C.c_void_p(temps[k].data_ptr()), tb, None, P, 0, per, in_ptrs, sizes, None, out_ptrs, osz,
                                                 C.c_void_p(streams[k].cuda_stream))
            assert rc == 0, L.dgpu_last_error().decode()
            lo += per
        for k in range(2):
            joins[k].record(streams[k])
            main_stream.wait_event(joins[k])

    rows = []
    for name, pattern in PATTERNS.items():
        assert sum(pattern) == B
        for i in range(args.warmup):
            compress(sets[i % len(sets)], pattern)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            compress(sets[i % len(sets)], pattern)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / args.steps * 1e6
        for c, rs in zip(sets, ref_sizes):
            assert torch.equal(c.sizes, rs)
        assert torch.equal(sets[0].comp, ref_comp) or all(
            torch.equal(sets[0].comp[i, : int(ref_sizes[0][i])], ref_comp[i, : int(ref_sizes[0][i])]) for i in range(0, B, 37))
        rows.append({"pattern": name, "us_per_batch": round(us, 1)})
        print(rows[-1], flush=True)
    print(json.dumps({"experiment": "sliced two-stream compress, cold (rotating sets), 256 x 1 MiB Zipf bytes", "rows": rows}))


if __name__ == "__main__":
    main()
