#!/usr/bin/env python3
"""Where the time of ONE compress / decompress call of a 2-128 MiB tensor goes (the reference's use case: gradient and
activation tensors, README.md:68-72; its benchmark protocol: a batch of ONE tensor, benchmark.py:35-86).

    python tools/small_call_probe.py [--sizes 1,4,16,64] [--reps 200]          events around each op, no profiler
    rocprofv3 --kernel-trace -d DIR -o sc -- python tools/small_call_probe.py --trace HOST.jsonl
    python tools/small_call_probe.py --merge DIR/sc_results.db HOST.jsonl       the timeline of a call

Without a profiler: per size, compress and decompress through torch.ops.dietgpu.* with temp_mem -- median / min of `reps`
calls timed one by one with events (a synchronise between them: the reference protocol), and the rate of `reps` calls
enqueued back to back.  With --trace the calls are made one by one with the host's clocks written around them; --merge
joins that with the profiler's kernel trace: call -> first kernel start, kernel durations, gaps, last kernel end ->
the host sees the call complete."""
import argparse
import json
import sqlite3
import statistics
import sys
import time

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def clocks():
    return [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]


BATCH = 1  # --batch: tensors per call (the rows of one [BATCH, n] tensor)


def setup(n, dt):
    import torch

    import dietgpu_amd

    ops = dietgpu_amd.load_torch_ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1234)
    t = torch.randn([BATCH, n], generator=g, device=dev, dtype=torch.float32).to(dt)
    ts = list(t.unbind(0))
    r, c = ops.max_float_compressed_output_size(ts)
    comp = torch.empty([r, c], dtype=torch.uint8, device=dev)
    sizes = torch.zeros([BATCH], dtype=torch.int32, device=dev)
    out = torch.empty_like(t)
    status = torch.empty([BATCH], dtype=torch.uint8, device=dev)
    osz = torch.empty([BATCH], dtype=torch.int32, device=dev)
    return ops, ts, comp, sizes, list(out.unbind(0)), status, osz


def measure(a):
    import torch

    temp = torch.empty([384 * 1024 * 1024], dtype=torch.uint8, device="cuda:0")
    rows = []
    for m in [float(x) for x in a.sizes.split(",")]:
        n = int(m * 1024 * 1024)
        ops, t, comp, sizes, out, status, osz = setup(n, torch.bfloat16)
        enc = lambda: ops.compress_data(True, t, False, temp, comp, sizes)
        enc()
        torch.cuda.synchronize()
        comp_ts = [comp[i, : int(s)] for i, s in enumerate(sizes.tolist())]
        dec = lambda: ops.decompress_data(True, comp_ts, out, False, temp, status, osz)
        dec()
        torch.cuda.synchronize()
        assert all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(t, out))
        row = {"mega_floats": m, "batch": BATCH, "bytes": n * 2 * BATCH}
        n = n * BATCH
        for name, fn in (("compress", enc), ("decompress", dec)):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            one = []
            for _ in range(a.reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fn()
                e.record()
                torch.cuda.synchronize()
                one.append(s.elapsed_time(e) * 1e3)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.record()
            for _ in range(a.reps):
                fn()
            t_enq = time.perf_counter() - t0
            e.record()
            torch.cuda.synchronize()
            b2b = s.elapsed_time(e) * 1e3 / a.reps
            row[name] = {"one_by_one_median_us": round(statistics.median(one), 2), "one_by_one_min_us": round(min(one), 2),
                         "back_to_back_us": round(b2b, 2), "host_enqueue_us": round(t_enq / a.reps * 1e6, 2),
                         "GBps_one_by_one": round(n * 2 / statistics.median(one) / 1e3, 1), "GBps_back_to_back": round(n * 2 / b2b / 1e3, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    return rows


def trace(a):
    import torch

    temp = torch.empty([384 * 1024 * 1024], dtype=torch.uint8, device="cuda:0")
    log = open(a.trace, "w")
    for m in [float(x) for x in a.sizes.split(",")]:
        n = int(m * 1024 * 1024)
        ops, t, comp, sizes, out, status, osz = setup(n, torch.bfloat16)
        enc = lambda: ops.compress_data(True, t, False, temp, comp, sizes)
        enc()
        torch.cuda.synchronize()
        comp_ts = [comp[i, : int(s)] for i, s in enumerate(sizes.tolist())]
        dec = lambda: ops.decompress_data(True, comp_ts, out, False, temp, status, osz)
        for name, fn in (("compress", enc), ("decompress", dec)):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            for i in range(a.trace_reps):
                c0 = clocks()
                fn()
                c1 = clocks()
                torch.cuda.synchronize()
                c2 = clocks()
                log.write(json.dumps({"mega_floats": m, "op": name, "i": i, "call": c0, "returned": c1, "synced": c2}) + "\n")
                time.sleep(0.0005)
    log.close()


def merge(db, host):
    con = sqlite3.connect(db)
    ks = [(s, e, n) for n, s, e in con.execute("select name, start, end from kernels order by start") if "dgpu::" in n]
    calls = [json.loads(line) for line in open(host)]
    col = None
    for c in range(3):
        if calls[0]["call"][c] <= ks[-1][0] and calls[-1]["synced"][c] >= ks[0][0] and abs(calls[0]["call"][c] - ks[0][0]) < 600e9:
            col = c
    print(f"# profiler time base: {['CLOCK_MONOTONIC', 'CLOCK_BOOTTIME', 'CLOCK_REALTIME'][col] if col is not None else 'NOT MATCHED'}")
    if col is None:
        return
    import bisect
    import re

    starts = [k[0] for k in ks]
    groups = {}
    for c in calls:
        lo, hi = c["call"][col], c["synced"][col]
        i, j = bisect.bisect_left(starts, lo), bisect.bisect_right(starts, hi)
        mine = ks[i:j]
        if not mine:
            continue
        rec = {"call_to_first_kernel": (mine[0][0] - lo) / 1e3, "host_in_call": (c["returned"][col] - lo) / 1e3,
               "last_kernel_to_sync_seen": (hi - mine[-1][1]) / 1e3, "call_to_sync_seen": (hi - lo) / 1e3,
               "gpu_span": (mine[-1][1] - mine[0][0]) / 1e3}
        for q, k in enumerate(mine):
            nm = re.sub(r"<.*", "", re.sub(r"\(.*", "", k[2]).replace("void ", "").replace("dgpu::", ""))
            rec[f"k{q}_{nm}"] = (k[1] - k[0]) / 1e3
            if q:
                rec[f"gap{q - 1}_{q}"] = (k[0] - mine[q - 1][1]) / 1e3
        groups.setdefault((c["mega_floats"], c["op"]), []).append(rec)
    for (m, op), recs in groups.items():
        print(f"\n== {m:g} Mi bf16, {op}: median over {len(recs)} calls made one by one (us)")
        keys = list(recs[0].keys())
        for k in keys:
            v = [r[k] for r in recs if k in r]
            print(f"   {k:34s} {statistics.median(v):9.2f}   (min {min(v):.2f}, max {max(v):.2f})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,4,16,64")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--trace", default=None)
    ap.add_argument("--trace-reps", type=int, default=40)
    ap.add_argument("--merge", nargs=2, default=None)
    ap.add_argument("--batch", type=int, default=1, help="tensors per call (each of --sizes Mi words)")
    a = ap.parse_args()
    BATCH = a.batch
    if a.merge:
        merge(*a.merge)
    elif a.trace:
        trace(a)
    else:
        measure(a)
