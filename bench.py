#!/usr/bin/env python3
"""Headline benchmark: batched rANS float codec, encode + decode, on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bf16|u8|fp16|fp32]

`--gpus N` with N > 1 and no torchrun environment starts the N ranks itself
(torch.distributed.run, one process per GPU, RCCL); under the driver's own
torchrun it checks WORLD_SIZE == N.  It refuses to run on fewer GPUs than asked.

Workload (BASELINE.json `metric`: "rANS encode+decode GB/s on 256x1 MiB bf16"):
256 tensors x 524288 bfloat16 ~ N(0,1) per GPU, probBits
10, through the C ABI (dgpu_float_compress / dgpu_float_decompress), inputs and
outputs resident in HBM, temp memory pre-supplied (no allocation in the timed
region).  One step = one compress pass + one decompress pass over the batch.

THE HEADLINE IS THE CACHE-COLD STEP.  `ms_per_step` / `value` / `roofline` come from
exactly W warm-up + K timed steps over `--rotate` (4) distinct {input, archive, output}
buffer sets of different data coded round robin -- 2.9 GB touched per rotation, so
neither an input nor an archive nor an output line can still be in the 256 MiB
memory-side cache when its turn comes: every byte the step moves comes from / goes to
HBM.  A loop that re-codes ONE buffer set (what rounds 1-3 reported, and what the
reference's own benchmark does) is 15-20 % faster because half its traffic is served
by that cache; it is printed as a named extra (`ms_per_step_one_buffer_set*`), as are
the steady-state figures after a pre-roll of >= 400 steps (`*_steady_state`: short
regions right after an idle fence include the clock ramp).

value = uncompressed input bytes moved per second over all ranks:
        N_gpus * 2 * batch_bytes / step_time   (same definition as the
        reference's benchmark.py:156-157, summed over encode and decode).

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank codes
its own 256-tensor shard (weak scaling, no data-path collective; the codec has
no exchange step); timing is barrier + synchronize on both sides, MAX over
ranks.  After the timed region the ranks all-gather their compressed sizes
(RCCL) only to report the aggregate ratio.

The JSON line also carries
  roofline     -- the dominant kernel OF THE HEADLINE LOOP: its algorithmic bytes /
                  its mean duration, durations measured with HIP events on the
                  launch stream in a second, instrumented pass of the same loop;
  roofline_by_direction -- compress = algorithmic bytes / (histogram + encode),
                  decompress = algorithmic bytes / decode, cold (rotating sets) and
                  warm (one buffer set), and each direction running alone;
  cpu_baseline -- the CPU oracle (oracle/, kind "port": the reference has no
                  CPU path) timed on this host's cores on a bounded sample.
Every run ends with a bit-exact round-trip check.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6290 measured copy


def make_workload(kind, batch, seed, device, n=512 * 1024):
    """Returns (tensors-as-2D tensor, float type or 0, element bytes, prob_bits, description)."""
    if kind == "bf16":
        g = torch.Generator(device=device).manual_seed(seed)
        t = torch.randn((batch, n), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        return t, 2, 2, 10, f"{batch}x{n} bfloat16 N(0,1), float codec, probBits 10 (BASELINE config 3)"
    if kind == "fp16":
        g = torch.Generator(device=device).manual_seed(seed)
        t = torch.randn((batch, n), generator=g, device=device, dtype=torch.float32)
        mask = torch.rand((batch, n), generator=g, device=device) < 0.5
        t = t.masked_fill(mask, 0.0).to(torch.float16)
        return t, 1, 2, 11, f"{batch}x{n} float16 N(0,1) 50% zeros, float codec, probBits 11 (BASELINE config 4)"
    if kind == "fp32":
        g = torch.Generator(device=device).manual_seed(seed)
        t = torch.randn((batch, n // 2), generator=g, device=device, dtype=torch.float32)
        return t, 3, 4, 10, f"{batch}x{n // 2} float32 N(0,1), float codec, probBits 10 (not a BASELINE config)"
    if kind == "u8":
        import refgen

        # every row is its own stream, default_rng(1234 + b)
        rows = refgen.zipf_bytes(batch, 1 << 20, seed=seed)
        t = torch.from_numpy(rows).to(device)
        return t, 0, 1, 10, f"{batch}x1MiB uint8 Zipf(1.2) (independent rows, default_rng(1234+b)), raw rANS, probBits 10 (BASELINE config 2)"
    raise ValueError(kind)


class Codec:
    """Pre-built C-ABI argument arrays so the timed loop is two library calls per step."""

    def __init__(self, dg, data, ft, prob_bits):
        self.dg, self.lib = dg, dg.lib()
        self.ft, self.P = ft, prob_bits
        self.data = data
        B = data.shape[0]
        self.B = B
        dev = data.device
        self.elems = data.shape[1]
        self.in_bytes = data.numel() * data.element_size()
        size_units = self.elems if ft else self.elems * data.element_size()
        if ft:
            self.row_cap = int(self.lib.dgpu_float_max_compressed_size(ft, size_units))
            tmp = max(int(self.lib.dgpu_float_compress_temp_bytes(ft, B, size_units)),
                      int(self.lib.dgpu_float_decompress_temp_bytes(ft, B, size_units, prob_bits)))
        else:
            self.row_cap = int(self.lib.dgpu_ans_max_compressed_size(size_units))
            tmp = max(int(self.lib.dgpu_ans_encode_temp_bytes(B, size_units)),
                      int(self.lib.dgpu_ans_decode_temp_bytes(B, size_units, prob_bits)))
        self.comp = torch.empty((B, self.row_cap), dtype=torch.uint8, device=dev)
        self.out = torch.empty_like(data)
        self.sizes = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.status = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self.osz = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.temp = torch.empty((tmp,), dtype=torch.uint8, device=dev)
        row_in = data.stride(0) * data.element_size()
        self.in_ptrs = (C.c_void_p * B)(*[data.data_ptr() + i * row_in for i in range(B)])
        self.comp_ptrs = (C.c_void_p * B)(*[self.comp.data_ptr() + i * self.row_cap for i in range(B)])
        self.out_ptrs = (C.c_void_p * B)(*[self.out.data_ptr() + i * row_in for i in range(B)])
        self.in_sizes = (C.c_uint32 * B)(*([size_units] * B))
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.err = C.c_int32(-1)

    def encode(self):
        L = self.lib
        tp, tb = C.c_void_p(self.temp.data_ptr()), self.temp.numel()
        if self.ft:
            rc = L.dgpu_float_compress(tp, tb, None, self.ft, self.P, 0, self.B, self.in_ptrs, self.in_sizes,
                                       self.comp_ptrs, C.c_void_p(self.sizes.data_ptr()), self.stream)
        else:
            rc = L.dgpu_ans_encode_batch_pointer(tp, tb, None, self.P, 0, self.B, self.in_ptrs, self.in_sizes, None,
                                                 self.comp_ptrs, C.c_void_p(self.sizes.data_ptr()), self.stream)
        if rc:
            raise RuntimeError(L.dgpu_last_error().decode())

    def decode(self):
        L = self.lib
        tp, tb = C.c_void_p(self.temp.data_ptr()), self.temp.numel()
        if self.ft:
            rc = L.dgpu_float_decompress(tp, tb, None, self.ft, self.P, 0, self.B, self.comp_ptrs, self.out_ptrs,
                                         self.in_sizes, C.c_void_p(self.status.data_ptr()),
                                         C.c_void_p(self.osz.data_ptr()), self.stream, C.byref(self.err))
        else:
            rc = L.dgpu_ans_decode_batch_pointer(tp, tb, None, self.P, 0, self.B, self.comp_ptrs, self.out_ptrs,
                                                 self.in_sizes, C.c_void_p(self.status.data_ptr()),
                                                 C.c_void_p(self.osz.data_ptr()), self.stream, C.byref(self.err))
        if rc:
            raise RuntimeError(L.dgpu_last_error().decode())

    def step(self):
        self.encode()
        self.decode()

    def scatter_pointers(self):
        """Swaps elements 0 and 1 in every pointer list: the batch is the same work, but its addresses no longer form
        an arithmetic progression, so the library cannot treat it as a stride batch and needs its parameter block
        (the device copy of the pointer / size arrays) -- the general pointer-list case.  Calling it again undoes it."""
        if self.B < 2:
            return
        for a in (self.in_ptrs, self.comp_ptrs, self.out_ptrs):
            a[0], a[1] = a[1], a[0]

    def verify(self):
        torch.cuda.synchronize()
        if ablation_run():
            return  # timing ablations of tools/ab.sh variants that produce WRONG archives by design; never a measurement
        assert bool(self.status.all().item()), "decode reported failure"
        a = self.data.view(torch.uint8)
        b = self.out.view(torch.uint8)
        assert torch.equal(a, b), "round trip is not bit-exact"


def ablation_run():
    """DGPU_BENCH_ABLATION=1 (tools/ab.sh timing ablations: WRONG archives by design) turns Codec.verify() off: such a
    run's line must not claim a round trip it never checked."""
    return os.environ.get("DGPU_BENCH_ABLATION") == "1"


def kernel_profile(codec, steps, step_fn=None):
    """Second pass of the same steps with per-kernel HIP events on the launch stream."""
    L = codec.lib
    L.dgpu_prof_reset()
    L.dgpu_prof_enable(1)
    for i in range(steps):
        if step_fn:
            step_fn(i)
        else:
            codec.step()
    torch.cuda.synchronize()
    L.dgpu_prof_enable(0)
    buf = C.create_string_buffer(1 << 16)
    n = L.dgpu_prof_summary(buf, len(buf))
    L.dgpu_prof_reset()
    return json.loads(buf.value.decode()) if n > 0 else {}


def algorithmic_bytes(kernel, codec, comp_total):
    """Algorithmic HBM bytes one launch of `kernel` must move (DESIGN.md section 5,
    SURVEY.md section 8d): every input byte read once, every output byte written once."""
    E = codec.B * codec.elems  # elements (float codec) or bytes (raw)
    if codec.ft:
        wb = 2 if codec.ft in (1, 2) else 4
        nc = E * (wb - 1)                      # non-compressed plane bytes
        ans = comp_total - 16 * codec.B - nc   # compressed exponent archives
        return {
            "k_float_histogram": E * wb,            # read the float words once
            "k_stats_single": E * wb,               # (batches of single-block elements: one wavefront per element)
            "k_ans_encode": E * wb + nc + ans,      # read words, write non-comp plane + rANS archive (split fused in)
            "k_ans_decode": ans + nc + E * wb,      # read archive + non-comp plane, write words (join fused in)
            "k_ans_encode_pair": E * wb + nc + ans,  # (single-block elements, two per wavefront: same bytes)
            "k_ans_decode_pair": ans + nc + E * wb,
        }.get(kernel)
    return {
        "k_histogram": E,
        "k_stats_single": E,
        "k_ans_encode": E + comp_total,
        "k_ans_decode": comp_total + E,
        "k_ans_encode_pair": E + comp_total,
        "k_ans_decode_pair": comp_total + E,
    }.get(kernel)


def compact_config_line(dg, kind, device, steps, warmup, rotate=4, batch=256, elems=512 * 1024):
    """One more BASELINE configuration on the same protocol as the headline (cache-cold loop over `rotate` buffer sets,
    exactly `warmup` + `steps` steps, bit-exact check of every set, per-kernel HIP events in a second pass), reduced
    to what a reader needs to judge it: step time, fraction of the HBM peak, the dominant kernel and its fraction."""
    data, ft, _, prob_bits, desc = make_workload(kind, batch, 1234, device, elems)
    sets = [Codec(dg, data, ft, prob_bits)]
    for r in range(1, rotate):
        d2, _, _, _, _ = make_workload(kind, batch, 1234 + 1000 * r, device, elems)
        c2 = Codec(dg, d2, ft, prob_bits)
        c2.temp = sets[0].temp
        sets.append(c2)
    for c in sets:
        c.step()
        c.verify()
    step = lambda i: sets[i % len(sets)].step()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    for c in sets:
        c.verify()
    codec = sets[0]
    comp_total = int(codec.sizes.to(torch.int64).sum().item())
    prof = kernel_profile(codec, max(steps, 50), step)
    E = codec.B * codec.elems
    wb = (2 if ft in (1, 2) else 4) if ft else 1
    step_alg = 2 * (E * wb + comp_total)
    table = {}
    for name, rec in prof.items():
        ab = algorithmic_bytes(name, codec, comp_total)
        us = rec["total_ms"] / max(rec["launches"], 1) * 1e3
        table[name] = {"avg_us": round(us, 2), "frac": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if ab else None}
    hot = [k for k in table if table[k]["frac"] is not None]
    dom = max(hot, key=lambda k: table[k]["avg_us"]) if hot else None
    out = {
        "workload": desc, "steps": steps, "warmup": warmup, "rotating_sets": len(sets),
        "ms_per_step": round(sec * 1e3, 4),
        "value_GBps": round(2 * codec.in_bytes / sec / 1e9, 2),
        "step_frac_of_hbm_peak": round(step_alg / sec / 1e9 / HBM_PEAK_GBPS, 4),
        "dominant_kernel": dom, "dominant_kernel_frac": table[dom]["frac"] if dom else None,
        "kernels_us": {k[2:]: v["avg_us"] for k, v in table.items()},
        "compression_ratio": round(comp_total / codec.in_bytes, 4),
        "round_trip_bit_exact": None if ablation_run() else True,
        **({"ablation": True} if ablation_run() else {}),
    }
    del sets, codec, data
    torch.cuda.empty_cache()
    return out


def baseline_metric():
    """BASELINE.json's wording of the metric this line reports (None if the file is not there)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f).get("metric")
    except (OSError, ValueError):
        return None


def kernel_source_hash():
    """sha256 over the sources the library is built from (dietgpu_amd/csrc/*, include/dietgpu_amd.h), the stamp
    tools/make_traffic_json.py puts into profiles/*_hbm_traffic.json: ties the PMC passes to the build that ran
    (the GPU box has no .git, so this is a content hash, not a commit id)."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "dietgpu_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip", ".cpp")))
    files.append(os.path.join(ROOT, "include", "dietgpu_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


TRAFFIC_FILES = ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json")


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (cannot be collected inside this process:
    rocprofv3 wraps the command) -- ONLY when those passes were taken on the sources this build was made from
    (`kernel_source_hash` stamped into the file by tools/make_traffic_json.py).  Returns (bytes or None, note)."""
    want = kernel_source_hash()
    seen = []
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                doc = json.load(f)
        except (OSError, ValueError):
            continue
        stamp = doc.get("kernel_source_hash")
        seen.append(f"{name}: {stamp}")
        if stamp != want:
            continue
        try:
            return doc["workloads"][workload][kernel].get("hbm_bytes_per_launch"), f"profiles/{name} (kernel_source_hash {stamp} = this build)"
        except KeyError:
            continue
    return None, f"no PMC pass for this build (kernel_source_hash {want}; on file: {', '.join(seen) or 'none'})"


def cpu_baseline(kind, prob_bits, budget_s=12.0):
    """Times the CPU oracle (restatement of the reference algorithm; the reference itself has no CPU path) on this
    host: persistent threads, one per core and pinned, every buffer allocated and touched before the clock starts
    (oracle/dietgpu_oracle.c: dgo_bench_roundtrip), two rows of the bench workload per thread.  Round 3's figure
    (7.9 GB/s on 256 threads = 12 x one thread) timed thread creation, first-touch of fresh output arrays and
    mmap/munmap of every per-row temporary, not the codec."""
    import oracle as O
    import refgen

    L = O.lib()
    L.dgo_bench_roundtrip.restype = C.c_int
    L.dgo_bench_roundtrip.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    # What the host really grants: a container may SEE every CPU of the machine and still be held to a CFS quota
    # (the GPU boxes of this project: 256 visible CPUs, cgroup cpu.max = 16 CPUs' worth -- more runnable threads than
    # that are throttled, profiles/r04_cpu_scaling_probe.txt).  One thread per granted CPU.
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            quota = None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                quota = q / period if q > 0 else None
        except (OSError, ValueError):
            pass
    cores = max(1, min(visible, int(quota))) if quota else visible
    # under a quota the threads are left to the scheduler: pinning 16 threads to the first 16 of 256 shared CPUs put them
    # where other tenants already ran (9.5 GB/s at 5.6 CPU-seconds per wall-second against 11.5 GB/s at 15.2 unpinned)
    pin = 0 if (quota and quota < visible) else 1
    rows = max(2 * cores, 16)
    n = 512 * 1024
    if kind == "u8":
        data, ft, size = refgen.zipf_bytes(rows, 1 << 20), 0, 1 << 20  # rows of the bench workload: default_rng(1234 + b)
    else:
        ft = {"bf16": O.BFLOAT16, "fp16": O.FLOAT16, "fp32": O.FLOAT32}[kind]
        if kind == "bf16":
            data = refgen.normal_bf16(rows, n)
        elif kind == "fp16":
            data = refgen.sparse_fp16(rows, n)
        else:
            n = n // 2
            data = np.random.default_rng(1234).standard_normal((rows, n), dtype=np.float32).view(np.uint32)
        size = n
    data = np.ascontiguousarray(data)
    row_bytes = data.shape[1] * data.itemsize

    def run(nrows, threads, budget):
        def call(reps):
            e, d, bad = C.c_double(0), C.c_double(0), C.c_int(-1)
            rc = L.dgo_bench_roundtrip(ft, data.ctypes.data, size, row_bytes, nrows, prob_bits, threads, reps, pin,
                                       C.byref(e), C.byref(d), C.byref(bad))
            assert rc == 0 and bad.value == 0, "CPU oracle round trip failed"
            return e.value, d.value
        e1, d1 = call(1)  # one rep to size the sample
        reps = int(max(1, min(50, budget / max(e1 + d1, 1e-6))))
        e, d = call(reps)
        nbytes = nrows * row_bytes * reps
        return 2 * nbytes / (e + d) / 1e9, nbytes / e / 1e9, nbytes / d / 1e9, reps

    import resource

    r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    allc, allc_enc, allc_dec, reps = run(rows, cores, budget_s)
    r1, t1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    granted = ((r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)) / max(t1 - t0, 1e-9)
    # SURVEY.md section 8(d) also asks for the single-thread figure: 4 rows on one (pinned) thread
    one, one_enc, one_dec, reps1 = run(4, 1, budget_s / 3)
    return {
        "value": round(allc, 4),
        "unit": "GB/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{rows} rows of the same workload x {reps} reps, oracle/ C restatement, {cores} persistent "
                  f"{'pinned ' if pin else ''}pthreads (rows b, b + {cores}, ... per thread), buffers pre-touched; encode {allc_enc:.3f} GB/s, "
                  f"decode {allc_dec:.3f} GB/s",
        "speedup_over_one_thread": round(allc / one, 1) if one else None,
        "host": {"visible_cpus": visible, "cgroup_cpu_quota": quota, "cpu_seconds_per_wall_second": round(granted, 1),
                 "note": "threads = CPUs the cgroup grants (quota), not the CPUs it can see" if quota and quota < visible else None},
        "single_thread": {"value": round(one, 4), "unit": "GB/s", "cores": 1,
                          "sample": f"4 rows x {reps1} reps on one pinned thread; encode {one_enc:.3f} GB/s, "
                                    f"decode {one_dec:.3f} GB/s"},
    }


def run_collective(args, data, ft, desc, world, rank, device, D):
    """Plain vs compressed all-gather of every rank's shard (the use the reference's README motivates,
    README.md:68-72,104).  One "step" = every rank ends up with every other rank's tensors, bit-exact."""
    import torch.distributed as dist

    if not ft:
        sys.exit("bench.py --collective: float workloads only (bf16 / fp16 / fp32)")
    if not dist.is_initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        D.init(backend=args.dist_backend, device=device)
    raw_bytes = data.numel() * data.element_size()

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def plain():
        out = [torch.empty_like(data) for _ in range(world)]
        dist.all_gather(out, data)
        return out

    if not args.chunks:
        args.chunks = 1 if world == 1 else 4
    plan = D.CompressedAllGatherPlan(data, chunks=args.chunks)

    def compressed():
        out, redo = plan.run(data)
        return out, dict(plan.last)

    def timed(fn, finish=None):
        for _ in range(args.warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = fn()
        if finish:
            res = finish()
        fence()
        return D.max_over_ranks(time.perf_counter() - t0, device) / args.steps, res

    # the same exchange with the host read of step k behind the launch of step k + 1 (a plan of depth 2: two buffer
    # sets): a steady-state step is launch-only; the result of a step is complete when its handle has been waited for
    plan2 = D.CompressedAllGatherPlan(data, chunks=args.chunks, depth=2)
    prev = [None]

    def compressed_pipelined():
        h = plan2.run_async(data)
        if prev[0] is not None:
            prev[0].wait()
        prev[0] = h

    def finish_pipelined():
        out, redo = prev[0].wait()
        prev[0] = None
        return out

    t_plain, got_plain = timed(plain)
    t_comp, (got_comp, stats) = timed(compressed)
    t_pipe, got_pipe = timed(compressed_pipelined, finish_pipelined)
    view = torch.int32 if ft == 3 else torch.int16
    for r in range(world):
        assert torch.equal(got_comp[r].view(view), got_plain[r].view(view)), "compressed all-gather is not bit-exact"
        assert torch.equal(got_pipe[r].view(view), got_plain[r].view(view)), "pipelined compressed all-gather is not bit-exact"
    if rank == 0:
        recv = (world - 1) * raw_bytes if world > 1 else raw_bytes  # bytes of other ranks' tensors each rank ends up with
        print(json.dumps({
            "metric": "compressed_all_gather_effective_GBps_per_rank",
            "value": round(recv / t_comp / 1e9, 2),
            "unit": "GB/s",
            "plain_all_gather_GBps_per_rank": round(recv / t_plain / 1e9, 2),
            "speedup_vs_plain": round(t_plain / t_comp, 3),
            "ms_compressed": round(t_comp * 1e3, 4),
            "ms_compressed_pipelined": round(t_pipe * 1e3, 4),  # host read of step k behind the launch of step k + 1
            "ms_plain": round(t_plain * 1e3, 4),
            "n_gpus": world,
            "dist_backend": args.dist_backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "config": {"workload": desc, "per_rank_bytes": raw_bytes, "chunks": args.chunks,
                       "wire_bytes_per_rank": stats["wire_bytes"], "row_width_bytes": stats["width"],
                       "largest_archive_bytes": stats["largest_archive"],
                       "rows_sent_uncompressed": stats["rows_sent_uncompressed"]},
            "note": "world 1: the exchange is a local copy; the line then only shows the codec cost of the pipeline"
                    if world == 1 else "ring all-gather over xGMI is bound by one link per hop",
            "bit_exact": True,
        }), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# A100 figures read off the reference's README plots (batch of one tensor, +-5 GB/s)
A100_README = {
    "bfloat16": {"compress": {1: 20, 16: 180, 128: 302, 1024: 345}, "decompress": {1: 23, 16: 257, 128: 488, 1024: 607}},
    "float16": {"compress": {1: 20, 16: 148, 128: 261, 1024: 305}, "decompress": {1: 20, 16: 221, 128: 384, 1024: 489}},
}


def run_reference_protocol(args, device):
    """The reference's own benchmark (dietgpu/benchmark.py:35-86,156-175; README.md:94 plots): ONE tensor of
    1 M / 16 M / 128 M / 1024 M floats ~ N(0,1), bf16 and fp16, through torch.ops.dietgpu.compress_data /
    decompress_data with a 384 MiB temp_mem, first run discarded, mean of `runs`, compress and decompress timed
    separately with events around the op, every run validated bit for bit.  GB/s = uncompressed bytes / time."""
    import dietgpu_amd

    ops = dietgpu_amd.load_torch_ops()
    temp = torch.empty([384 * 1024 * 1024], dtype=torch.uint8, device=device)
    rows_out = []
    sizes_m = [int(x) for x in args.ref_sizes.split(",")]
    runs = 3
    for dt, name in ((torch.bfloat16, "bfloat16"), (torch.float16, "float16")):
        for m in sizes_m:
            n = m * 1024 * 1024
            g = torch.Generator(device=device).manual_seed(1234 + m)
            # (torch.normal in 64 M-element pieces: a 1024 M-element fp32 intermediate would be 4 GiB more)
            t = torch.empty([n], dtype=dt, device=device)
            piece = 64 * 1024 * 1024
            for lo in range(0, n, piece):
                hi = min(n, lo + piece)
                t[lo:hi] = torch.randn([hi - lo], generator=g, device=device, dtype=torch.float32).to(dt)
            ts = [t]
            r, c = ops.max_float_compressed_output_size(ts)
            comp = torch.empty([r, c], dtype=torch.uint8, device=device)
            sizes = torch.zeros([1], dtype=torch.int32, device=device)
            out = torch.empty_like(t)
            status = torch.empty([1], dtype=torch.uint8, device=device)
            osz = torch.empty([1], dtype=torch.int32, device=device)
            ct = dct = 0.0
            for i in range(1 + runs):
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                comp, sizes, _ = ops.compress_data(True, ts, False, temp, comp, sizes)
                e0.record()
                torch.cuda.synchronize()
                if i > 0:
                    ct += s0.elapsed_time(e0)
                comp_ts = [*comp]
                s1, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s1.record()
                ops.decompress_data(True, comp_ts, [out], False, temp, status, osz)
                e1.record()
                torch.cuda.synchronize()
                if i > 0:
                    dct += s1.elapsed_time(e1)
                assert bool(status.all().item()) and int(osz[0].item()) == n
                assert torch.equal(t.view(torch.int16), out.view(torch.int16)), "round trip is not bit-exact"
            ct, dct = ct / runs, dct / runs
            nbytes = n * 2
            row = {"dtype": name, "mega_floats": m, "compress_ms": round(ct, 4), "compress_GBps": round(nbytes / ct / 1e6, 1),
                   "decompress_ms": round(dct, 4), "decompress_GBps": round(nbytes / dct / 1e6, 1),
                   "ratio": round(int(sizes[0].item()) / nbytes, 4),
                   "a100_readme_compress_GBps": A100_README[name]["compress"].get(m),
                   "a100_readme_decompress_GBps": A100_README[name]["decompress"].get(m)}
            rows_out.append(row)
            del t, comp, out, comp_ts, ts
            torch.cuda.empty_cache()
    print(json.dumps({
        "metric": "reference_protocol_float_codec_GBps",
        "protocol": "dietgpu/benchmark.py:35-86,156-175: batch of ONE tensor ~ N(0,1), torch.ops.dietgpu.compress_data / "
                    "decompress_data with a 384 MiB temp_mem, first run discarded, mean of 3, events around each op, "
                    "GB/s = uncompressed bytes / time, every run validated bit for bit",
        "hardware": "1 x MI355X (A100 columns: read off the reference's README plots, BASELINE.md section 1)",
        "rows": rows_out, "round_trip_bit_exact": True}), flush=True)


def bind_rank_to_gpu_numa_node(dev_index):
    """Pins this process to the CPUs of the NUMA node its GPU hangs off (one rank per GPU: host-side launch latency
    and the pinned parameter blocks then stay node-local).  Best effort; returns what was done, for the JSON line."""
    info = {"gpu": dev_index, "numa_node": None, "cpus": None}
    try:
        props = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        info["pci"] = bdf
        if node < 0:
            return info  # single-node host (or the platform does not say)
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if allowed:
            os.sched_setaffinity(0, allowed)
            info["numa_node"], info["cpus"] = node, len(allowed)
    except (OSError, ValueError, AttributeError) as e:
        info["note"] = f"not bound: {e}"
    return info


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:  # noqa: BLE001 -- reporting only
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~0.13 s.  Short regions (20 steps = 5 ms) include the GPU's clock ramp after an idle
    # fence (the *_steady_state extras show the difference)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="bf16", choices=["bf16", "u8", "fp16", "fp32"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--elems", type=int, default=512 * 1024,
                    help="elements per tensor for the float workloads (default = BASELINE configs: 524288)")
    ap.add_argument("--prob-bits", type=int, default=0, help="override the workload's probBits (9, 10 or 11)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective", action="store_true",
                    help="instead of the codec step: plain vs compressed all-gather of every rank's shard "
                         "(dietgpu_amd.distributed.CompressedAllGatherPlan), effective GB/s per rank")
    ap.add_argument("--chunks", type=int, default=0,
                    help="--collective: pipeline chunks per shard (default: 4 over a link, 1 at world 1, where a step has "
                         "nothing to overlap and stays on the caller's stream)")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct {input, archive, output} buffer sets of the headline (cache-cold) loop; "
                         "1 = re-code one buffer set (the memory-side cache then serves half the traffic)")
    ap.add_argument("--reference-protocol", action="store_true",
                    help="instead of the batched step: the reference's own benchmark protocol (benchmark.py:35-86, "
                         "156-175) -- batch of ONE tensor, 1 M / 16 M / 128 M / 1024 M floats, bf16 and fp16, through "
                         "torch.ops.dietgpu.compress_data / decompress_data with temp_mem, compress and decompress "
                         "GB/s separately; one JSON line with the table")
    ap.add_argument("--ref-sizes", default="1,16,128,1024", help="--reference-protocol: tensor sizes in Mi floats")
    ap.add_argument("--quick", action="store_true",
                    help="headline loop, one-buffer-set loop and the per-kernel profiles only (A/B runs: tools/ab.sh)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default line only: skip the compact lines of BASELINE configs 2 (u8) and 4 (fp16, probBits 11)")
    ap.add_argument("--timeline", action="store_true",
                    help="only the timed steps (no per-phase timing / kernel profile): for rocprofv3 timelines")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")

    # DGPU_BENCH_ONE_DEVICE=1 (with --dist-backend gloo): every rank on GPU 0 -- lets the N > 1 code path
    # be exercised on a single-GPU box; never used for measurements
    one_device = os.environ.get("DGPU_BENCH_ONE_DEVICE") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks ourselves, one per GPU, the way the
        # driver does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...)
        have = torch.cuda.device_count()
        if have < args.gpus and not one_device:
            sys.exit(f"bench.py: --gpus {args.gpus} asked for, but this node has {have} GPU(s); refusing to "
                     f"report a smaller job under that flag")
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N does it by itself)")
    distributed = world > 1
    if not one_device and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    binding = bind_rank_to_gpu_numa_node(dev_index) if (distributed and not one_device) else None
    import dietgpu_amd as dg
    from dietgpu_amd import distributed as D

    if distributed:
        import torch.distributed as dist

        D.init(backend=args.dist_backend, device=device)  # "nccl" is RCCL on ROCm

    if args.reference_protocol:
        if distributed:
            sys.exit("bench.py --reference-protocol: single GPU only (the reference's benchmark is)")
        return run_reference_protocol(args, device)
    data, ft, _, prob_bits, desc = make_workload(args.workload, args.batch, 1234 + rank, device, args.elems)
    if args.prob_bits:
        prob_bits = args.prob_bits
        desc += f" [probBits overridden: {prob_bits}]"
    if args.collective:
        return run_collective(args, data, ft, desc, world, rank, device, D)
    codec = Codec(dg, data, ft, prob_bits)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def over_ranks(seconds):
        return D.max_over_ranks(seconds, device) if distributed else seconds

    def timed_loop(fn, warmup, steps):
        """EXACTLY `warmup` untimed + `steps` timed calls of fn(i), barrier + synchronize on both sides, MAX over ranks.
        Returns seconds for the `steps` calls."""
        for i in range(warmup):
            fn(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        fence()
        return time.perf_counter() - t0

    K, W = args.steps, args.warmup
    codec.step()
    codec.verify()

    # ROTATING buffers: R distinct {input, archive, output} sets (different data) coded round robin.  With the default
    # shape a set touches ~0.7 GB and four of them 2.9 GB: neither an input nor an archive nor an output can still be
    # in the 256 MiB memory-side cache when its turn comes again.  This is the headline loop.
    sets = [codec]
    rot_sets = 1
    if args.rotate > 1:
        free_b = torch.cuda.mem_get_info(device)[0]
        per_set = codec.in_bytes * 2 + codec.comp.numel()
        rot_sets = int(max(1, min(args.rotate, (free_b - (2 << 30)) // max(per_set, 1) + 1)))
        if distributed:
            # every rank runs the same sequence of timed loops (their fences are collectives): agree on the set count
            rot_sets = int(-D.max_over_ranks(-float(rot_sets), device))
        for r in range(1, rot_sets):
            d2, _, _, _, _ = make_workload(args.workload, args.batch, 1234 + rank + 1000 * r, device, args.elems)
            c2 = Codec(dg, d2, ft, prob_bits)
            c2.temp = codec.temp  # one temp region: the calls are ordered on one stream
            sets.append(c2)
        for c2 in sets[1:]:
            c2.step()
            c2.verify()
    cold = rot_sets > 1
    rot_bytes = sum(int(c2.sizes.to(torch.int64).sum().item()) + 2 * c2.in_bytes for c2 in sets)
    step_rot = lambda i: sets[i % rot_sets].step()
    step_one = lambda i: codec.step()

    # ---- THE HEADLINE: W warm-up + K timed steps of the cache-cold loop (one buffer set only under --rotate 1)
    elapsed_local = timed_loop(step_rot, W, K)
    per_rank_ms = [elapsed_local / K * 1e3]
    world_seen = 1
    if distributed:
        per_rank_ms = [t / K * 1e3 for t in D.gather_scalars(elapsed_local, device)]
        world_seen = dist.get_world_size()
        assert world_seen == world and len(per_rank_ms) == world
    elapsed = over_ranks(elapsed_local)
    for c2 in sets:
        c2.verify()

    if args.timeline:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(elapsed / K * 1e3, 4), "loop": "rotating" if cold else "one_buffer_set"}))
        return

    extras = {}
    ms = lambda seconds, steps=K: round(over_ranks(seconds) / steps * 1e3, 4)
    # (*_steady_state: the GPU's clocks take tens of milliseconds of load to settle -- the same build measured 263 us per
    # step over 20 steps after 3 warm-up steps, 223 us in steady state, docs/HISTORY.md section 3 -- so the same loops again
    # behind a pre-roll of >= 400 steps.  Named extras, not the headline.)
    preroll = max(0, 400 - W)
    KS = max(K, 200)
    # ONE buffer set (rounds 1-3's headline, and the shape of the reference's own benchmark, which re-codes its
    # tensors): the ~180 MB archive and half of the input / output lines are served by the memory-side cache
    extras["ms_per_step_one_buffer_set"] = ms(timed_loop(step_one, W, K))
    if not args.quick:
        extras["ms_per_step_steady_state"] = ms(timed_loop(step_rot, preroll + W, KS), KS)
        extras["ms_per_step_one_buffer_set_steady_state"] = ms(timed_loop(step_one, preroll + W, KS), KS)
        # The rows of one tensor are a stride batch to the library (no parameter block on the device).  The same steps
        # as a GENERAL pointer list (elements 0 and 1 swapped): with the parameter cache warm, and with the cache off
        # (every call uploads its pointer / size arrays: one blit + event records per call).
        codec.scatter_pointers()
        extras["ms_per_step_one_buffer_set_pointer_list"] = ms(timed_loop(step_one, W, K))
        codec.lib.dgpu_debug_set_param_cache(0)
        extras["ms_per_step_one_buffer_set_param_upload_every_call"] = ms(timed_loop(step_one, W, K))
        codec.lib.dgpu_debug_set_param_cache(1)
        codec.verify()
        codec.scatter_pointers()
    # The two directions ALONE -- a sender that only compresses, a receiver that only decompresses
    if cold:
        extras["ms_compress_only"] = ms(timed_loop(lambda i: sets[i % rot_sets].encode(), W, K))
        extras["ms_decompress_only"] = ms(timed_loop(lambda i: sets[i % rot_sets].decode(), W, K))
    if cold and not args.quick:
        # ... and the round trip with the histogram pass reading through the cache (dgpu_set_histogram_load_policy(1)):
        # best for exactly this loop, not the default because it loses wherever only one direction runs (docs/HISTORY.md section 3)
        codec.lib.dgpu_set_histogram_load_policy(1)
        extras["ms_per_step_cached_histogram_loads"] = ms(timed_loop(step_rot, W, K))
        extras["ms_compress_only_cached_histogram_loads"] = ms(timed_loop(lambda i: sets[i % rot_sets].encode(), W, K))
        codec.lib.dgpu_set_histogram_load_policy(-1)
    extras["ms_compress_only_one_buffer_set"] = ms(timed_loop(lambda i: codec.encode(), W, K))
    extras["ms_decompress_only_one_buffer_set"] = ms(timed_loop(lambda i: codec.decode(), W, K))

    sizes = codec.sizes.to(torch.int64)
    comp_total = int(sizes.sum().item())
    if distributed:
        # the only collective: all-gather the per-element compressed sizes (RCCL over xGMI)
        all_sizes = D.gather_sizes(codec.sizes, args.batch * world)
        ratio = float(all_sizes.to(torch.int64).sum().item()) / (codec.in_bytes * world)
    else:
        ratio = comp_total / codec.in_bytes

    # per-kernel durations (HIP events around every launch) of the headline loop and of the one-buffer-set loop;
    # at least 100 launches per kernel: stable averages for small K
    P = max(K, 100)
    for i in range(W):
        step_rot(i)
    prof_cold = kernel_profile(codec, P, step_rot)
    for i in range(W):
        step_one(i)
    prof_warm = kernel_profile(codec, P, step_one) if cold else prof_cold
    for c2 in sets:
        c2.verify()

    if rank == 0:
        ms_per_step = elapsed / K * 1e3
        value = world * 2 * codec.in_bytes / (elapsed / K) / 1e9

        def kernel_table(prof):
            table = {}
            for name, rec in prof.items():
                avg_ms = rec["total_ms"] / max(rec["launches"], 1)
                ab = algorithmic_bytes(name, codec, comp_total)
                table[name] = {"avg_us": round(avg_ms * 1e3, 2), "launches": rec["launches"]}
                if ab:
                    table[name]["algorithmic_GBps"] = round(ab / (avg_ms * 1e-3) / 1e9, 1)
            return table

        kernels = kernel_table(prof_cold)
        kernels_warm = kernel_table(prof_warm)
        hot = [k for k in kernels if "algorithmic_GBps" in kernels[k]]
        dom = max(hot, key=lambda k: kernels[k]["avg_us"]) if hot else None
        roofline = None
        if dom:
            ach = kernels[dom]["algorithmic_GBps"]
            roofline = {
                "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4),
                "avg_us": kernels[dom]["avg_us"],
                "algorithmic_bytes": algorithmic_bytes(dom, codec, comp_total),
                "loop": "rotating buffer sets (cache-cold): the headline loop" if cold else "one buffer set (--rotate 1)",
            }
            # the PMC passes are taken on the default shape only, and count only for the build they were taken on
            traffic, note = (measured_traffic(args.workload, dom) if (args.batch == 256 and args.elems == 512 * 1024)
                             else (None, "PMC passes exist for the default shape only"))
            roofline["traffic"] = traffic
            roofline["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc.sh), "
                                          "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch; " + note)
        # whole-step figure: algorithmic bytes of encode + decode over the step time
        E = codec.B * codec.elems
        wb = (2 if ft in (1, 2) else 4) if ft else 1
        dir_alg = E * wb + comp_total  # one direction: every input byte read once, every output byte written once
        step_alg = 2 * dir_alg

        def direction(table):
            """compress = algorithmic bytes of the direction / (histogram + encode), decompress = ... / decode"""
            hist = next((v["avg_us"] for k, v in table.items() if "histogram" in k or "stats" in k), 0.0)
            enc = sum(v["avg_us"] for k, v in table.items() if k.startswith("k_ans_encode"))
            dec = sum(v["avg_us"] for k, v in table.items() if k.startswith("k_ans_decode"))
            frac = lambda us: round(dir_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if us else None
            return {"compress": {"kernels_us": round(hist + enc, 2), "histogram_us": hist, "encode_us": enc, "frac": frac(hist + enc)},
                    "decompress": {"kernels_us": round(dec, 2), "frac": frac(dec)}}

        call_frac = lambda key: (round(dir_alg / (extras[key] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if extras.get(key) else None)
        by_direction = {
            "algorithmic_bytes_per_direction": dir_alg, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "cold": direction(kernels) if cold else None,            # kernels of the rotating round trip
            "warm_one_buffer_set": direction(kernels_warm),
            # each direction running ALONE (wall clock of the library call, launch gaps included)
            "cold_alone": {"compress": {"ms": extras.get("ms_compress_only"), "frac": call_frac("ms_compress_only")},
                           "decompress": {"ms": extras.get("ms_decompress_only"), "frac": call_frac("ms_decompress_only")}} if cold else None,
            "warm_alone": {"compress": {"ms": extras["ms_compress_only_one_buffer_set"], "frac": call_frac("ms_compress_only_one_buffer_set")},
                           "decompress": {"ms": extras["ms_decompress_only_one_buffer_set"], "frac": call_frac("ms_decompress_only_one_buffer_set")}},
            "dominant_kernel_cold": dom,
        }
        out = {
            "metric": "rans_encode_decode_GBps",
            "baseline_metric": baseline_metric(),
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "world_size_seen_by_backend": world_seen,
            "dist_backend": args.dist_backend if distributed else None,
            "rccl_version": rccl_version() if distributed else None,
            "per_rank_ms_per_step": [round(t, 4) for t in per_rank_ms],
            "rank_binding": binding,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(ms_per_step, 4),
            "headline_loop": (f"{rot_sets} rotating {{input, archive, output}} buffer sets, {rot_bytes} bytes touched per "
                              f"rotation (cache-cold); exactly {W} warm-up + {K} timed steps" if cold else
                              f"ONE buffer set (--rotate 1): half the traffic is served by the 256 MiB memory-side cache"),
            "rotating_sets": rot_sets,
            "rotating_footprint_bytes": rot_bytes,
            "step_algorithmic_GBps": round(step_alg / (elapsed / K) / 1e9, 1),
            "step_frac_of_hbm_peak": round(step_alg / (elapsed / K) / 1e9 / HBM_PEAK_GBPS, 4),
            **extras,
            "steady_state_preroll_steps": preroll,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if not ft else "u8 (byte-wise rANS on the split exponent plane of 16-bit words)",
            "data": "synthetic",
            "config": {"workload": desc, "per_gpu_batch_bytes": codec.in_bytes,
                       "api": "C ABI: dgpu_float_compress + dgpu_float_decompress" if ft else
                              "C ABI: dgpu_ans_encode_batch_pointer + dgpu_ans_decode_batch_pointer",
                       "sharding": f"{world} ranks x {args.batch} independent tensors, no data-path collective"},
            "compression_ratio": round(ratio, 4),
            "roofline": roofline,
            "roofline_by_direction": by_direction,
            "kernels": kernels,
            "kernels_one_buffer_set": kernels_warm if cold else None,
            "round_trip_bit_exact": None if ablation_run() else True,
            **({"ablation": "DGPU_BENCH_ABLATION=1: the round trip was NOT checked; not a measurement"} if ablation_run() else {}),
        }
        default_line = (world == 1 and args.workload == "bf16" and args.batch == 256 and args.elems == 512 * 1024
                        and not args.prob_bits and not args.quick)
        if default_line and not args.no_other_configs:
            # The other single-GPU BASELINE configurations on the same protocol (cache-cold loop, bit-exact check), so
            # that the one command the driver runs carries driver-run evidence for every one of them.
            del sets[1:]
            torch.cuda.empty_cache()
            KO, WO = max(K, 50), max(W, 5)
            out["other_configs"] = {
                "config2_u8": compact_config_line(dg, "u8", device, KO, WO, rot_sets),
                "config4_fp16_p11": compact_config_line(dg, "fp16", device, KO, WO, rot_sets),
            }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, prob_bits)
        print(json.dumps(out), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
