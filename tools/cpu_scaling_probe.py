#!/usr/bin/env python3
"""Why does the CPU baseline not scale with the visible cores?  Prints what the host gives this process (visible CPUs,
cgroup CPU quota, load) and the oracle's round-trip rate for a sweep of thread counts, each with the process's CPU
seconds per wall second (= cores actually granted).  Usage (GPU box): python tools/cpu_scaling_probe.py"""
import ctypes as C
import os
import resource
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import oracle as O  # noqa: E402
import refgen  # noqa: E402


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


print("visible cpus (sched_getaffinity):", len(os.sched_getaffinity(0)), " os.cpu_count():", os.cpu_count())
print("cgroup v2 cpu.max:", read("/sys/fs/cgroup/cpu.max"), " v1 cfs_quota_us / period_us:",
      read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "/", read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
print("cgroup v2 cpu.stat:", (read("/sys/fs/cgroup/cpu.stat") or "").replace("\n", "; "))
print("loadavg:", read("/proc/loadavg"))
print("model:", next((l.split(":")[1].strip() for l in (read("/proc/cpuinfo") or "").splitlines() if l.startswith("model name")), None))
L = O.lib()
L.dgo_bench_roundtrip.restype = C.c_int
L.dgo_bench_roundtrip.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
n = 512 * 1024
cores = len(os.sched_getaffinity(0))
data = refgen.normal_bf16(min(2 * cores, 512), n)
for threads in sorted({1, 2, 4, 8, 16, 32, 64, 128, cores}):
    if threads > cores:
        continue
    rows = min(data.shape[0], max(4, 2 * threads))
    for pin in (1, 0):
        e, d, bad = C.c_double(), C.c_double(), C.c_int()
        r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
        reps = 3
        rc = L.dgo_bench_roundtrip(2, data.ctypes.data, n, n * 2, rows, 10, threads, reps, pin, C.byref(e), C.byref(d), C.byref(bad))
        r1, t1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
        cpu = (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)
        nb = rows * n * 2 * reps
        print(f"threads {threads:4d} pin {pin} rows {rows:4d}: {2 * nb / (e.value + d.value) / 1e9:8.2f} GB/s "
              f"(enc {nb / e.value / 1e9:7.2f} dec {nb / d.value / 1e9:7.2f}); cpu-seconds per wall-second {cpu / (t1 - t0):6.1f}; rc {rc} bad {bad.value}")
print("cgroup v2 cpu.stat after:", (read("/sys/fs/cgroup/cpu.stat") or "").replace("\n", "; "))
