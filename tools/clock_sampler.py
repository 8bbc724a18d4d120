#!/usr/bin/env python3
"""Samples the GPU's clocks and power from sysfs about once per millisecond until killed (SIGTERM / SIGINT) or until
`--seconds` have passed, one line per sample:

    monotonic_ns boottime_ns realtime_ns sclk_MHz mclk_MHz fclk_MHz socclk_MHz power_uW

Runs as an ordinary user beside the process being measured (tools/gpu_protocol_trace.sh); the three host clocks are
all written because the profiler's time base has to be matched afterwards (rocprofv3 stamps kernels in one of them).
`--probe` lists what the box exposes and what one read costs, and exits."""
import argparse
import glob
import os
import signal
import sys
import time


def find_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        src["card"] = card
        for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk"), ("socclk", "pp_dpm_socclk")):
            p = os.path.join(card, name)
            if os.path.exists(p):
                src[key] = p
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            for key, name in (("freq1", "freq1_input"), ("freq2", "freq2_input"), ("power", "power1_average"), ("power_in", "power1_input")):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    src[key] = p
        break
    return src


def read_dpm(fd):
    """pp_dpm_*: lines `N: 1234Mhz [*]`; the starred level is the current one."""
    os.lseek(fd, 0, 0)
    txt = os.read(fd, 4096).decode(errors="replace")
    cur = None
    for line in txt.splitlines():
        if "*" in line:
            try:
                cur = int(line.split(":")[1].strip().split("M")[0])
            except (IndexError, ValueError):
                pass
    return cur


def read_int(fd):
    os.lseek(fd, 0, 0)
    try:
        return int(os.read(fd, 64).decode().strip())
    except ValueError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--period-us", type=float, default=1000.0)
    ap.add_argument("-o", "--out", default="-")
    a = ap.parse_args()
    src = find_sources()
    if a.probe:
        print("sources:", src)
        for k, p in src.items():
            if k == "card":
                continue
            try:
                t0 = time.perf_counter()
                with open(p) as f:
                    txt = f.read()
                dt = time.perf_counter() - t0
                print(f"--- {k} {p} ({dt * 1e6:.0f} us per open+read)\n{txt.strip()}")
            except OSError as e:
                print(f"--- {k} {p}: {e}")
        return
    fds = {}
    for k in ("sclk", "mclk", "fclk", "socclk", "freq1", "freq2", "power", "power_in"):
        if k in src:
            try:
                fds[k] = os.open(src[k], os.O_RDONLY)
            except OSError:
                pass
    out = sys.stdout if a.out == "-" else open(a.out, "w")
    stop = []
    signal.signal(signal.SIGTERM, lambda *_: stop.append(1))
    signal.signal(signal.SIGINT, lambda *_: stop.append(1))
    out.write("# monotonic_ns boottime_ns realtime_ns sclk_MHz mclk_MHz fclk_MHz socclk_MHz power_uW  (sources: %s)\n" % src)
    t_end = time.monotonic() + a.seconds
    period = a.period_us * 1e-6
    nxt = time.monotonic()
    while not stop and time.monotonic() < t_end:
        mono, boot, real = time.clock_gettime_ns(time.CLOCK_MONOTONIC), time.clock_gettime_ns(time.CLOCK_BOOTTIME), time.clock_gettime_ns(time.CLOCK_REALTIME)

        def dpm(k, alt=None):
            if k in fds:
                return read_dpm(fds[k])
            if alt in fds:
                v = read_int(fds[alt])
                return v // 1000000 if v else None
            return None

        row = [mono, boot, real, dpm("sclk", "freq1"), dpm("mclk", "freq2"), dpm("fclk"), dpm("socclk"),
               read_int(fds["power"]) if "power" in fds else (read_int(fds["power_in"]) if "power_in" in fds else None)]
        out.write(" ".join("-" if v is None else str(v) for v in row) + "\n")
        nxt += period
        d = nxt - time.monotonic()
        if d > 0:
            time.sleep(d)
        else:
            nxt = time.monotonic()
    out.flush()


if __name__ == "__main__":
    main()
