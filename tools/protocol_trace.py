#!/usr/bin/env python3
"""Per-step view of the driver's bench command from a rocprofv3 kernel trace and a clock log:

    tools/protocol_trace.py <bench_results.db> <clock_sampler output> [warmup steps] [timed steps]

Finds the HEADLINE region of bench.py in the trace -- the first run of exactly warmup + steps consecutive
{histogram, encode, decode} triples with no other kernel between them -- and prints, per step: start, the three kernel
durations, the three gaps behind them, the step period, and the GPU's clocks / power at that moment.  Below the table:
the same averages over the steady-state loop later in the same process (the longest run of triples), and the
difference of the two attributed to kernels (per kernel) and gaps."""
import bisect
import re
import sqlite3
import statistics
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*", "", name)
    return name.replace("dgpu::", "")


def load_kernels(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    return [(s, e, short(n), "dgpu::" in n) for n, s, e in rows]


def kind(name):
    if "histogram" in name or "stats" in name:
        return "h"
    if "encode" in name:
        return "e"
    if "decode" in name:
        return "d"
    return "?"


def runs_of_triples(ks):
    """maximal runs of consecutive h, e, d triples: list of lists of (h, e, d)"""
    runs, cur, i = [], [], 0
    while i < len(ks):
        if i + 2 < len(ks) and all(k[3] for k in ks[i:i + 3]) and "".join(kind(k[2]) for k in ks[i:i + 3]) == "hed":
            cur.append((ks[i], ks[i + 1], ks[i + 2]))
            i += 3
        else:
            if cur:
                runs.append(cur)
            cur = []
            i += 1
    if cur:
        runs.append(cur)
    return runs


def load_clocks(path):
    rows = []
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.split()
        if len(f) < 8:
            continue
        rows.append([None if x == "-" else int(x) for x in f])
    return rows


def main():
    db, clk = sys.argv[1], sys.argv[2]
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    ks = load_kernels(db)
    runs = runs_of_triples(ks)
    print(f"# {len(ks)} kernels in the trace, {sum(len(r) for r in runs)} codec steps in {len(runs)} runs "
          f"(run lengths: {[len(r) for r in runs][:40]})")
    head = next((r for r in runs if len(r) == W + K), None)
    if head is None:
        print(f"# no run of exactly {W + K} steps: taking the first run of at least that length")
        head = next(r for r in runs if len(r) >= W + K)[: W + K]
    # bench.py's loops after the headline: one buffer set (W + K), then the steady-state pair -- the ROTATING loop behind
    # its pre-roll (400 + max(K, 200) steps), then the one-buffer-set loop the same way -- with no other kernel between
    # them: one long run.  The rotating steady state is steps [W + K + 400, W + K + 400 + max(K, 200)) of it.
    longest = max(runs, key=len)
    KS = max(K, 200)
    lo = W + K + 400
    steady = longest[lo: lo + KS] if len(longest) >= lo + KS else longest[len(longest) // 2:]
    clocks = load_clocks(clk)
    col = None
    if clocks:
        t = head[0][0][0]
        for c in range(3):
            if clocks[0][c] <= t <= clocks[-1][c]:
                col = c
        print(f"# clock log: {len(clocks)} samples, median period "
              f"{statistics.median(b[0] - a[0] for a, b in zip(clocks, clocks[1:])) / 1e3:.0f} us; profiler time base = "
              f"{['CLOCK_MONOTONIC', 'CLOCK_BOOTTIME', 'CLOCK_REALTIME'][col] if col is not None else 'NOT MATCHED'}")
    times = [r[col] for r in clocks] if col is not None else []

    def clock_at(t):
        if not times:
            return ("-",) * 5
        i = min(max(bisect.bisect_left(times, t), 0), len(times) - 1)
        r = clocks[i]
        return tuple("-" if v is None else v for v in (r[3], r[4], r[5], r[6], None if r[7] is None else r[7] // 1000000))

    t0 = head[0][0][0]
    print("%5s %10s | %8s %7s | %8s %7s | %8s %7s | %9s | %6s %6s %6s %6s %5s" % (
        "step", "start_us", "hist_us", "gap", "enc_us", "gap", "dec_us", "gap", "period_us", "sclk", "mclk", "fclk", "soc", "W"))
    recs = []
    for i, (h, e, d) in enumerate(head):
        nxt = head[i + 1][0][0] if i + 1 < len(head) else None
        rec = {"h": (h[1] - h[0]) / 1e3, "g1": (e[0] - h[1]) / 1e3, "e": (e[1] - e[0]) / 1e3, "g2": (d[0] - e[1]) / 1e3,
               "d": (d[1] - d[0]) / 1e3, "g3": (nxt - d[1]) / 1e3 if nxt else None, "p": (nxt - h[0]) / 1e3 if nxt else None}
        recs.append(rec)
        c = clock_at(h[0])
        print("%5s %10.1f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7s | %9s | %6s %6s %6s %6s %5s" % (
            ("w%d" % i) if i < W else ("t%d" % (i - W)), (h[0] - t0) / 1e3, rec["h"], rec["g1"], rec["e"], rec["g2"], rec["d"],
            "-" if rec["g3"] is None else "%.1f" % rec["g3"], "-" if rec["p"] is None else "%.1f" % rec["p"], *c))
    timed = recs[W:]
    span = (head[-1][2][1] - head[W][0][0]) / 1e3 / K

    def avg(rs, k):
        v = [r[k] for r in rs if r[k] is not None]
        return sum(v) / max(len(v), 1)

    st = []
    for i, (h, e, d) in enumerate(steady):
        nxt = steady[i + 1][0][0] if i + 1 < len(steady) else None
        st.append({"h": (h[1] - h[0]) / 1e3, "g1": (e[0] - h[1]) / 1e3, "e": (e[1] - e[0]) / 1e3, "g2": (d[0] - e[1]) / 1e3,
                   "d": (d[1] - d[0]) / 1e3, "g3": (nxt - d[1]) / 1e3 if nxt else None, "p": (nxt - h[0]) / 1e3 if nxt else None})
    print()
    print("# averages per step                     hist      enc      dec   gap h>e  gap e>d  gap d>h   kernels     gaps    period")
    for label, rs in (("timed region (%d steps)" % K, timed), ("steady state (rotating loop behind 400 steps, %d steps)" % len(st), st)):
        kern = avg(rs, "h") + avg(rs, "e") + avg(rs, "d")
        gaps = avg(rs, "g1") + avg(rs, "g2") + avg(rs, "g3")
        print("# %-36s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f  %8.1f %8.1f  %8.1f" % (
            label, avg(rs, "h"), avg(rs, "e"), avg(rs, "d"), avg(rs, "g1"), avg(rs, "g2"), avg(rs, "g3"), kern, gaps, avg(rs, "p")))
    print("# timed region, first kernel start -> last kernel end, per step: %.1f us" % span)
    print("# timed minus steady state: histogram %+.1f, encode %+.1f, decode %+.1f, gaps %+.1f us per step" % (
        avg(timed, "h") - avg(st, "h"), avg(timed, "e") - avg(st, "e"), avg(timed, "d") - avg(st, "d"),
        (avg(timed, "g1") + avg(timed, "g2") + avg(timed, "g3")) - (avg(st, "g1") + avg(st, "g2") + avg(st, "g3"))))
    if times:
        a, b = head[0][0][0], head[-1][2][1]
        inside = [r for r in clocks if a <= r[col] <= b]
        s_lo, s_hi = steady[0][0][0], steady[-1][2][1]
        inside_st = [r for r in clocks if s_lo <= r[col] <= s_hi]
        for label, rs in (("headline region", inside), ("steady-state loop", inside_st)):
            if rs:
                for idx, nm in ((3, "sclk"), (4, "mclk"), (5, "fclk"), (6, "socclk")):
                    v = [r[idx] for r in rs if r[idx] is not None]
                    if v:
                        print(f"# {label}: {nm} min / median / max = {min(v)} / {statistics.median(v)} / {max(v)} MHz over {len(v)} samples")
                v = [r[7] for r in rs if r[7] is not None]
                if v:
                    print(f"# {label}: power median {statistics.median(v) / 1e6:.0f} W")


if __name__ == "__main__":
    main()
