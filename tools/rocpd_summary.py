#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd sqlite output (bench_results.db).

  tools/rocpd_summary.py stats  <db>            kernel-trace --stats style table
  tools/rocpd_summary.py pmc    <db> [<db>...]  per-kernel mean of each collected counter
  tools/rocpd_summary.py timeline <db> [n]      last n kernels / copies in start order with gaps
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.replace("dgpu::", "")[:60]


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-62s %7s %14s %12s %12s %12s %7s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "PCT"))
    for n, c, t, a, mn, mx in rows:
        print("%-62s %7d %14d %12.0f %12d %12d %6.2f%%" % (short(n), c, t, a, mn, mx, 100.0 * t / total))


def pmc(dbs, only="dgpu::"):
    acc = {}
    for db in dbs:
        con = sqlite3.connect(db)
        cols = [d[0] for d in con.execute("select * from counters_collection limit 1").description]
        kname = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {kname}, counter_name, avg(value), count(*) from counters_collection group by {kname}, counter_name"
        for n, c, v, k in con.execute(q):
            if only and only not in n:
                continue
            acc.setdefault(short(n), {})[c] = (v, k)
    counters = sorted({c for d in acc.values() for c in d})
    print("%-50s " % "KERNEL (mean per dispatch)" + " ".join("%16s" % c[:16] for c in counters))
    for n, d in sorted(acc.items()):
        print("%-50s " % n[:50] + " ".join("%16.1f" % d[c][0] if c in d else "%16s" % "-" for c in counters))


def pmcrows(dbs, only="dgpu::"):
    """One block per kernel, one counter per line (mean per dispatch)."""
    acc = {}
    for db in dbs:
        con = sqlite3.connect(db)
        q = "select kernel_name, counter_name, avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
        for n, c, v, d in con.execute(q):
            if only and only not in n:
                continue
            acc.setdefault(short(n), {})[c] = (v, d)
    for n, d in sorted(acc.items()):
        print(n)
        for c in sorted(d):
            print("    %-28s %16.1f   (dispatch %.1f us)" % (c, d[c][0], d[c][1] / 1e3))


def timeline(db, n=40):
    con = sqlite3.connect(db)
    ev = [(s, e, short(nm)) for nm, s, e in con.execute("select name, start, end from kernels")]
    try:
        ev += [(s, e, "COPY " + str(nm)) for nm, s, e in con.execute("select name, start, end from memory_copies")]
    except sqlite3.Error:
        pass
    ev.sort()
    ev = ev[-n:]
    t0, prev = ev[0][0], ev[0][0]
    print("%10s %9s %9s  %s" % ("start_us", "dur_us", "gap_us", "what"))
    for s, e, nm in ev:
        print("%10.1f %9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm))
        prev = max(prev, e)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    elif sys.argv[1] == "pmcrows":
        only = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--only=")]
        pmcrows([a for a in sys.argv[2:] if not a.startswith("--only=")], *(only[:1]))
    else:
        pmc(sys.argv[2:])
