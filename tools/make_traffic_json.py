#!/usr/bin/env python3
"""Derives HBM bytes per launch from the PMC passes of tools/gpu_pmc.sh (pmcrows text) and writes
profiles/<tag>_hbm_traffic.json (read by bench.py for roofline.traffic).
  tools/make_traffic_json.py <tag> <workload>=<pmc file> ...
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts half of the bytes of wide streaming reads on
gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for every store shape the codec uses
(profiles/r02_write_calib.txt).  The file is stamped with bench.kernel_source_hash(): bench.py reports
`roofline.traffic` only when the stamp matches the sources of the build it runs."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (the stamp bench.py compares with the build it runs)

tag = sys.argv[1]
out = {"kernel_source_hash": kernel_source_hash(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 1`, "
                 "mean per dispatch; tools/gpu_pmc.sh",
       "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "workloads": {}}
for arg in sys.argv[2:]:
    wl, path = arg.split("=", 1)
    cur, rec = None, {}
    for line in open(path):
        m = re.match(r"^(k_\w+)", line)
        if m:
            cur = m.group(1)
            rec[cur] = {}
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)\s+\(dispatch ([\d.]+) us\)", line)
        if m and cur:
            rec[cur][m.group(1)] = float(m.group(2))
            rec[cur][m.group(1) + "_dispatch_us"] = float(m.group(3))
    for k, r in rec.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["hbm_bytes_per_launch"] = int((2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024)
    out["workloads"][wl] = rec
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_hbm_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(path)
for wl, rec in out["workloads"].items():
    for k, r in rec.items():
        print(wl, k, r.get("hbm_bytes_per_launch"))
