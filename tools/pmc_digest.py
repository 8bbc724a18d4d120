#!/usr/bin/env python3
"""Per-kernel digest of tools/gpu_pmc.sh outputs: instructions per wavefront, share of wave-cycles spent waiting,
LDS bank-conflict ratio, and the time the VALU instructions alone would take at one per 4 cycles per SIMD.

  python tools/pmc_digest.py profiles/r04_pmc_bf16.txt profiles/r04_pmc_bf16_32768x4096.txt ...
"""
import re
import sys

SIMDS, GHZ = 1024, 2.4

for path in sys.argv[1:]:
    print("==", path)
    cur, d = None, {}
    for line in open(path, errors="replace"):
        if line.strip() and not line.startswith(" "):
            cur = line.strip()
            d[cur] = {}
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(dispatch ([\d.]+) us\)", line)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
            d[cur]["us"] = float(m.group(3))
    for k, v in d.items():
        if "SQ_WAVES" not in v:
            continue
        w, cyc = v["SQ_WAVES"], v["SQ_WAVE_CYCLES"]
        print("%-34s %6.1f us  waves %6d  per wave: VALU %5d SALU %5d LDS %5d VMEM %3d | wave-cycles: waiting %2d %% (on LDS %2d %%), issuing %2d %% | "
              "LDS bank-conflict cycles / active %2d %% | VALU alone %5.1f us" % (
                  k, v["us"], w, v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w,
                  (v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]) / w, 100 * v["SQ_WAIT_ANY"] / cyc, 100 * v["SQ_WAIT_INST_LDS"] / cyc,
                  100 * v["SQ_ACTIVE_INST_ANY"] / cyc, 100 * v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1),
                  v["SQ_INSTS_VALU"] * 4 / SIMDS / (GHZ * 1e3)))
