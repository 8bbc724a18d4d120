#!/usr/bin/env python3
"""Rewrites the measurement table of DESIGN.md (between the `table:begin` / `table:end` markers) from profiles/<tag>_bench_*.json,
so that the document quotes exactly what the evidence run wrote.  Usage: python tools/design_table.py [tag]"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def load(name):
    with open(os.path.join(root, "profiles", f"{tag}_bench_{name}.json")) as f:
        return json.load(f)


def kern(table):
    t = {k[2:]: v["avg_us"] for k, v in (table or {}).items()}
    hist = next((v for k, v in t.items() if "histogram" in k or "stats" in k), 0.0)
    enc = sum(v for k, v in t.items() if k.startswith("ans_encode"))
    dec = sum(v for k, v in t.items() if k.startswith("ans_decode"))
    return f"{hist:.1f} + {enc:.1f} + {dec:.1f}"


def row(label, d, warm=False, bold=False):
    ms = d["ms_per_step_one_buffer_set"] if warm else d["ms_per_step"]
    frac = d["step_frac_of_hbm_peak"] * d["ms_per_step"] / ms
    kt = kern(d["kernels_one_buffer_set"] if warm else d["kernels"])
    alone = (f'{d.get("ms_compress_only_one_buffer_set")} / {d.get("ms_decompress_only_one_buffer_set")}' if warm else
             f'{d.get("ms_compress_only")} / {d.get("ms_decompress_only")}')
    b = "**" if bold else ""
    return f"| {label} | {b}{ms:.4f}{b} | {b}{frac:.3f}{b} | {kt} | {alone} |"


bf = load("bf16")
rows = ["| config (cold loop, `profiles/%s_bench_*.json`) | ms / step | fraction of 8 TB/s | kernels, µs (hist + encode + decode) | alone: compress / decompress ms |" % tag,
        "|---|---|---|---|---|",
        row("**3: 256 × 512 Ki bf16, P 10** (%d steps)" % bf["steps"], bf, bold=True),
        row("... driver protocol (20 steps after 5)", load("bf16_driver_protocol")),
        row("... one buffer set (warm)", bf, warm=True),
        row("2: 256 × 1 MiB Zipf bytes, P 10", load("u8")),
        row("4: 256 × 512 Ki fp16 50 % zeros, P 11", load("fp16")),
        row("256 × 256 Ki fp32", load("fp32"))]
for shape, label in (("32768x4096", "32768 × 4 Ki bf16 (single-block pairs)"), ("16384x8192", "16384 × 8 Ki bf16 (2-block tiles)"),
                     ("8192x16384", "8192 × 16 Ki bf16 (4-block tiles)"), ("2048x65536", "2048 × 64 Ki bf16"),
                     ("64x2097152", "64 × 2 Mi bf16"), ("16x8388608", "16 × 8 Mi bf16"), ("1x134217728", "1 × 128 Mi bf16")):
    rows.append(row(label, load("bf16_" + shape)))
path = os.path.join(root, "DESIGN.md")
s = open(path).read()
new = "<!-- table:begin -->\n" + "\n".join(rows) + "\n<!-- table:end -->"
s2, n = re.subn(r"<!-- table:begin -->.*?<!-- table:end -->", lambda m: new, s, flags=re.S)
assert n == 1, "markers not found"
open(path, "w").write(s2)
print("\n".join(rows))
