#!/usr/bin/env python3
"""Generates the golden fixtures (tests/golden/golden.json + *.bin) FROM THE REFERENCE ITSELF.

The archives are produced by oracle/_ref/libdietgpu_ref.so: the reference's own sources
(facebookresearch/dietgpu, /root/reference) compiled with g++ against the CPU emulation of the CUDA
execution model in oracle/ref_shim/ and executed (see oracle/Makefile, oracle/ref.py).  The reference
ships no golden bitstreams, so these are the reference's outputs "run here".  Bytes the reference
leaves indeterminate (uninitialised header words, tests/refmask.py) are blanked before hashing /
storing, which is also how the oracle and the HIP path write them.

Inputs are the reference's own deterministic generators (ANSTest.cu:18-31, FloatTest.cu:110-120) and
the BASELINE.md config generators.
Run from the repo root in the container that has /root/reference:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import refgen  # noqa: E402
from oracle import ref as R  # noqa: E402
from refmask import mask_ans, mask_float  # noqa: E402

if not R.available():
    R.build()

FLOAT16, BFLOAT16, FLOAT32 = 1, 2, 3


def digest(a):
    return hashlib.sha256(a.tobytes()).hexdigest()


cases = []
for p in (9, 10, 11):
    for lam in (1.0, 100.0):
        for n in (1, 4095, 4096, 10013, 100000):
            for ck in (True, False):
                if not ck and n not in (4096, 10013):
                    continue
                x = refgen.generate_symbols(n, lam)
                a = mask_ans(R.ans_encode_batch([x], p, ck)[0])
                cases.append(dict(kind="ans", prob_bits=p, lam=lam, n=n, checksum=ck, size=int(a.size), sha256=digest(a)))
for ft in (FLOAT16, BFLOAT16, FLOAT32):
    for p in (9, 10, 11):
        for n in (1, 17, 8192, 50003):
            w = refgen.generate_floats(ft, n)
            a = mask_float(R.float_compress_batch(ft, [w], p, True)[0])
            cases.append(dict(kind="float", float_type=ft, prob_bits=p, n=n, checksum=True, size=int(a.size), sha256=digest(a)))
# BASELINE configs 2, 3, 4: rows 0 and 5 of each, full size
for row in (0, 5):
    x = refgen.zipf_bytes(row + 1, 1 << 20)[row]
    a = mask_ans(R.ans_encode_batch([x], 10, False)[0])
    cases.append(dict(kind="config2", row=row, prob_bits=10, checksum=False, size=int(a.size), sha256=digest(a)))
    w = refgen.normal_bf16(row + 1, 512 * 1024)[row]
    a = mask_float(R.float_compress_batch(BFLOAT16, [w], 10, False)[0])
    cases.append(dict(kind="config3", row=row, prob_bits=10, checksum=False, size=int(a.size), sha256=digest(a)))
    h = refgen.sparse_fp16(row + 1, 512 * 1024)[row]
    a = mask_float(R.float_compress_batch(FLOAT16, [h], 11, False)[0])
    cases.append(dict(kind="config4", row=row, prob_bits=11, checksum=False, size=int(a.size), sha256=digest(a)))
# two small archives kept in full
x = refgen.generate_symbols(5000, 20.0)
mask_ans(R.ans_encode_batch([x], 10, True)[0]).tofile(os.path.join(HERE, "ans_p10_lam20_n5000.bin"))
w = refgen.generate_floats(BFLOAT16, 6000)
mask_float(R.float_compress_batch(BFLOAT16, [w], 10, True)[0]).tofile(os.path.join(HERE, "bf16_p10_n6000.bin"))
meta = dict(source="reference: oracle/_ref/libdietgpu_ref.so = facebookresearch/dietgpu sources on the dgemu CPU SIMT "
                   "emulation (oracle/ref_shim/); indeterminate header bytes blanked (tests/refmask.py)",
            generator="tests/golden/make_golden.py")
json.dump(dict(meta=meta, cases=cases), open(os.path.join(HERE, "golden.json"), "w"), indent=1)
print(len(cases), "cases written")
