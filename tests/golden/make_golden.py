#!/usr/bin/env python3
"""Freezes oracle outputs as golden fixtures (tests/golden/golden.json + *.bin).

The reference ships no golden bitstreams and cannot be built here (CUDA only), so these vectors come from
the CPU oracle, after it has been pinned to the reference's known-answer tests
(tests/test_oracle_known_answers.py).  Inputs are the reference's own deterministic generators
(ANSTest.cu:18-31, FloatTest.cu:110-120).  Run from the repo root: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as O  # noqa: E402
import refgen  # noqa: E402

cases = []
for p in (9, 10, 11):
    for lam in (1.0, 100.0):
        for n in (1, 4095, 4096, 10013, 100000):
            x = refgen.generate_symbols(n, lam)
            a = O.ans_encode(x, p, use_checksum=True)
            cases.append(dict(kind="ans", prob_bits=p, lam=lam, n=n, size=int(a.size),
                              sha256=hashlib.sha256(a.tobytes()).hexdigest()))
for ft in (O.FLOAT16, O.BFLOAT16, O.FLOAT32):
    for p in (9, 10):
        for n in (1, 17, 8192, 50003):
            w = refgen.generate_floats(ft, n)
            a = O.float_compress(ft, w, p, use_checksum=True)
            cases.append(dict(kind="float", float_type=ft, prob_bits=p, n=n, size=int(a.size),
                              sha256=hashlib.sha256(a.tobytes()).hexdigest()))
# two small archives kept in full
x = refgen.generate_symbols(5000, 20.0)
O.ans_encode(x, 10, use_checksum=True).tofile(os.path.join(HERE, "ans_p10_lam20_n5000.bin"))
w = refgen.generate_floats(O.BFLOAT16, 6000)
O.float_compress(O.BFLOAT16, w, 10, use_checksum=True).tofile(os.path.join(HERE, "bf16_p10_n6000.bin"))
json.dump(dict(cases=cases), open(os.path.join(HERE, "golden.json"), "w"), indent=1)
print(len(cases), "cases written")
