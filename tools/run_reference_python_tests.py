#!/usr/bin/env python3
"""Runs dietgpu/ans_test.py and dietgpu/float_test.py of a facebookresearch/dietgpu checkout, unmodified, against this
repository's torch.ops.dietgpu.* (needs a GPU).  Usage: python tools/run_reference_python_tests.py <checkout>"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DIETGPU_REFERENCE_ROOT"] = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
import test_reference_python_tests as t  # noqa: E402

ok = True
for f in t.FILES:
    ok = t.run_reference_test_file(f).wasSuccessful() and ok
sys.exit(0 if ok else 1)
