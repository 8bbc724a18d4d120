#!/usr/bin/env python3
"""Builds an experimental variant of the library next to the default one: dietgpu_amd/lib/v_<name>.so.

  python tools/build_variant.py <name> [-DFLAG=VALUE ...]

Variants are selected at run time with DGPU_LIB=<path> (dietgpu_amd/build.py); tools/ab.sh interleaves them."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(root, "dietgpu_amd", "lib", f"v_{name}.so")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *flags, "-o", out,
       os.path.join(root, "dietgpu_amd", "csrc", "capi.hip")]
print(" ".join(cmd), flush=True)
subprocess.check_call(cmd)
