"""Loads tests/gen/refgen.cpp (built on demand with g++) -- test helper only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "gen", "refgen.cpp")
_SO = os.path.join(_HERE, "gen", "librefgen.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", _SO, _SRC])
        _lib = C.CDLL(_SO)
        _lib.generate_symbols.argtypes = [C.c_void_p, C.c_int, C.c_float]
        _lib.generate_normals.argtypes = [C.c_void_p, C.c_int]
    return _lib


def generate_symbols(num, lam=20.0):
    """dietgpu/ans/ANSTest.cu:18-31"""
    out = np.zeros(max(num, 1), np.uint8)
    _load().generate_symbols(out.ctypes.data_as(C.c_void_p), num, lam)
    return out[:num]


def generate_normals(num):
    out = np.zeros(max(num, 1), np.float32)
    _load().generate_normals(out.ctypes.data_as(C.c_void_p), num)
    return out[:num]


def generate_floats(ft, num):
    """dietgpu/float/FloatTest.cu:21-120. ft: 1 fp16 (RNE), 2 bf16 (truncate), 3 fp32."""
    f = generate_normals(num)
    if ft == 1:
        return f.astype(np.float16).view(np.uint16).copy()
    if ft == 2:
        return (f.view(np.uint32) >> 16).astype(np.uint16)
    return f.view(np.uint32).copy()


# ---- BASELINE.md config generators ----
def zipf_bytes(batch, n, s=1.2, seed=1234):
    p = 1.0 / (np.arange(256) + 1.0) ** s
    p /= p.sum()

    def row(b):  # every row is its own stream: rows can be drawn concurrently
        return np.random.default_rng(seed + b).choice(256, size=n, p=p).astype(np.uint8)

    if batch * n >= (1 << 24):
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:  # (numpy releases the GIL in choice / searchsorted)
            return np.stack(list(ex.map(row, range(batch))))
    return np.stack([row(b) for b in range(batch)])


def f32_to_bf16_rne(f):
    u = np.ascontiguousarray(f, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def normal_bf16(batch, n, seed=1234):
    f = np.random.default_rng(seed).standard_normal((batch, n), dtype=np.float32)
    return f32_to_bf16_rne(f)


def sparse_fp16(batch, n, seed=1234):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((batch, n), dtype=np.float32)
    f[rng.random((batch, n)) < 0.5] = 0.0
    return f.astype(np.float16).view(np.uint16)
