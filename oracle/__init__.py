"""ctypes binding of the CPU oracle (oracle/dietgpu_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package dietgpu_amd/.

Parity status: PINNED.  The reference ships no golden bitstreams and its build
system cannot run here, but its own sources compile with g++ against a CPU
emulation of the CUDA execution model (oracle/_ref, oracle/ref.py); tests/
test_reference_pin.py compares this restatement with it byte for byte, and the
golden fixtures under tests/golden/ are its outputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdietgpu_oracle.so")

FLOAT16, BFLOAT16, FLOAT32 = 1, 2, 3


def build(force=False):
    src = os.path.join(_HERE, "dietgpu_oracle.c")
    hdr = os.path.join(_HERE, "dietgpu_oracle.h")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdietgpu_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u32, u8p, vp, i32, sz = C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t
        L.dgo_ans_max_compressed_size.restype = u32
        L.dgo_ans_max_compressed_size.argtypes = [u32]
        L.dgo_float_max_compressed_size.restype = u32
        L.dgo_float_max_compressed_size.argtypes = [u32, u32]
        L.dgo_ans_compressed_overhead.restype = u32
        L.dgo_ans_compressed_overhead.argtypes = [u32]
        L.dgo_float_uncomp_data_size.restype = u32
        L.dgo_float_uncomp_data_size.argtypes = [u32, u32]
        L.dgo_histogram.restype = None
        L.dgo_histogram.argtypes = [u8p, u32, vp]
        L.dgo_normalize.restype = None
        L.dgo_normalize.argtypes = [vp, u32, i32, vp]
        L.dgo_checksum.restype = u32
        L.dgo_checksum.argtypes = [u8p, u32]
        L.dgo_ans_encode_block.restype = u32
        L.dgo_ans_encode_block.argtypes = [u8p, u32, i32, vp, vp, vp]
        L.dgo_ans_decode_block.restype = i32
        L.dgo_ans_decode_block.argtypes = [vp, u32, vp, u32, i32, vp, u8p]
        L.dgo_ans_decode_table.restype = None
        L.dgo_ans_decode_table.argtypes = [vp, i32, vp]
        L.dgo_ans_encode.restype = u32
        L.dgo_ans_encode.argtypes = [u8p, u32, i32, i32, vp, u8p]
        L.dgo_ans_decode.restype = i32
        L.dgo_ans_decode.argtypes = [u8p, i32, u8p, u32, vp]
        L.dgo_ans_info.restype = i32
        L.dgo_ans_info.argtypes = [u8p, vp, vp, vp, vp]
        L.dgo_float_split.restype = None
        L.dgo_float_split.argtypes = [u32, vp, u32, u8p, u8p]
        L.dgo_float_join.restype = None
        L.dgo_float_join.argtypes = [u32, u8p, u8p, u32, vp]
        L.dgo_float_compress.restype = u32
        L.dgo_float_compress.argtypes = [u32, vp, u32, i32, i32, u8p]
        L.dgo_float_decompress.restype = i32
        L.dgo_float_decompress.argtypes = [u32, u8p, i32, vp, u32, vp]
        L.dgo_float_info.restype = i32
        L.dgo_float_info.argtypes = [u8p, vp, vp, vp, vp]
        L.dgo_ans_encode_batch.restype = None
        L.dgo_ans_encode_batch.argtypes = [u8p, u32, sz, u32, i32, u8p, sz, vp, i32]
        L.dgo_ans_decode_batch.restype = None
        L.dgo_ans_decode_batch.argtypes = [u8p, sz, u32, i32, u8p, sz, u32, i32]
        L.dgo_float_compress_batch.restype = None
        L.dgo_float_compress_batch.argtypes = [u32, vp, u32, sz, u32, i32, u8p, sz, vp, i32]
        L.dgo_float_decompress_batch.restype = None
        L.dgo_float_decompress_batch.argtypes = [u32, u8p, sz, u32, i32, vp, sz, u32, i32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _bytes(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


_WORD = {FLOAT16: 2, BFLOAT16: 2, FLOAT32: 4}


# ---------------------------------------------------------------- sizes
def ans_max_compressed_size(n):
    return int(lib().dgo_ans_max_compressed_size(n))


def float_max_compressed_size(ft, n):
    return int(lib().dgo_float_max_compressed_size(ft, n))


def ans_compressed_overhead(nb):
    return int(lib().dgo_ans_compressed_overhead(nb))


def float_uncomp_data_size(ft, n):
    return int(lib().dgo_float_uncomp_data_size(ft, n))


# ----------------------------------------------------------- statistics
def histogram(data):
    d = _bytes(data)
    out = np.zeros(256, np.uint32)
    lib().dgo_histogram(_p(d), d.size, _p(out))
    return out


def normalize(counts, total, prob_bits):
    """Returns table [256,4] = pdf, cdf, magic, shift."""
    counts = np.ascontiguousarray(counts, np.uint32)
    table = np.zeros((256, 4), np.uint32)
    lib().dgo_normalize(_p(counts), int(total), prob_bits, _p(table))
    return table


def checksum(data):
    d = _bytes(data)
    return int(lib().dgo_checksum(_p(d), d.size))


def decode_table(pdf, prob_bits):
    pdf = np.ascontiguousarray(pdf, np.uint16)
    lut = np.zeros(1 << prob_bits, np.uint32)
    lib().dgo_ans_decode_table(_p(pdf), prob_bits, _p(lut))
    return lut


def encode_block(data, prob_bits, table):
    d = _bytes(data)
    table = np.ascontiguousarray(table, np.uint32)
    words = np.zeros(4096 + 8, np.uint16)
    state = np.zeros(32, np.uint32)
    n = lib().dgo_ans_encode_block(_p(d), d.size, prob_bits, _p(table), _p(words), _p(state))
    return words[:n].copy(), state


# -------------------------------------------------------------- archive
def ans_encode(data, prob_bits=10, use_checksum=False, counts=None):
    d = _bytes(data)
    out = np.zeros(ans_max_compressed_size(d.size) + 8192, np.uint8)
    cp = None
    if counts is not None:
        counts = np.ascontiguousarray(counts, np.uint32)
        cp = _p(counts)
    n = lib().dgo_ans_encode(_p(d), d.size, prob_bits, int(use_checksum), cp, _p(out))
    return out[:n].copy()


def ans_decode(archive, prob_bits=10, capacity=None):
    """Returns (rc, out bytes, reported size)."""
    a = _bytes(archive)
    size = C.c_uint32(0)
    if lib().dgo_ans_info(_p(a), C.byref(size), None, None, None) != 0:
        raise ValueError("bad ANS archive")
    cap = size.value if capacity is None else capacity
    out = np.zeros(max(cap, 1), np.uint8)
    rep = C.c_uint32(0)
    rc = lib().dgo_ans_decode(_p(a), prob_bits, _p(out), cap, C.byref(rep))
    return rc, out[: min(cap, rep.value)] if rc == 0 else out[:0], rep.value


def ans_info(archive):
    a = _bytes(archive)
    u, c, k, p = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().dgo_ans_info(_p(a), C.byref(u), C.byref(c), C.byref(k), C.byref(p))
    if rc != 0:
        raise ValueError("bad ANS archive")
    return dict(uncompressed=u.value, compressed=c.value, checksum=k.value, prob_bits=p.value)


# ---------------------------------------------------------------- float
def float_split(ft, words):
    w = np.ascontiguousarray(words)
    n = w.size
    comp = np.zeros(max(n, 1), np.uint8)
    nc = np.zeros(max(float_uncomp_data_size(ft, n), 1), np.uint8)
    lib().dgo_float_split(ft, _p(w), n, _p(comp), _p(nc))
    return comp[:n], nc[: float_uncomp_data_size(ft, n)]


def float_join(ft, comp, noncomp, n):
    comp = np.ascontiguousarray(comp, np.uint8)
    noncomp = np.ascontiguousarray(noncomp, np.uint8)
    out = np.zeros(max(n, 1), np.uint32 if ft == FLOAT32 else np.uint16)
    lib().dgo_float_join(ft, _p(comp), _p(noncomp), n, _p(out))
    return out[:n]


def float_compress(ft, words, prob_bits=10, use_checksum=False):
    """words: numpy array of uint16 (fp16/bf16 bit patterns) or uint32 (fp32)."""
    w = np.ascontiguousarray(words)
    assert w.dtype.itemsize == _WORD[ft]
    n = w.size
    out = np.zeros(float_max_compressed_size(ft, n) + 8192, np.uint8)
    sz = lib().dgo_float_compress(ft, _p(w), n, prob_bits, int(use_checksum), _p(out))
    return out[:sz].copy()


def float_decompress(ft, archive, prob_bits=10, capacity=None):
    a = _bytes(archive)
    n, t = C.c_uint32(0), C.c_uint32(0)
    if lib().dgo_float_info(_p(a), C.byref(n), C.byref(t), None, None) != 0:
        raise ValueError("bad float archive")
    cap = n.value if capacity is None else capacity
    out = np.zeros(max(cap, 1), np.uint32 if ft == FLOAT32 else np.uint16)
    rep = C.c_uint32(0)
    rc = lib().dgo_float_decompress(ft, _p(a), prob_bits, _p(out), cap, C.byref(rep))
    return rc, out[: min(cap, rep.value)] if rc == 0 else out[:0], rep.value


def float_info(archive):
    a = _bytes(archive)
    n, t, k, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().dgo_float_info(_p(a), C.byref(n), C.byref(t), C.byref(k), C.byref(c))
    if rc != 0:
        raise ValueError("bad float archive")
    return dict(num_floats=n.value, float_type=t.value, checksum=k.value, compressed=c.value)


# ------------------------------------------------- batch (cpu_baseline)
def ans_encode_batch(rows, prob_bits=10, threads=1):
    rows = np.ascontiguousarray(rows, np.uint8)
    b, n = rows.shape
    stride = ans_max_compressed_size(n)
    out = np.zeros((b, stride), np.uint8)
    sizes = np.zeros(b, np.uint32)
    lib().dgo_ans_encode_batch(_p(rows), n, n, b, prob_bits, _p(out), stride, _p(sizes), threads)
    return out, sizes


def ans_decode_batch(comp, n, prob_bits=10, threads=1):
    comp = np.ascontiguousarray(comp, np.uint8)
    b, stride = comp.shape
    out = np.zeros((b, n), np.uint8)
    lib().dgo_ans_decode_batch(_p(comp), stride, b, prob_bits, _p(out), n, n, threads)
    return out


def float_compress_batch(ft, rows, prob_bits=10, threads=1):
    rows = np.ascontiguousarray(rows)
    b, n = rows.shape
    stride = float_max_compressed_size(ft, n)
    out = np.zeros((b, stride), np.uint8)
    sizes = np.zeros(b, np.uint32)
    lib().dgo_float_compress_batch(
        ft, _p(rows), n, n * _WORD[ft], b, prob_bits, _p(out), stride, _p(sizes), threads
    )
    return out, sizes


def float_decompress_batch(ft, comp, n, prob_bits=10, threads=1):
    comp = np.ascontiguousarray(comp, np.uint8)
    b, stride = comp.shape
    out = np.zeros((b, n), np.uint32 if ft == FLOAT32 else np.uint16)
    lib().dgo_float_decompress_batch(
        ft, _p(comp), stride, b, prob_bits, _p(out), n * _WORD[ft], n, threads
    )
    return out
