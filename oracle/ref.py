"""ctypes binding of oracle/_ref/libdietgpu_ref.so: the REFERENCE's own sources
(facebookresearch/dietgpu) compiled with g++ against a CPU emulation of the CUDA
execution model (oracle/ref_shim/).  TEST INFRASTRUCTURE ONLY: it exists to pin
oracle/dietgpu_oracle.c to what the reference's kernels really compute.

The library is built here (`make -C oracle ref`, needs /root/reference) and travels
to the GPU box prebuilt; `available()` says whether it is there.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libdietgpu_ref.so")
REFERENCE_ROOT = "/root/reference"
_lib = None

FLOAT16, BFLOAT16, FLOAT32 = 1, 2, 3
_WORD = {FLOAT16: np.uint16, BFLOAT16: np.uint16, FLOAT32: np.uint32}


def can_build():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "dietgpu", "ans"))


def build(force=False):
    """Compiles the reference sources where they lie (never copied into the repo)."""
    if not can_build():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present: oracle/_ref can only be built where the reference tree is")
    if force and os.path.exists(LIB):
        os.remove(LIB)
    subprocess.check_call(["make", "-C", _HERE, "ref", f"REF={REFERENCE_ROOT}"], stdout=subprocess.DEVNULL)
    return LIB


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise ImportError(f"{LIB} is missing (make -C oracle ref)")
        L = C.CDLL(LIB)
        u32, vp, i32 = C.c_uint32, C.c_void_p, C.c_int
        L.dgref_version.restype = C.c_char_p
        for name in ("dgref_ans_max_compressed_size", "dgref_ans_compressed_overhead"):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = [u32]
        for name in ("dgref_float_max_compressed_size", "dgref_float_uncomp_data_size"):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = [u32, u32]
        for name in ("dgref_sizeof_ans_header", "dgref_sizeof_float_header", "dgref_sizeof_warp_state"):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = []
        L.dgref_histogram.restype = None
        L.dgref_histogram.argtypes = [vp, u32, vp]
        L.dgref_normalize_batch.restype = None
        L.dgref_normalize_batch.argtypes = [u32, i32, vp, vp, vp]
        L.dgref_ans_header_fields.restype = None
        L.dgref_ans_header_fields.argtypes = [vp, vp]
        L.dgref_ans_encode_batch.restype = None
        L.dgref_ans_encode_batch.argtypes = [i32, i32, u32, vp, vp, vp, vp]
        L.dgref_ans_decode_batch.restype = i32
        L.dgref_ans_decode_batch.argtypes = [i32, i32, u32, vp, vp, vp, vp, vp]
        L.dgref_float_compress_batch.restype = None
        L.dgref_float_compress_batch.argtypes = [u32, i32, i32, u32, vp, vp, vp, vp]
        L.dgref_float_decompress_batch.restype = i32
        L.dgref_float_decompress_batch.argtypes = [u32, i32, i32, u32, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _aligned(nbytes, fill=0):
    """16-byte aligned uint8 buffer (the reference requires 16-byte aligned compressed buffers)."""
    raw = np.full(nbytes + 16, fill, np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off : off + nbytes]


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def histogram(data, misalign=0):
    """ansHistogramBatch on a copy of `data` placed `misalign` bytes past a 16-byte boundary."""
    buf = _aligned(len(data) + misalign + 16)
    view = buf[misalign : misalign + len(data)]
    view[:] = data
    counts = np.zeros(256, np.uint32)
    lib().dgref_histogram(C.c_void_p(view.ctypes.data), len(data), counts.ctypes.data_as(C.c_void_p))
    return counts


def normalize_batch(counts, totals, prob_bits):
    """[B][256] counts + [B] totals -> [B][256][4] {pdf, cdf, magic, shift} by the reference's quantizeWeights."""
    counts = np.ascontiguousarray(counts, np.uint32)
    totals = np.ascontiguousarray(totals, np.uint32)
    b = counts.shape[0]
    table = _aligned(b * 256 * 16).view(np.uint32).reshape(b, 256, 4)
    lib().dgref_normalize_batch(b, prob_bits, totals.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                table.ctypes.data_as(C.c_void_p))
    return table.copy()


def ans_encode_batch(rows, prob_bits=10, use_checksum=False):
    """List of uint8 arrays -> list of archives, produced by the reference's ansEncodeBatchPointer."""
    L = lib()
    ins = []
    for r in rows:
        a = _aligned(max(len(r), 4))
        a[: len(r)] = r
        ins.append(a)
    sizes = (C.c_uint32 * len(rows))(*[len(r) for r in rows])
    outs = [_aligned(int(L.dgref_ans_max_compressed_size(len(r)))) for r in rows]
    out_sizes = np.zeros(len(rows), np.uint32)
    L.dgref_ans_encode_batch(prob_bits, int(use_checksum), len(rows), _ptrs(ins), sizes, _ptrs(outs),
                             out_sizes.ctypes.data_as(C.c_void_p))
    return [o[:n].copy() for o, n in zip(outs, out_sizes)]


def ans_decode_batch(archives, capacities, prob_bits=10, use_checksum=False):
    L = lib()
    ins = []
    for a in archives:
        b = _aligned(len(a))
        b[:] = a
        ins.append(b)
    outs = [_aligned(max(c, 1), 0xCD) for c in capacities]
    caps = (C.c_uint32 * len(archives))(*capacities)
    ok = np.zeros(len(archives), np.uint8)
    osz = np.zeros(len(archives), np.uint32)
    rc = L.dgref_ans_decode_batch(prob_bits, int(use_checksum), len(archives), _ptrs(ins), _ptrs(outs), caps,
                                  ok.ctypes.data_as(C.c_void_p), osz.ctypes.data_as(C.c_void_p))
    return [o[:c].copy() for o, c in zip(outs, capacities)], ok, osz, rc


def float_compress_batch(ft, rows, prob_bits=10, use_checksum=False):
    L = lib()
    wb = np.dtype(_WORD[ft]).itemsize
    ins = []
    for r in rows:
        r = np.ascontiguousarray(r, _WORD[ft])
        a = _aligned(max(r.size * wb, 16))
        a[: r.size * wb] = r.view(np.uint8)
        ins.append(a)
    sizes = (C.c_uint32 * len(rows))(*[len(r) for r in rows])
    outs = [_aligned(int(L.dgref_float_max_compressed_size(ft, len(r)))) for r in rows]
    out_sizes = np.zeros(len(rows), np.uint32)
    L.dgref_float_compress_batch(ft, prob_bits, int(use_checksum), len(rows), _ptrs(ins), sizes, _ptrs(outs),
                                 out_sizes.ctypes.data_as(C.c_void_p))
    return [o[:n].copy() for o, n in zip(outs, out_sizes)]


def float_decompress_batch(ft, archives, capacities, prob_bits=10, use_checksum=False):
    L = lib()
    wb = np.dtype(_WORD[ft]).itemsize
    ins = []
    for a in archives:
        b = _aligned(len(a))
        b[:] = a
        ins.append(b)
    outs = [_aligned(max(c * wb, 16), 0xCD) for c in capacities]
    caps = (C.c_uint32 * len(archives))(*capacities)
    ok = np.zeros(len(archives), np.uint8)
    osz = np.zeros(len(archives), np.uint32)
    rc = L.dgref_float_decompress_batch(ft, prob_bits, int(use_checksum), len(archives), _ptrs(ins), _ptrs(outs), caps,
                                        ok.ctypes.data_as(C.c_void_p), osz.ctypes.data_as(C.c_void_p))
    return [o[: c * wb].view(_WORD[ft]).copy() for o, c in zip(outs, capacities)], ok, osz, rc


def ans_header_fields(archive):
    a = _aligned(len(archive))
    a[:] = archive
    out = np.zeros(13, np.uint32)
    lib().dgref_ans_header_fields(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    keys = ("magic_ok", "magic_word", "num_blocks", "total_uncompressed_words", "total_compressed_words", "prob_bits",
            "use_checksum", "checksum", "pdf_offset", "states_offset", "block_words_offset", "block_data_offset",
            "total_compressed_size")
    return dict(zip(keys, (int(v) for v in out)))
