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


def stage_python_tests():
    """Stages the reference's own Python tests (dietgpu/ans_test.py, float_test.py) in oracle/_ref/ next to the library
    built from its sources: same category (reference-derived TEST artefact, git-ignored, never committed), same reason
    (the GPU box has no /root/reference; oracle/_ref/ travels there).  tests/test_reference_python_tests.py runs them."""
    import shutil

    dst = os.path.join(os.path.dirname(LIB), "reference_python_tests")
    os.makedirs(dst, exist_ok=True)
    for name in ("ans_test.py", "float_test.py"):
        shutil.copyfile(os.path.join(REFERENCE_ROOT, "dietgpu", name), os.path.join(dst, name))
    return dst


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
        L.dgref_float_compress_batch.argtypes = [u32, i32, i32, i32, u32, vp, vp, vp, vp]
        L.dgref_float_decompress_batch.restype = i32
        L.dgref_float_decompress_batch.argtypes = [u32, i32, i32, i32, u32, vp, vp, vp, vp, vp]
        L.dgref_ans_encode_batch_stride.restype = None
        L.dgref_ans_encode_batch_stride.argtypes = [i32, i32, u32, vp, u32, u32, vp, u32, vp]
        L.dgref_ans_decode_batch_stride.restype = i32
        L.dgref_ans_decode_batch_stride.argtypes = [i32, i32, u32, vp, u32, vp, u32, u32, vp, vp]
        L.dgref_ans_encode_batch_split_size.restype = None
        L.dgref_ans_encode_batch_split_size.argtypes = [i32, i32, u32, vp, vp, vp, u32, vp]
        L.dgref_ans_decode_batch_split_size.restype = i32
        L.dgref_ans_decode_batch_split_size.argtypes = [i32, i32, u32, vp, vp, vp, vp, vp]
        L.dgref_float_compress_split_size.restype = None
        L.dgref_float_compress_split_size.argtypes = [u32, i32, i32, i32, u32, vp, vp, vp, u32, vp]
        L.dgref_float_decompress_split_size.restype = i32
        L.dgref_float_decompress_split_size.argtypes = [u32, i32, i32, i32, u32, vp, vp, vp, vp, vp]
        L.dgref_ans_encode_batch_hist.restype = None
        L.dgref_ans_encode_batch_hist.argtypes = [i32, i32, u32, vp, vp, vp, vp, vp]
        L.dgref_ans_encode_batch_stride_hist.restype = None
        L.dgref_ans_encode_batch_stride_hist.argtypes = [i32, i32, u32, vp, u32, u32, vp, vp, u32, vp]
        L.dgref_ans_encode_batch_split_size_hist.restype = None
        L.dgref_ans_encode_batch_split_size_hist.argtypes = [i32, i32, u32, vp, vp, vp, vp, u32, vp]
        for name in ("dgref_ans_get_compressed_info", "dgref_ans_get_compressed_info_device"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [u32, vp, vp, vp]
        for name in ("dgref_float_get_compressed_info", "dgref_float_get_compressed_info_device"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [u32, vp, vp, vp, vp]
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


def float_compress_batch(ft, rows, prob_bits=10, use_checksum=False, aligned16=False):
    """aligned16: FloatCodecConfig::is16ByteAligned (every buffer here IS 16-byte aligned): the reference's
    vectorised split path (SplitFloatAligned16, GpuFloatCompress.cuh:85-278)."""
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
    L.dgref_float_compress_batch(ft, prob_bits, int(use_checksum), int(aligned16), len(rows), _ptrs(ins), sizes, _ptrs(outs),
                                 out_sizes.ctypes.data_as(C.c_void_p))
    return [o[:n].copy() for o, n in zip(outs, out_sizes)]


def float_decompress_batch(ft, archives, capacities, prob_bits=10, use_checksum=False, aligned16=False):
    """aligned16: the reference's JoinFloatAligned16 path (GpuFloatDecompress.cuh:25-270)."""
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
    rc = L.dgref_float_decompress_batch(ft, prob_bits, int(use_checksum), int(aligned16), len(archives), _ptrs(ins), _ptrs(outs), caps,
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


# ---- the reference's other batch providers (stride, split size) ---------------------------------------------
def ans_encode_batch_stride(matrix, prob_bits=10, use_checksum=False, in_stride=None):
    """[B][n] uint8 -> list of archives through ansEncodeBatchStride (rows `in_stride` bytes apart)."""
    L = lib()
    b, n = matrix.shape
    in_stride = in_stride or n
    buf = _aligned(b * in_stride + 16)
    for i in range(b):
        buf[i * in_stride : i * in_stride + n] = matrix[i]
    out_stride = int(L.dgref_ans_max_compressed_size(n))
    out = _aligned(b * out_stride)
    sizes = np.zeros(b, np.uint32)
    L.dgref_ans_encode_batch_stride(prob_bits, int(use_checksum), b, C.c_void_p(buf.ctypes.data), n, in_stride,
                                    C.c_void_p(out.ctypes.data), out_stride, sizes.ctypes.data_as(C.c_void_p))
    return [out[i * out_stride : i * out_stride + sizes[i]].copy() for i in range(b)]


def ans_decode_batch_stride(archives, n, prob_bits=10, use_checksum=False):
    L = lib()
    b = len(archives)
    in_stride = (max(len(a) for a in archives) + 15) // 16 * 16
    buf = _aligned(b * in_stride)
    for i, a in enumerate(archives):
        buf[i * in_stride : i * in_stride + len(a)] = a
    out = _aligned(b * n, 0xCD)
    ok = np.zeros(b, np.uint8)
    osz = np.zeros(b, np.uint32)
    rc = L.dgref_ans_decode_batch_stride(prob_bits, int(use_checksum), b, C.c_void_p(buf.ctypes.data), in_stride,
                                         C.c_void_p(out.ctypes.data), n, n, ok.ctypes.data_as(C.c_void_p),
                                         osz.ctypes.data_as(C.c_void_p))
    return out.reshape(b, n).copy(), ok, osz, rc


def ans_encode_batch_split_size(rows, prob_bits=10, use_checksum=False):
    """rows back to back in ONE buffer (BatchProviderSplitSize) -> list of archives."""
    L = lib()
    sizes_in = [len(r) for r in rows]
    buf = _aligned(sum(sizes_in) + 16)
    pos = 0
    for r in rows:
        buf[pos : pos + len(r)] = r
        pos += len(r)
    out_stride = int(L.dgref_ans_max_compressed_size(max(sizes_in)))
    out = _aligned(len(rows) * out_stride)
    split = (C.c_uint32 * len(rows))(*sizes_in)
    sizes = np.zeros(len(rows), np.uint32)
    L.dgref_ans_encode_batch_split_size(prob_bits, int(use_checksum), len(rows), C.c_void_p(buf.ctypes.data), split,
                                        C.c_void_p(out.ctypes.data), out_stride, sizes.ctypes.data_as(C.c_void_p))
    return [out[i * out_stride : i * out_stride + sizes[i]].copy() for i in range(len(rows))]


def ans_decode_batch_split_size(archives, sizes_out, prob_bits=10, use_checksum=False):
    L = lib()
    ins = []
    for a in archives:
        b = _aligned(len(a))
        b[:] = a
        ins.append(b)
    out = _aligned(sum(sizes_out) + 16, 0xCD)
    split = (C.c_uint32 * len(archives))(*sizes_out)
    ok = np.zeros(len(archives), np.uint8)
    osz = np.zeros(len(archives), np.uint32)
    rc = L.dgref_ans_decode_batch_split_size(prob_bits, int(use_checksum), len(archives), _ptrs(ins),
                                             C.c_void_p(out.ctypes.data), split, ok.ctypes.data_as(C.c_void_p),
                                             osz.ctypes.data_as(C.c_void_p))
    outs, pos = [], 0
    for n in sizes_out:
        outs.append(out[pos : pos + n].copy())
        pos += n
    return outs, ok, osz, rc


def float_compress_split_size(ft, rows, prob_bits=10, use_checksum=False, aligned16=False):
    L = lib()
    wb = np.dtype(_WORD[ft]).itemsize
    sizes_in = [len(r) for r in rows]
    buf = _aligned(sum(sizes_in) * wb + 16)
    pos = 0
    for r in rows:
        r = np.ascontiguousarray(r, _WORD[ft])
        buf[pos : pos + r.size * wb] = r.view(np.uint8)
        pos += r.size * wb
    out_stride = int(L.dgref_float_max_compressed_size(ft, max(sizes_in)))
    out = _aligned(len(rows) * out_stride)
    split = (C.c_uint32 * len(rows))(*sizes_in)
    sizes = np.zeros(len(rows), np.uint32)
    L.dgref_float_compress_split_size(ft, prob_bits, int(use_checksum), int(aligned16), len(rows),
                                      C.c_void_p(buf.ctypes.data), split, C.c_void_p(out.ctypes.data), out_stride,
                                      sizes.ctypes.data_as(C.c_void_p))
    return [out[i * out_stride : i * out_stride + sizes[i]].copy() for i in range(len(rows))]


def float_decompress_split_size(ft, archives, sizes_out, prob_bits=10, use_checksum=False, aligned16=False):
    L = lib()
    wb = np.dtype(_WORD[ft]).itemsize
    ins = []
    for a in archives:
        b = _aligned(len(a))
        b[:] = a
        ins.append(b)
    out = _aligned(sum(sizes_out) * wb + 16, 0xCD)
    split = (C.c_uint32 * len(archives))(*sizes_out)
    ok = np.zeros(len(archives), np.uint8)
    osz = np.zeros(len(archives), np.uint32)
    rc = L.dgref_float_decompress_split_size(ft, prob_bits, int(use_checksum), int(aligned16), len(archives), _ptrs(ins),
                                             C.c_void_p(out.ctypes.data), split, ok.ctypes.data_as(C.c_void_p),
                                             osz.ctypes.data_as(C.c_void_p))
    outs, pos = [], 0
    for n in sizes_out:
        outs.append(out[pos * wb : (pos + n) * wb].view(_WORD[ft]).copy())
        pos += n
    return outs, ok, osz, rc


# ---- caller-supplied histograms: `histogram_dev` of ansEncodeBatch{Pointer,Stride,SplitSize} (GpuANSCodec.h:65-164) ----
def _hist_arg(counts, b):
    counts = np.ascontiguousarray(counts, np.uint32).reshape(b, 256)
    return counts, counts.ctypes.data_as(C.c_void_p)


def ans_encode_batch_hist(rows, counts, prob_bits=10, use_checksum=False, provider="pointer"):
    """rows (equal lengths for provider="stride") + [B][256] counts -> list of archives.  The reference normalises the
    counts it is GIVEN against each row's size and never looks at the data's own statistics (GpuANSEncode.cuh:692-700)."""
    L = lib()
    b = len(rows)
    keep, hp = _hist_arg(counts, b)
    sizes_in = [len(r) for r in rows]
    out_sizes = np.zeros(b, np.uint32)
    if provider == "pointer":
        ins = []
        for r in rows:
            a = _aligned(max(len(r), 4))
            a[: len(r)] = r
            ins.append(a)
        sizes = (C.c_uint32 * b)(*sizes_in)
        outs = [_aligned(int(L.dgref_ans_max_compressed_size(len(r)))) for r in rows]
        L.dgref_ans_encode_batch_hist(prob_bits, int(use_checksum), b, _ptrs(ins), sizes, hp, _ptrs(outs),
                                      out_sizes.ctypes.data_as(C.c_void_p))
        return [o[:n].copy() for o, n in zip(outs, out_sizes)]
    out_stride = int(L.dgref_ans_max_compressed_size(max(sizes_in)))
    out = _aligned(b * out_stride)
    if provider == "stride":
        n = sizes_in[0]
        assert all(x == n for x in sizes_in)
        buf = _aligned(b * n + 16)
        for i, r in enumerate(rows):
            buf[i * n : (i + 1) * n] = r
        L.dgref_ans_encode_batch_stride_hist(prob_bits, int(use_checksum), b, C.c_void_p(buf.ctypes.data), n, n, hp,
                                             C.c_void_p(out.ctypes.data), out_stride, out_sizes.ctypes.data_as(C.c_void_p))
    elif provider == "split_size":
        buf = _aligned(sum(sizes_in) + 16)
        pos = 0
        for r in rows:
            buf[pos : pos + len(r)] = r
            pos += len(r)
        split = (C.c_uint32 * b)(*sizes_in)
        L.dgref_ans_encode_batch_split_size_hist(prob_bits, int(use_checksum), b, C.c_void_p(buf.ctypes.data), split, hp,
                                                 C.c_void_p(out.ctypes.data), out_stride, out_sizes.ctypes.data_as(C.c_void_p))
    else:
        raise ValueError(provider)
    del keep
    return [out[i * out_stride : i * out_stride + out_sizes[i]].copy() for i in range(b)]


# ---- header info: ansGetCompressedInfo(Device) / floatGetCompressedInfo(Device) ---------------------------------
def _hold(archives):
    ins = []
    for a in archives:
        b = _aligned(len(a))
        b[:] = a
        ins.append(b)
    return ins


def ans_get_compressed_info(archives, want_checksum=False, device=False):
    """-> (sizes, checksums or None).  device=True goes through the *Device entry point (pointer array "on the device")."""
    ins = _hold(archives)
    sizes = np.zeros(len(archives), np.uint32)
    ck = np.zeros(len(archives), np.uint32) if want_checksum else None
    fn = lib().dgref_ans_get_compressed_info_device if device else lib().dgref_ans_get_compressed_info
    fn(len(archives), _ptrs(ins), sizes.ctypes.data_as(C.c_void_p), ck.ctypes.data_as(C.c_void_p) if want_checksum else None)
    return sizes, ck


def float_get_compressed_info(archives, want_checksum=False, device=False):
    """-> (sizes in float words, float types, checksums or None)."""
    ins = _hold(archives)
    sizes = np.zeros(len(archives), np.uint32)
    types = np.zeros(len(archives), np.uint32)
    ck = np.zeros(len(archives), np.uint32) if want_checksum else None
    fn = lib().dgref_float_get_compressed_info_device if device else lib().dgref_float_get_compressed_info
    fn(len(archives), _ptrs(ins), sizes.ctypes.data_as(C.c_void_p), types.ctypes.data_as(C.c_void_p),
       ck.ctypes.data_as(C.c_void_p) if want_checksum else None)
    return sizes, types, ck
