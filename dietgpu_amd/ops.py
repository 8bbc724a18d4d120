"""Tensor API of the codec: the ten ops of `torch.ops.dietgpu.*`.

Mirrors dietgpu/DietGpu.cpp (schema strings at DietGpu.cpp:915-937; argument
validation at :149-275, :310-522, :530-911) on PyTorch-ROCm tensors: same
names, argument order, defaults, return values and error conditions
(TORCH_CHECK -> RuntimeError).  PyTorch is only plumbing here (device memory
and the current HIP stream); all work happens in libdietgpu_amd.so through the
C ABI of include/dietgpu_amd.h.

Extra keyword `prob_bits` (default 10, as kDefaultPrecision DietGpu.cpp:114)
exposes the C++ API's ANSCodecConfig.probBits in {9, 10, 11}.

Two routes to the same C ABI.  The six codec ops are handed to `torch.ops.dietgpu.*` (csrc/torch_ops.cpp: argument
checks and pointer marshalling in C++, ~2 us of host time per call) when libdietgpu_torch.so is there -- at `prob_bits`
9 / 11 with the library's thread-local precision set around the call (the registered ops themselves fix the precision
at 10, as upstream); the ctypes route below serves builds without the op library.  256 tensors per call cost ~120 us of Python per op on the
ctypes route (profiles/r03_api_rate.txt) -- `prefer_torch_ops(False)` forces it (the test-suite runs every parity test
on both routes).  Either way the work is done by libdietgpu_amd.so: there is no CPU path.
"""
import ctypes as C
import os
import threading

import torch

from ._lib import check, lib

FLOAT16, BFLOAT16, FLOAT32 = 1, 2, 3
_DTYPE_TO_FT = {torch.float16: FLOAT16, torch.bfloat16: BFLOAT16, torch.float32: FLOAT32}
_FT_TO_DTYPE = {v: k for k, v in _DTYPE_TO_FT.items()}
K_DEFAULT_PRECISION = 10
_U32_MAX = (1 << 32) - 1


_PREFER_TORCH_OPS = True
_TORCH_OPS = None  # None: not tried yet; False: libdietgpu_torch.so is not there (or stale)
_TORCH_OPS_LOCK = threading.Lock()


def prefer_torch_ops(enable=True):
    """Route the codec ops through torch.ops.dietgpu.* at prob_bits 10 (default) or force the ctypes route."""
    global _PREFER_TORCH_OPS
    _PREFER_TORCH_OPS = bool(enable)


class _OpsAtPrecision:
    """torch.ops.dietgpu.* at prob_bits 9 / 11: the op library's thread-local precision (dietgpu_amd::set_precision, an
    extra op beside the reference's ten) is set around the call and put back to the default."""

    def __init__(self, ops, prob_bits):
        self._ops, self._p = ops, prob_bits

    def __getattr__(self, name):
        fn = getattr(self._ops, name)

        def call(*args):
            torch.ops.dietgpu_amd.set_precision(self._p)
            try:
                return fn(*args)
            finally:
                torch.ops.dietgpu_amd.set_precision(K_DEFAULT_PRECISION)

        return call


def _fast_ops(prob_bits):
    global _TORCH_OPS
    if prob_bits not in (9, 10, 11) or not _PREFER_TORCH_OPS:
        return None
    if _TORCH_OPS is None:
        with _TORCH_OPS_LOCK:
            if _TORCH_OPS is None:
                _TORCH_OPS = _load_fast_ops()
    if not _TORCH_OPS:
        return None
    return _TORCH_OPS if prob_bits == K_DEFAULT_PRECISION else _OpsAtPrecision(_TORCH_OPS, prob_bits)


def _load_fast_ops():
    """torch.ops.dietgpu from the in-tree op library, or False.  An op library built against another version of the C
    ABI than the core library it finds registers NO op implementations (torch_ops.cpp: nothing may throw out of a static
    initialiser under dlopen) and says so through two plain C symbols, which are compared here: the ctypes route then
    serves the calls.  An op library from before those symbols existed is treated the same way.  (File times are NOT
    consulted: a rebuild of the core library from unchanged sources, or a copy of the tree, says nothing about the
    sources the op library was built from.)"""
    import warnings

    from .build import TORCH_LIB_PATH

    if not os.path.exists(TORCH_LIB_PATH):
        return False
    core = lib()  # libdietgpu_amd.so first (the op library links against it)
    try:
        torch.ops.load_library(TORCH_LIB_PATH)
        handle = C.CDLL(TORCH_LIB_PATH)  # (already mapped: the same handle)
    except (OSError, RuntimeError) as e:
        warnings.warn(f"{TORCH_LIB_PATH} could not be loaded ({e}): the ctypes route is used")
        return False
    try:
        handle.dgpu_torch_built_abi.restype = C.c_uint32
        built, registered = int(handle.dgpu_torch_built_abi()), int(handle.dgpu_torch_ops_registered())
    except AttributeError:
        built, registered = None, 0
    have = int(core.dgpu_abi_version())
    if built != have or not registered:
        warnings.warn(f"{TORCH_LIB_PATH} was built against C ABI version {built} of dietgpu_amd.h, libdietgpu_amd.so has "
                      f"version {have}: the ctypes route is used (rebuild: python -m dietgpu_amd.build)")
        return False
    if not hasattr(torch.ops, "dietgpu_amd") or not hasattr(torch.ops.dietgpu_amd, "set_precision"):
        return False
    return torch.ops.dietgpu


def _check(cond, msg="argument check failed"):
    if not cond:
        raise RuntimeError(msg)


def _float_type(t):
    """getFloatTypeFromDtype, DietGpu.cpp:18-32"""
    _check(t.dtype in _DTYPE_TO_FT, "tensor must be float16, bfloat16 or float32")
    return _DTYPE_TO_FT[t.dtype]


def _total_and_max(ts):
    """getTotalAndMaxSize, DietGpu.cpp:54-73"""
    total = mx = 0
    for t in ts:
        n = t.numel()
        _check(n * t.element_size() <= _U32_MAX)
        total += n
        mx = max(mx, n)
    return total, mx


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ctypes arrays are cached by content: a loop that compresses the same tensors step after step (the steady
# state the library's parameter cache is built for) then spends its host time on one tuple hash instead of
# three array constructions per call.
_ARRAY_CACHE = {}
_ARRAY_CACHE_MAX = 256


def _cached_array(ctype, values):
    key = (ctype, values)
    arr = _ARRAY_CACHE.get(key)
    if arr is None:
        if len(_ARRAY_CACHE) >= _ARRAY_CACHE_MAX:
            _ARRAY_CACHE.clear()
        arr = (ctype * len(values))(*values)
        _ARRAY_CACHE[key] = arr
    return arr


def _ptr_array(ts):
    return _cached_array(C.c_void_p, tuple(t.data_ptr() for t in ts))


def _u32_array(vals):
    return _cached_array(C.c_uint32, tuple(vals))


def _in_bytes(ts):
    """bytes each compressed input tensor holds: the decoder rejects an archive that claims more"""
    return _cached_array(C.c_uint32, tuple(min(t.numel() * t.element_size(), _U32_MAX) for t in ts))


def _temp(temp_mem, dev):
    if temp_mem is None:
        return None, 0
    _check(temp_mem.is_cuda and temp_mem.is_contiguous())
    _check(temp_mem.get_device() == dev)
    return C.c_void_p(temp_mem.data_ptr()), temp_mem.numel() * temp_mem.element_size()


# ---------------------------------------------------------------- size queries
def max_float_compressed_output_size(ts):
    _, mx = _total_and_max(ts)
    return len(ts), _guarded(int(lib().dgpu_float_max_compressed_size(_float_type(ts[0]), mx)), mx)


def max_float_compressed_size(dtype, size):
    return _guarded(int(lib().dgpu_float_max_compressed_size(_float_type(dtype), size)), size)


def max_any_compressed_output_size(ts):
    _, mx = _total_and_max(ts)
    return len(ts), _guarded(int(lib().dgpu_ans_max_compressed_size(mx * ts[0].element_size())), mx * ts[0].element_size())


def max_any_compressed_size(nbytes):
    return _guarded(int(lib().dgpu_ans_max_compressed_size(nbytes)), nbytes)


def _guarded(size, n):
    # 0 = beyond getMaxCompressedSize's CHECK_LE(rawSize, INT32_MAX) (GpuANSEncode.cu:22; upstream aborts)
    _check(size != 0, f"input of {n} symbols: its maximum compressed size exceeds INT32_MAX (1717538816 is the largest)")
    return size


# -------------------------------------------------------------------- compress
def _validate_out(out_compressed, out_compressed_bytes, rows, cols, dev, device):
    if out_compressed is not None:
        oc = out_compressed
        _check(oc.dtype == torch.uint8 and oc.is_cuda and oc.is_contiguous() and oc.dim() == 2)
        _check(oc.size(0) >= rows and oc.size(1) >= cols and oc.get_device() == dev)
        comp = oc
    else:
        comp = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    if out_compressed_bytes is not None:
        ob = out_compressed_bytes
        _check(ob.dtype == torch.int32 and ob.is_cuda and ob.dim() == 1 and ob.is_contiguous())
        _check(ob.size(0) >= rows and ob.get_device() == dev)
        sizes = ob
    else:
        sizes = torch.empty((rows,), dtype=torch.int32, device=device)
    return comp, sizes


def compress_data(compress_as_float, ts_in, checksum=False, temp_mem=None, out_compressed=None,
                  out_compressed_bytes=None, prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:149-308 -> (comp [B, maxSize] u8, sizes [B] i32, temp bytes used)."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.compress_data(compress_as_float, ts_in, checksum, temp_mem, out_compressed, out_compressed_bytes)
    _check(len(ts_in) > 0)
    dev = ts_in[0].get_device()
    rows, cols = (max_float_compressed_output_size(ts_in) if compress_as_float
                  else max_any_compressed_output_size(ts_in))
    for t in ts_in:
        _check(t.is_cuda and t.is_contiguous() and t.get_device() == dev)
        if compress_as_float:
            _check(t.dtype == ts_in[0].dtype)
            _float_type(t)
    with torch.cuda.device(dev):
        comp, sizes = _validate_out(out_compressed, out_compressed_bytes, rows, cols, dev, ts_in[0].device)
        tp, tb = _temp(temp_mem, dev)
        in_ptrs = _ptr_array(ts_in)
        row = comp.size(1)
        out_ptrs = (C.c_void_p * rows)(*[comp.data_ptr() + i * row for i in range(rows)])
        used = C.c_size_t(0)
        if compress_as_float:
            in_size = _u32_array([t.numel() for t in ts_in])
            check(lib().dgpu_float_compress(
                tp, tb, C.byref(used), _float_type(ts_in[0]), prob_bits, int(checksum), rows,
                in_ptrs, in_size, out_ptrs, _ptr(sizes), _stream()))
        else:
            in_size = _u32_array([t.numel() * t.element_size() for t in ts_in])
            check(lib().dgpu_ans_encode_batch_pointer(
                tp, tb, C.byref(used), prob_bits, int(checksum), rows, in_ptrs, in_size, None,
                out_ptrs, _ptr(sizes), _stream()))
    return comp, sizes, int(used.value)


def compress_data_split_size(compress_as_float, t_in, t_in_split_sizes, checksum=False,
                             temp_mem=None, out_compressed=None, out_compressed_bytes=None,
                             prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:310-452 -> (list of compressed row views, sizes, temp bytes used)."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.compress_data_split_size(compress_as_float, t_in, t_in_split_sizes, checksum, temp_mem, out_compressed,
                                             out_compressed_bytes)
    dev = t_in.get_device()
    _check(t_in.is_cuda and t_in.is_contiguous())
    ft = _float_type(t_in) if compress_as_float else 0
    if not compress_as_float:
        _check(t_in.data_ptr() % 4 == 0, "start pointer is not aligned")
    ss = t_in_split_sizes
    _check(ss.is_contiguous() and not ss.is_cuda and ss.dtype == torch.int32)
    split = [int(v) for v in ss.tolist()]
    n = len(split)
    for i, s in enumerate(split):
        _check(s > 0)
        if not compress_as_float and i != n - 1:
            _check(s % 4 == 0, "the size of an interior split is not a multiple of the alignment")
    mx = max(split)
    cols = _guarded(int(lib().dgpu_float_max_compressed_size(ft, mx)) if compress_as_float
                    else int(lib().dgpu_ans_max_compressed_size(mx)), mx)
    with torch.cuda.device(dev):
        comp, sizes = _validate_out(out_compressed, out_compressed_bytes, n, cols, dev, t_in.device)
        tp, tb = _temp(temp_mem, dev)
        used = C.c_size_t(0)
        if compress_as_float:
            check(lib().dgpu_float_compress_split_size(
                tp, tb, C.byref(used), ft, prob_bits, int(checksum), n, _ptr(t_in),
                _u32_array(split), _ptr(comp), comp.size(1), _ptr(sizes), _stream()))
        else:
            check(lib().dgpu_ans_encode_batch_split_size(
                tp, tb, C.byref(used), prob_bits, int(checksum), n, _ptr(t_in), _u32_array(split),
                None, _ptr(comp), comp.size(1), _ptr(sizes), _stream()))
        # compressedMatrixToTensors, DietGpu.cpp:77-104
        host_sizes = sizes[:n].tolist()
        flat = comp.view(-1)
        outs = [flat.narrow(0, i * comp.size(1), host_sizes[i]) for i in range(n)]
    return outs, sizes, int(used.value)


def compress_data_simple(compress_as_float, ts_in, checksum=False, temp_mem=67108864,
                         prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:454-522 -> list of exactly-sized compressed tensors."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.compress_data_simple(compress_as_float, ts_in, checksum, temp_mem)
    _check(len(ts_in) > 0)
    scratch = None
    if temp_mem is not None and temp_mem > 0:
        scratch = torch.empty((temp_mem,), dtype=torch.uint8, device=ts_in[0].device)
    comp, sizes, _ = compress_data(compress_as_float, ts_in, checksum, scratch, None, None,
                                   prob_bits=prob_bits)
    host = sizes.to("cpu").tolist()
    return [comp[i, : host[i]].clone() for i in range(len(ts_in))]


# ------------------------------------------------------------------ decompress
def _validate_status(out_status, out_sizes, n, dev):
    if out_status is not None:
        _check(out_status.is_contiguous() and out_status.is_cuda and out_status.dtype == torch.uint8)
        _check(out_status.numel() == n and out_status.get_device() == dev)
    if out_sizes is not None:
        _check(out_sizes.is_contiguous() and out_sizes.is_cuda and out_sizes.dtype == torch.int32)
        _check(out_sizes.numel() == n and out_sizes.get_device() == dev)


def _raise_checksum(rc, is_float):
    if rc == 3:  # DGPU_ERR_CHECKSUM_MISMATCH
        raise RuntimeError(
            ("floatDecompress" if is_float else "ANSDecode")
            + ": checksum mismatch seen on decoded data; archive cannot be unpacked\n"
            + lib().dgpu_last_error().decode())
    check(rc)


def decompress_data(compress_as_float, ts_in, ts_out, checksum=False, temp_mem=None,
                    out_status=None, out_decompressed_words=None, prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:530-677 -> temp bytes used."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.decompress_data(compress_as_float, ts_in, ts_out, checksum, temp_mem, out_status, out_decompressed_words)
    _check(len(ts_in) > 0)
    _check(len(ts_in) == len(ts_out))
    dev = ts_in[0].get_device()
    caps = []
    for ti, to in zip(ts_in, ts_out):
        _check(ti.is_cuda and ti.get_device() == dev and ti.is_contiguous())
        _check(to.is_cuda and to.get_device() == dev and to.is_contiguous())
        _check(ti.dtype == torch.uint8)
        if compress_as_float:
            _float_type(to)
        cap = to.numel() if compress_as_float else to.numel() * to.element_size()
        _check(cap <= _U32_MAX)
        caps.append(cap)
    n = len(ts_in)
    _validate_status(out_status, out_decompressed_words, n, dev)
    with torch.cuda.device(dev):
        tp, tb = _temp(temp_mem, dev)
        used = C.c_size_t(0)
        err = C.c_int32(-1)
        if compress_as_float:
            rc = lib().dgpu_float_decompress_bounded(
                tp, tb, C.byref(used), _float_type(ts_out[0]), prob_bits, int(checksum), n,
                _ptr_array(ts_in), _in_bytes(ts_in), _ptr_array(ts_out), _u32_array(caps), _ptr(out_status),
                _ptr(out_decompressed_words), _stream(), C.byref(err))
        else:
            rc = lib().dgpu_ans_decode_batch_pointer_bounded(
                tp, tb, C.byref(used), prob_bits, int(checksum), n, _ptr_array(ts_in), _in_bytes(ts_in),
                _ptr_array(ts_out), _u32_array(caps), _ptr(out_status),
                _ptr(out_decompressed_words), _stream(), C.byref(err))
        _raise_checksum(rc, compress_as_float)
    return int(used.value)


def decompress_data_split_size(compress_as_float, ts_in, t_out, t_out_split_sizes, checksum=False,
                               temp_mem=None, out_status=None, out_decompressed_words=None,
                               prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:679-816 -> temp bytes used."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.decompress_data_split_size(compress_as_float, ts_in, t_out, t_out_split_sizes, checksum, temp_mem, out_status,
                                               out_decompressed_words)
    _check(len(ts_in) > 0)
    dev = ts_in[0].get_device()
    ss = t_out_split_sizes
    _check(ss.is_contiguous() and not ss.is_cuda and ss.dtype == torch.int32)
    split = [int(v) for v in ss.tolist()]
    n = len(split)
    _check(n == len(ts_in))
    for ti, s in zip(ts_in, split):
        _check(ti.is_cuda and ti.get_device() == dev and ti.is_contiguous() and ti.dtype == torch.uint8)
        _check(s > 0)
    _check(t_out.is_cuda and t_out.get_device() == dev and t_out.is_contiguous())
    if compress_as_float:
        _float_type(t_out)
    _validate_status(out_status, out_decompressed_words, n, dev)
    with torch.cuda.device(dev):
        tp, tb = _temp(temp_mem, dev)
        used = C.c_size_t(0)
        err = C.c_int32(-1)
        if compress_as_float:
            rc = lib().dgpu_float_decompress_split_size_bounded(
                tp, tb, C.byref(used), _float_type(t_out), prob_bits, int(checksum), n,
                _ptr_array(ts_in), _in_bytes(ts_in), _ptr(t_out), _u32_array(split), _ptr(out_status),
                _ptr(out_decompressed_words), _stream(), C.byref(err))
        else:
            rc = lib().dgpu_ans_decode_batch_split_size_bounded(
                tp, tb, C.byref(used), prob_bits, int(checksum), n, _ptr_array(ts_in), _in_bytes(ts_in), _ptr(t_out),
                _u32_array(split), _ptr(out_status), _ptr(out_decompressed_words), _stream(),
                C.byref(err))
        _raise_checksum(rc, compress_as_float)
    return int(used.value)


def decompress_data_simple(compress_as_float, ts_in, checksum=False, temp_mem=67108864,
                           prob_bits=K_DEFAULT_PRECISION):
    """DietGpu.cpp:818-911 -> list of decompressed tensors (sizes/dtypes read from the headers)."""
    fast = _fast_ops(prob_bits)
    if fast is not None:
        return fast.decompress_data_simple(compress_as_float, ts_in, checksum, temp_mem)
    _check(len(ts_in) > 0)
    dev = ts_in[0].get_device()
    device = ts_in[0].device
    n = len(ts_in)
    for t in ts_in:
        _check(t.is_cuda and t.get_device() == dev and t.is_contiguous())
    with torch.cuda.device(dev):
        scratch = None
        if temp_mem is not None and temp_mem >= 256:  # kSDMAlignment, DietGpu.cpp:831-834
            scratch = torch.empty((temp_mem,), dtype=torch.uint8, device=device)
        tp, tb = _temp(scratch, dev)
        sizes = torch.empty((n,), dtype=torch.int32, device=device)
        types = torch.zeros((n,), dtype=torch.int32, device=device)
        if compress_as_float:
            check(lib().dgpu_float_get_compressed_info(
                tp, tb, _ptr_array(ts_in), n, _ptr(sizes), _ptr(types), None, _stream()))
        else:
            check(lib().dgpu_ans_get_compressed_info(
                tp, tb, _ptr_array(ts_in), n, _ptr(sizes), None, _stream()))
        hs, ht = sizes.tolist(), types.tolist()
        outs = []
        for i in range(n):
            if compress_as_float:
                _check(ht[i] == ht[0])  # must be a consistent dtype
                outs.append(torch.empty((hs[i],), dtype=_FT_TO_DTYPE[ht[i]], device=device))
            else:
                outs.append(torch.empty((hs[i],), dtype=torch.uint8, device=device))
    decompress_data(compress_as_float, ts_in, outs, checksum, scratch, None, None, prob_bits=prob_bits)
    return outs
