"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every function include/dietgpu_amd.h declares (no compute calls without a GPU),
and the pure host-side size queries agree with the oracle."""
import ctypes
import os
import re

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dietgpu_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import dietgpu_amd

    names = declared_functions()
    assert len(names) >= 25
    raw = ctypes.CDLL(dietgpu_amd.build.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libdietgpu_amd.so does not export {n}"
    # the ctypes binding covers the same set
    assert set(names) == set(dietgpu_amd.EXPORTED_SYMBOLS)
    dietgpu_amd.lib()


def test_size_queries_match_oracle():
    import dietgpu_amd

    L = dietgpu_amd.lib()
    for n in (0, 1, 4095, 4096, 4097, 1 << 20, 123456789):
        assert L.dgpu_ans_max_compressed_size(n) == O.ans_max_compressed_size(n)
        for ft in (1, 2, 3):
            assert L.dgpu_float_max_compressed_size(ft, n) == O.float_max_compressed_size(ft, n)
    # the reference CHECKs the maximum compressed size against INT32_MAX (GpuANSEncode.cu:22): 419 321 blocks are the
    # largest input it accepts; beyond that the size queries return 0 (the C++ mirrors abort like upstream)
    guard = 419321 * 4096
    assert L.dgpu_ans_max_compressed_size(guard) == O.ans_max_compressed_size(guard) == 557600 + 5120 * 419321
    assert L.dgpu_ans_max_compressed_size(guard) <= 2**31 - 1
    assert L.dgpu_ans_max_compressed_size(guard + 1) == 0 == O.ans_max_compressed_size(guard + 1)
    assert L.dgpu_float_max_compressed_size(2, guard + 1) == 0 and L.dgpu_float_max_compressed_size(2, 1 << 30) > (1 << 31)
    assert L.dgpu_ans_max_compressed_size(1 << 20) == 1868320
    assert L.dgpu_float_max_compressed_size(2, 524288) == 1737264
    # temp memory of the 256 x 1 MiB configs: ~10 MiB (partial histograms, tables, tile
    # descriptors) + for floats the spill slots of the resident encoder workgroups (~70 MiB,
    # independent of the batch size), not the reference's 328 MiB block scratch + 128 MiB
    # exponent plane
    assert L.dgpu_float_compress_temp_bytes(2, 256, 524288) < 96 * 1024 * 1024
    assert L.dgpu_float_compress_temp_bytes(2, 1, 1 << 30) < 96 * 1024 * 1024 + (1 << 30) // 1000
    assert L.dgpu_ans_encode_temp_bytes(256, 1 << 20) < 12 * 1024 * 1024


def test_ops_argument_validation_without_gpu():
    import torch

    import dietgpu_amd as dg
    import pytest

    with pytest.raises(RuntimeError):
        dg.compress_data(True, [])
    with pytest.raises(RuntimeError):  # CPU tensors are rejected like TORCH_CHECK(t.device().type() == kCUDA)
        dg.compress_data(False, [torch.zeros(16, dtype=torch.uint8)])
    assert dg.max_any_compressed_size(1 << 20) == 1868320
    assert dg.max_float_compressed_size(torch.empty(0, dtype=torch.bfloat16), 524288) == 1737264


def _read(*parts):
    return open(os.path.join(ROOT, *parts)).read()


def test_every_c_abi_function_has_a_test_caller():
    """Every dgpu_* function of include/dietgpu_amd.h is CALLED by a test: directly (tests/*.py, tests/cpp/*.cpp) or
    through one of the two tensor surfaces the GPU suite runs on (dietgpu_amd/ops.py, csrc/torch_ops.cpp) /
    dietgpu_amd/distributed.py.  A function that only the C++ mirror headers mention does not count (the mirror's own
    functions are checked below)."""
    direct = "".join(_read("tests", f) for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py") and f != "test_cabi_symbols.py")
    direct += "".join(_read("tests", "cpp", f) for f in os.listdir(os.path.join(ROOT, "tests", "cpp")))
    surfaces = _read("dietgpu_amd", "ops.py") + _read("dietgpu_amd", "csrc", "torch_ops.cpp") + _read("dietgpu_amd", "distributed.py")
    called = lambda name, text: re.search(r"\b%s\s*\(" % re.escape(name), text) is not None
    missing, only_indirect = [], []
    for n in declared_functions():
        if called(n, direct):
            continue
        (only_indirect if called(n, surfaces) else missing).append(n)
    assert not missing, f"C-ABI functions no test reaches: {missing}"
    # The five the round-3 review found with no caller at all now have DIRECT ones (tests/test_gpu_cabi.py):
    for n in ("dgpu_ans_decode_batch_split_size", "dgpu_float_decompress_split_size", "dgpu_ans_get_compressed_info_device",
              "dgpu_float_get_compressed_info_device", "dgpu_ans_decode_batch_pointer", "dgpu_float_decompress"):
        assert called(n, direct), n
    # ... and `histogram_dev` is passed as a real pointer to all three encode entry points
    cabi = _read("tests", "test_gpu_cabi.py")
    for n in ("dgpu_ans_encode_batch_pointer", "dgpu_ans_encode_batch_stride", "dgpu_ans_encode_batch_split_size"):
        assert re.search(r"%s\([^;]*hist\.data_ptr" % n, cabi, flags=re.S), n


def test_every_function_of_the_cpp_mirror_is_called_by_the_cpp_test():
    """include/dietgpu_amd/Gpu{ANS,Float}Codec.h re-create the dietgpu:: signatures; tests/cpp/api_roundtrip.cpp calls
    every one of them (stride, pointer, split-size, info, device-info; encode with a caller's histogram)."""
    cpp = _read("tests", "cpp", "api_roundtrip.cpp")
    names = []
    for hdr in ("GpuANSCodec.h", "GpuFloatCodec.h"):
        text = _read("include", "dietgpu_amd", hdr)
        text = text.split("}  // namespace detail")[-1]  # the public functions follow the detail namespace
        names += re.findall(r"^inline\s+[\w:]+\s+(\w+)\s*\(", text, flags=re.M)
    assert len(names) >= 14, names
    missing = [n for n in names if not re.search(r"\b%s\s*\(" % n, cpp)]
    assert not missing, missing
    assert re.search(r"ansEncodeBatch\w+\([^;]*hist", cpp), "no C++ call passes histogram_dev"
