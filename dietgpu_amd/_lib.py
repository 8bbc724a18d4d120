"""ctypes binding of the C ABI declared in include/dietgpu_amd.h.

There is no CPU fallback: if libdietgpu_amd.so is missing the import fails
loudly (build it with `python -m dietgpu_amd.build`).
"""
import ctypes as C
import os

from .build import LIB_PATH

_lib = None

u32, i32, sz, vp = C.c_uint32, C.c_int, C.c_size_t, C.c_void_p
_SIGS = {
    "dgpu_version": (C.c_char_p, []),
    "dgpu_abi_version": (u32, []),
    "dgpu_last_error": (C.c_char_p, []),
    "dgpu_last_checksum_mismatches": (C.c_uint32, [C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]),
    "dgpu_ans_max_compressed_size": (u32, [u32]),
    "dgpu_float_max_compressed_size": (u32, [u32, u32]),
    "dgpu_ans_encode_temp_bytes": (sz, [u32, u32]),
    "dgpu_ans_decode_temp_bytes": (sz, [u32, u32, i32]),
    "dgpu_float_compress_temp_bytes": (sz, [u32, u32, u32]),
    "dgpu_float_decompress_temp_bytes": (sz, [u32, u32, u32, i32]),
    "dgpu_ans_encode_batch_stride": (i32, [vp, sz, vp, i32, i32, u32, vp, u32, u32, vp, vp, u32, vp, vp]),
    "dgpu_ans_encode_batch_pointer": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp]),
    "dgpu_ans_encode_batch_split_size": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, u32, vp, vp]),
    "dgpu_ans_decode_batch_stride": (i32, [vp, sz, vp, i32, i32, u32, vp, u32, vp, u32, u32, vp, vp, vp, vp]),
    "dgpu_ans_decode_batch_pointer": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_ans_decode_batch_split_size": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_ans_get_compressed_info": (i32, [vp, sz, vp, u32, vp, vp, vp]),
    "dgpu_ans_get_compressed_info_device": (i32, [vp, u32, vp, vp, vp]),
    "dgpu_float_compress": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, vp, vp]),
    "dgpu_float_compress_split_size": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, u32, vp, vp]),
    "dgpu_float_decompress": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_float_decompress_split_size": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_ans_decode_batch_pointer_bounded": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_ans_decode_batch_split_size_bounded": (i32, [vp, sz, vp, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_float_decompress_bounded": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_float_decompress_split_size_bounded": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgpu_float_compress_stride_capped": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, u32, u32, vp, u32, u32, vp, vp]),
    "dgpu_float_decompress_stride_bounded": (i32, [vp, sz, vp, u32, i32, i32, u32, vp, u32, u32, vp, u32, u32, vp, vp, vp, vp]),
    "dgpu_float_get_compressed_info": (i32, [vp, sz, vp, u32, vp, vp, vp, vp]),
    "dgpu_float_get_compressed_info_device": (i32, [vp, u32, vp, vp, vp, vp]),
    "dgpu_ans_histogram_batch_stride": (i32, [u32, vp, u32, u32, vp, vp]),
    "dgpu_ans_calc_weights": (i32, [u32, i32, vp, u32, vp, vp, vp]),
    "dgpu_release_stream_state": (i32, [vp]),
    "dgpu_release_all_stream_state": (i32, []),
    "dgpu_debug_stream_state_count": (u32, []),
    "dgpu_prof_enable": (None, [i32]),
    "dgpu_prof_reset": (None, []),
    "dgpu_prof_summary": (i32, [C.c_char_p, sz]),
    "dgpu_debug_set_absent_workgroups": (None, [u32]),
    "dgpu_debug_set_encoder_dispatch": (None, [i32]),
    "dgpu_debug_set_decoder_order": (None, [i32]),
    "dgpu_debug_set_param_cache": (None, [i32]),
    "dgpu_debug_set_work_lists": (None, [i32]),
    "dgpu_debug_set_size_classes": (None, [i32]),
    "dgpu_set_histogram_load_policy": (None, [i32]),
    "dgpu_release_graph_state": (i32, []),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m dietgpu_amd.build`). There is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("DGPU_LIB") and not hasattr(L, name):
                continue  # A/B against a library built from an earlier revision (tools/ab.sh): entry points may be newer
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class DietGpuError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().dgpu_last_error().decode()
        raise DietGpuError(f"dietgpu_amd error {rc}: {msg}")
