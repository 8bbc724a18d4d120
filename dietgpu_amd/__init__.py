"""dietgpu_amd: MI355X-native batched rANS + float codec behind dietgpu's API.

`dietgpu_amd.ops` mirrors `torch.ops.dietgpu.*`; the product path is
libdietgpu_amd.so (hand-written HIP for gfx950) reached through the C ABI in
include/dietgpu_amd.h.  There is no CPU fallback and nothing here imports
oracle/.
"""
from . import build  # noqa: F401
from . import ops  # noqa: F401
from ._lib import DietGpuError, EXPORTED_SYMBOLS, lib  # noqa: F401
from .ops import (  # noqa: F401
    compress_data,
    compress_data_simple,
    compress_data_split_size,
    decompress_data,
    decompress_data_simple,
    decompress_data_split_size,
    max_any_compressed_output_size,
    max_any_compressed_size,
    max_float_compressed_output_size,
    max_float_compressed_size,
    prefer_torch_ops,
)


def load_torch_ops():
    """Registers torch.ops.dietgpu.* (the reference's op names and schemas,
    dietgpu/DietGpu.cpp:915-972) from the in-tree C++ extension."""
    import os

    import torch

    from .build import TORCH_LIB_PATH

    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError(f"{TORCH_LIB_PATH} is missing: run `python -m dietgpu_amd.build`")
    lib()  # libdietgpu_amd.so first (the op library links against it)
    torch.ops.load_library(TORCH_LIB_PATH)
    return torch.ops.dietgpu
