"""Builds libdietgpu_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
# DGPU_LIB: developer override used to A/B experimental builds of the same sources
LIB_PATH = os.environ.get("DGPU_LIB") or os.path.join(LIB_DIR, "libdietgpu_amd.so")

_SOURCES = ["capi.hip"]


def _deps():
    """Everything the library is compiled from: csrc/*.h, csrc/*.hip and the public C header."""
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    return [os.path.join(CSRC, f) for f in names] + [os.path.join(_HERE, "..", "include", "dietgpu_amd.h")]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in _SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


TORCH_LIB_PATH = os.path.join(LIB_DIR, "libdietgpu_torch.so")


def build_torch_ops(force=False, verbose=False):
    """torch.ops.dietgpu.* (csrc/torch_ops.cpp): plain host C++ against PyTorch-ROCm and the C ABI."""
    src = os.path.join(CSRC, "torch_ops.cpp")
    if (not force and os.path.exists(TORCH_LIB_PATH)
            and os.path.getmtime(TORCH_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(LIB_PATH))):
        return TORCH_LIB_PATH
    import torch

    tl = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", f"-I{tl}/include", f"-I{tl}/include/torch/csrc/api/include",
           "-I/opt/rocm/include", src, "-o", TORCH_LIB_PATH, f"-L{LIB_DIR}", "-ldietgpu_amd", f"-L{tl}/lib",
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_torch_ops(force=True, verbose=True)
