"""The dietgpu:: C++ API mirror (include/dietgpu_amd/*.h over the C ABI)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "api_roundtrip.cpp")


def test_headers_are_plain_host_cxx17():
    # compiles with g++ alone: no device code, no torch types behind the boundary
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), SRC])
    hdr = open(os.path.join(ROOT, "include", "dietgpu_amd.h")).read()
    assert "torch" not in hdr.lower().replace("pytorch", "") and "at::" not in hdr


@pytest.mark.gpu
def test_cpp_api_roundtrip(tmp_path):
    exe = str(tmp_path / "api_roundtrip")
    lib = os.path.join(ROOT, "dietgpu_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + lib, "-ldietgpu_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "api_roundtrip: OK" in out.stdout
