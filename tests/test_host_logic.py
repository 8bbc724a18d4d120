"""Host-side logic that needs no GPU: the build helper and bench.py's launch rules."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_needs_build_lists_existing_sources():
    from dietgpu_amd import build as b

    deps = b._deps()
    assert all(os.path.exists(d) for d in deps), [d for d in deps if not os.path.exists(d)]
    names = {os.path.basename(d) for d in deps}
    assert {"capi.hip", "format.h", "kernels_encode.h", "kernels_decode.h", "kernels_stats.h", "dietgpu_amd.h"} <= names
    assert b.needs_build() in (True, False)  # must not raise whether or not the library exists
    if os.path.exists(b.LIB_PATH):
        assert b.build() == b.LIB_PATH  # no force: returns without compiling when up to date


def _bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=300, env=env)


def test_bench_refuses_fewer_gpus_than_asked():
    import torch

    if torch.cuda.device_count() >= 2:
        return  # a multi-GPU box would really run it
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert p.returncode != 0
    assert "--gpus 2 asked for" in p.stderr and "{" not in p.stdout  # no JSON line with a smaller n_gpus


def test_bench_checks_world_size_against_flag():
    p = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
    p = _bench(["--gpus", "0"], {})
    assert p.returncode != 0


def test_collection_from_the_repo_root_needs_no_gpu():
    """`pytest --collect-only` from the repo root (no path argument) must succeed on a CPU box: nothing outside
    tests/ may look like a test module, and no test module may touch the GPU at import."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "--collect-only", "-q", "-p", "no:cacheprovider"],
                       cwd=root, capture_output=True, text=True, timeout=600,
                       env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "error" not in p.stdout.lower().split("\n")[-2], p.stdout[-500:]
    # and nothing under tools/ (experiment scripts that do GPU work at import) is collected
    assert "tools/" not in p.stdout


def test_abi_version_of_the_library_is_the_headers():
    # what csrc/torch_ops.cpp checks at load and dietgpu_amd.ops before it hands calls to the op library
    import re

    import dietgpu_amd

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "dietgpu_amd.h")) as f:
        want = int(re.search(r"#define DGPU_ABI_VERSION (\d+)u", f.read()).group(1))
    assert dietgpu_amd.lib().dgpu_abi_version() == want


def test_op_library_that_does_not_load_is_not_used(tmp_path, monkeypatch):
    # an op library that cannot be loaded -- built against another ABI version (it refuses itself, torch_ops.cpp), or
    # damaged -- leaves dietgpu_amd.ops on the ctypes route with a warning; file times play no part
    import warnings

    from dietgpu_amd import build, ops

    broken = tmp_path / "libdietgpu_torch.so"
    broken.write_bytes(b"not a shared object")
    monkeypatch.setattr(build, "TORCH_LIB_PATH", str(broken))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ops._load_fast_ops() is False
    assert any("could not be loaded" in str(x.message) for x in w)
    monkeypatch.setattr(build, "TORCH_LIB_PATH", str(tmp_path / "absent.so"))
    assert ops._load_fast_ops() is False


def test_op_library_built_against_another_abi_version_registers_nothing(tmp_path):
    # The real thing: torch_ops.cpp compiled against a header whose DGPU_ABI_VERSION is one ahead of the core library's.
    # Loading it must not end the process (torch.ops.load_library is a dlopen: an exception leaving a static initialiser
    # there is std::terminate), it must register no op implementation, and dietgpu_amd.ops must stay on the ctypes route.
    import re
    import shutil
    import subprocess
    import sys

    import torch

    from dietgpu_amd import build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "include").mkdir()
    (tmp_path / "a" / "csrc").mkdir(parents=True)
    with open(os.path.join(root, "include", "dietgpu_amd.h")) as f:
        header = f.read()
    bumped, n = re.subn(r"#define DGPU_ABI_VERSION (\d+)u", lambda m: f"#define DGPU_ABI_VERSION {int(m.group(1)) + 1}u", header)
    assert n == 1
    (tmp_path / "include" / "dietgpu_amd.h").write_text(bumped)
    shutil.copy(os.path.join(build.CSRC, "torch_ops.cpp"), tmp_path / "a" / "csrc" / "torch_ops.cpp")
    build.build()
    tl = os.path.dirname(torch.__file__)
    stale = tmp_path / "libdietgpu_torch.so"
    subprocess.check_call(
        ["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
         f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{tl}/include",
         f"-I{tl}/include/torch/csrc/api/include", "-I/opt/rocm/include", str(tmp_path / "a" / "csrc" / "torch_ops.cpp"),
         "-o", str(stale), f"-L{build.LIB_DIR}", "-ldietgpu_amd", f"-L{tl}/lib", "-ltorch", "-ltorch_cpu", "-lc10",
         "-lc10_hip", "-ltorch_hip", f"-Wl,-rpath,{build.LIB_DIR}"])
    script = f"""
import sys, warnings
sys.path.insert(0, {root!r})
import torch
from dietgpu_amd import build, ops
build.TORCH_LIB_PATH = {str(stale)!r}
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    got = ops._load_fast_ops()
assert got is False, got
assert any("C ABI version" in str(x.message) for x in w), [str(x.message) for x in w]
assert ops._fast_ops(10) is None and ops._fast_ops(11) is None
try:
    torch.ops.dietgpu.max_any_compressed_size(4096)
except (RuntimeError, NotImplementedError):
    pass
else:
    raise AssertionError("an op of the mismatching library has an implementation")
print("alive")
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "alive" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
