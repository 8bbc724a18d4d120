"""Properties of the compiled gfx950 code that the measurements depend on (no GPU needed: hipcc cross-compiles).

* No FLAT loads / stores in any kernel.  A batch address that reaches a load as a GENERIC pointer (an integer from a
  kernel argument, or a pointer carried across a loop edge) compiles to flat_load / flat_store; FLAT operations also
  tick the LDS counter, and with one of them pending the compiler turns every counted `s_waitcnt vmcnt(N)` of the rANS
  row loops into `vmcnt(0)` -- measured 183 -> 291 us on a decoder build that carried pointers through its pipeline
  (docs/HISTORY.md section 4.4).  `BatchView::ptr` and the kernels materialise global (address space 1) pointers instead.
* No kernel spills to scratch (spill code waits for the loads it parks, which serialises prefetches: docs/HISTORY.md section 3).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc is not on PATH")
    out = tmp_path_factory.mktemp("isa") / "capi.s"
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                        os.path.join(ROOT, "dietgpu_amd", "csrc", "capi.hip")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return out.read_text()


def test_no_flat_memory_instructions(isa):
    kernel, offenders = None, {}
    for line in isa.splitlines():
        m = re.match(r"^(_ZN4dgpu\w+):", line)
        if m:
            kernel = m.group(1)
        elif kernel and re.match(r"\s+flat_(load|store|atomic)", line):
            offenders.setdefault(kernel, []).append(line.strip())
    assert not offenders, {k: v[:3] for k, v in offenders.items()}


def test_no_kernel_uses_scratch(isa):
    kernel, sizes = None, {}
    for line in isa.splitlines():
        m = re.match(r"\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kernel = m.group(1)
        m = re.match(r"\s+\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
        if m and kernel:
            sizes[kernel] = int(m.group(1))
    assert len(sizes) > 50  # every instantiation of the coders
    assert not {k: v for k, v in sizes.items() if v}, "a kernel spills registers to scratch"
