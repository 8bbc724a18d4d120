"""The reference's OWN Python tests, executed as they are (SURVEY.md section 8(f)-1's done-criterion).

`dietgpu/ans_test.py` and `dietgpu/float_test.py` of a facebookresearch/dietgpu checkout are run unmodified with
`torch.ops.load_library` pointed at this repository's libdietgpu_torch.so (their only tie to the reference's build is
the line `torch.ops.load_library("//dietgpu:dietgpu")`).  They need a GPU AND a checkout: /root/reference, or
$DIETGPU_REFERENCE_ROOT.  The build container has the checkout and no GPU, the GPU boxes of this project have a GPU
and no checkout (reference sources may not be copied into this repository), so in this project's own runs the test
skips in both places -- tests/test_torch_ops.py is the restatement that does run; this file is what a maintainer with
both at hand runs (`python tools/run_reference_python_tests.py <checkout>` does the same outside pytest)."""
import os
import runpy
import sys
import unittest

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIETGPU_REFERENCE_ROOT", "/root/reference")
FILES = [os.path.join(REF, "dietgpu", f) for f in ("ans_test.py", "float_test.py")]

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not all(os.path.exists(f) for f in FILES), reason=f"no facebookresearch/dietgpu checkout at {REF}"),
]


def run_reference_test_file(path):
    """-> unittest result of every TestCase the file defines, with load_library redirected to our library."""
    sys.path.insert(0, ROOT)
    import dietgpu_amd

    dietgpu_amd.load_torch_ops()  # registers torch.ops.dietgpu.* from dietgpu_amd/lib/libdietgpu_torch.so
    real = torch.ops.load_library
    torch.ops.load_library = lambda name: None if name == "//dietgpu:dietgpu" else real(name)
    try:
        ns = runpy.run_path(path, run_name="reference_test")
    finally:
        torch.ops.load_library = real
    suite = unittest.TestSuite()
    for obj in ns.values():
        if isinstance(obj, type) and issubclass(obj, unittest.TestCase):
            suite.addTests(unittest.defaultTestLoader.loadTestsFromTestCase(obj))
    assert suite.countTestCases() > 0, path
    return unittest.TextTestRunner(verbosity=2).run(suite)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_reference_python_test_file_runs_unmodified(path):
    result = run_reference_test_file(path)
    assert result.wasSuccessful(), (result.failures, result.errors)
