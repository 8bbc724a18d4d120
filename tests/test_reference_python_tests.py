"""The reference's OWN Python tests, executed as they are (SURVEY.md section 8(f)-1's done-criterion).

`dietgpu/ans_test.py` and `dietgpu/float_test.py` of a facebookresearch/dietgpu checkout are run unmodified with
`torch.ops.load_library` pointed at this repository's libdietgpu_torch.so (their only tie to the reference's build is
the line `torch.ops.load_library("//dietgpu:dietgpu")`).  They need a GPU AND the two files: from a checkout
(/root/reference, or $DIETGPU_REFERENCE_ROOT), or -- the GPU boxes of this project have no checkout -- from
oracle/_ref/reference_python_tests/, where `__graft_entry__.build()` stages them in the build container next to
libdietgpu_ref.so.  oracle/_ref/ is git-ignored (reference-derived test artefacts are never committed) but travels to
the GPU box with the built libraries.  Without a GPU the tests skip; tests/test_torch_ops.py is the restatement that
covers the same calls with the oracle as the judge.  `python tools/run_reference_python_tests.py <checkout>` does the
same outside pytest."""
import os
import runpy
import sys
import unittest

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIETGPU_REFERENCE_ROOT", "/root/reference")
STAGED = os.path.join(ROOT, "oracle", "_ref", "reference_python_tests")
NAMES = ("ans_test.py", "float_test.py")


def _files():
    for d in (os.path.join(REF, "dietgpu"), STAGED):
        fs = [os.path.join(d, f) for f in NAMES]
        if all(os.path.exists(f) for f in fs):
            return fs
    return [os.path.join(STAGED, f) for f in NAMES]


FILES = _files()

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU"),
    pytest.mark.skipif(not all(os.path.exists(f) for f in FILES),
                       reason=f"neither a facebookresearch/dietgpu checkout at {REF} nor staged copies in {STAGED}"),
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
