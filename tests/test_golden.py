"""Golden fixtures (tests/golden/): archives produced by the REFERENCE ITSELF -- its own sources run on
the CPU SIMT emulation of oracle/ref_shim/ (tests/golden/make_golden.py; indeterminate header bytes
blanked, tests/refmask.py).  The oracle must reproduce them (CPU) and the HIP path must produce the
same bytes (GPU).  /root/reference is not needed to run these."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as O
import refgen
from refmask import mask_ans, mask_float

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = json.load(open(os.path.join(HERE, "golden.json")))
CASES = GOLDEN["cases"]


def _input(c):
    """(is_float, float type, data)"""
    k = c["kind"]
    if k == "ans":
        return False, 0, refgen.generate_symbols(c["n"], c["lam"])
    if k == "float":
        return True, c["float_type"], refgen.generate_floats(c["float_type"], c["n"])
    if k == "config2":
        return False, 0, refgen.zipf_bytes(c["row"] + 1, 1 << 20)[c["row"]]
    if k == "config3":
        return True, O.BFLOAT16, refgen.normal_bf16(c["row"] + 1, 512 * 1024)[c["row"]]
    if k == "config4":
        return True, O.FLOAT16, refgen.sparse_fp16(c["row"] + 1, 512 * 1024)[c["row"]]
    raise ValueError(k)


def _check(c, archive, is_float):
    a = mask_float(archive) if is_float else mask_ans(archive)
    assert a.size == c["size"] and hashlib.sha256(a.tobytes()).hexdigest() == c["sha256"], c


def test_fixtures_come_from_the_reference():
    assert "reference" in GOLDEN["meta"]["source"] and len(CASES) >= 80


def test_oracle_reproduces_golden():
    for c in CASES:
        is_float, ft, x = _input(c)
        a = (O.float_compress(ft, x, c["prob_bits"], use_checksum=c["checksum"]) if is_float
             else O.ans_encode(x, c["prob_bits"], use_checksum=c["checksum"]))
        _check(c, a, is_float)
    a = np.fromfile(os.path.join(HERE, "ans_p10_lam20_n5000.bin"), np.uint8)
    x = refgen.generate_symbols(5000, 20.0)
    assert (a == O.ans_encode(x, 10, use_checksum=True)).all()
    rc, y, _ = O.ans_decode(a, 10)
    assert rc == 0 and (y == x).all()
    a = np.fromfile(os.path.join(HERE, "bf16_p10_n6000.bin"), np.uint8)
    w = refgen.generate_floats(O.BFLOAT16, 6000)
    assert (a == O.float_compress(O.BFLOAT16, w, 10, use_checksum=True)).all()
    rc, y, _ = O.float_decompress(O.BFLOAT16, a, 10)
    assert rc == 0 and (y == w).all()


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    import torch

    import dietgpu_amd as dg

    dt = {1: torch.float16, 2: torch.bfloat16, 3: torch.float32}
    for c in CASES:
        is_float, ft, x = _input(c)
        if not is_float:
            t = torch.from_numpy(x.copy()).cuda()
            comp, sizes, _ = dg.compress_data(False, [t], c["checksum"], prob_bits=c["prob_bits"])
        else:
            view = np.int32 if ft == 3 else np.int16
            t = torch.from_numpy(x.view(view).copy()).cuda().view(dt[ft])
            comp, sizes, _ = dg.compress_data(True, [t], c["checksum"], prob_bits=c["prob_bits"])
        n = int(sizes[0])
        _check(c, comp[0, :n].cpu().numpy(), is_float)
    # the stored reference archives: the HIP encoder reproduces them byte for byte, the HIP decoder reads them
    a = np.fromfile(os.path.join(HERE, "bf16_p10_n6000.bin"), np.uint8)
    w = refgen.generate_floats(O.BFLOAT16, 6000)
    t = torch.from_numpy(w.view(np.int16).copy()).cuda().view(torch.bfloat16)
    comp, sizes, _ = dg.compress_data(True, [t], True)
    assert int(sizes[0]) == a.size and (comp[0, : a.size].cpu().numpy() == a).all()
    out = dg.decompress_data_simple(True, [torch.from_numpy(a).cuda()], True)[0]
    assert (out.view(torch.int16).cpu().numpy().view(np.uint16) == w).all()
    a = np.fromfile(os.path.join(HERE, "ans_p10_lam20_n5000.bin"), np.uint8)
    x = refgen.generate_symbols(5000, 20.0)
    out = dg.decompress_data_simple(False, [torch.from_numpy(a).cuda()], True)[0]
    assert (out.cpu().numpy() == x).all()
