"""GPU parity of the one-kernel float compress (k_float_compress_fused, kernels_fused.h): batches of equally sized
tensors of whole 32 Ki-word tiles take it; its archives must equal the oracle's -- and the two-kernel path's -- byte for
byte, whatever is resident and whoever counts a tile.  Same helpers and both tensor surfaces as test_gpu_parity.py."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

import oracle as O
import refgen
from test_gpu_parity import DEV, FT_DTYPE, dg, tensor_to_words, words_to_tensor  # noqa: F401

pytestmark = pytest.mark.gpu
TILE = 8 * 4096


def kernels_launched(L, fn):
    """Names of the library's kernels launched by fn() (the library's own event profile)."""
    L.dgpu_prof_reset()
    L.dgpu_prof_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        L.dgpu_prof_enable(0)
    buf = C.create_string_buffer(1 << 14)
    n = L.dgpu_prof_summary(buf, len(buf))
    L.dgpu_prof_reset()
    return out, set(json.loads(buf.value.decode())) if n > 0 else set()


def batch_words(ft, B, tiles, seed, incompressible_every=0):
    rng = np.random.default_rng(seed)
    dt = np.uint32 if ft == O.FLOAT32 else np.uint16
    ws = []
    for b in range(B):
        if incompressible_every and b % incompressible_every == 0:
            ws.append(rng.integers(0, 1 << (8 * dt().itemsize), tiles * TILE, dtype=np.uint64).astype(dt))
        else:
            w = refgen.generate_floats(ft, tiles * TILE)
            ws.append(np.ascontiguousarray(np.roll(w, b * 977)))
    return ws


def compress_and_check(dg, ft, ws, prob_bits, checksum, expect_fused=True):
    L = dg.lib()
    ts = [words_to_tensor(ft, w) for w in ws]
    (comp, sizes, _), names = kernels_launched(L, lambda: dg.compress_data(True, ts, checksum, prob_bits=prob_bits))
    assert ("k_float_compress_fused" in names) == expect_fused, names
    if expect_fused:
        assert "k_ans_encode" not in names and "k_float_histogram" not in names, names
    hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
    rows = []
    for i, w in enumerate(ws):
        want = O.float_compress(ft, w, prob_bits, use_checksum=checksum)
        assert hs[i] == want.size, (i, hs[i], want.size)
        bad = np.nonzero(hc[i, : hs[i]] != want)[0]
        assert bad.size == 0, (i, ft, w.size, bad[:8])
        rows.append(comp[i, : hs[i]].clone())
    outs = [torch.empty_like(t) for t in ts]
    status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, rows, outs, checksum, None, status, prob_bits=prob_bits)
    assert status.cpu().numpy().all()
    assert all((tensor_to_words(ft, o) == w).all() for o, w in zip(outs, ws))
    return rows


@pytest.fixture()
def fused(dg):
    L = dg.lib()
    L.dgpu_debug_set_fused_compress(1, 0)
    yield L
    L.dgpu_debug_set_fused_compress(-1, 0)
    L.dgpu_debug_set_absent_workgroups(0)


@pytest.mark.parametrize("ft,prob_bits", [(O.BFLOAT16, 10), (O.FLOAT16, 11), (O.FLOAT32, 10), (O.BFLOAT16, 9), (O.FLOAT16, 10)])
@pytest.mark.parametrize("B,tiles", [(1, 1), (5, 1), (7, 3), (3, 16)])
def test_fused_archives_equal_the_oracle(dg, fused, ft, prob_bits, B, tiles):
    compress_and_check(dg, ft, batch_words(ft, B, tiles, 100 * B + tiles + ft), prob_bits, checksum=(B % 2 == 1))


def test_fused_elements_of_many_tiles_meet_in_atomic_counters(dg, fused):
    # more than 64 tiles per element: the tiles' counts meet in 256 atomic counters per element instead of in per-tile
    # partial histograms; twice in a row (the counters are zero at rest)
    ws = batch_words(O.BFLOAT16, 2, 70, 7)
    a = compress_and_check(dg, O.BFLOAT16, ws, 10, checksum=False)
    b = compress_and_check(dg, O.BFLOAT16, ws, 10, checksum=True)
    assert len(a) == len(b)


def test_fused_incompressible_tiles_spill(dg, fused):
    # random words: the compressed bytes are incompressible, every block overflows its LDS stage into the spill slots
    for ft in (O.BFLOAT16, O.FLOAT16):
        compress_and_check(dg, ft, batch_words(ft, 6, 2, 11, incompressible_every=2), 10, checksum=False)


@pytest.mark.parametrize("help_after", [1, 0])
def test_fused_with_absent_workgroups(dg, fused, help_after):
    # every third workgroup becomes resident ~0.5 ms late.  help_after 1: the workgroups that wait for an element's
    # table count the tiles whose owners have not shown up (count claims), so the table is made without them; the late
    # owners find their counts claimed.  help_after 0 (the default threshold): they mostly just wait.
    L = fused
    L.dgpu_debug_set_fused_compress(1, help_after)
    L.dgpu_debug_set_absent_workgroups(3)
    try:
        for ft, B, tiles in ((O.BFLOAT16, 9, 4), (O.FLOAT32, 3, 2), (O.BFLOAT16, 1, 70)):
            compress_and_check(dg, ft, batch_words(ft, B, tiles, 31 + B), 10, checksum=False)
    finally:
        L.dgpu_debug_set_absent_workgroups(0)


def test_fused_and_two_kernel_paths_agree_at_full_size(dg):
    # BASELINE config 3's shape (256 x 512 Ki bf16): the same archives from both paths, alternating on one stream (the
    # hand-off words of both are zero at rest), and the round trip
    L = dg.lib()
    g = torch.Generator(device=DEV).manual_seed(1234)
    t = torch.randn((256, 512 * 1024), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    ts = list(t.unbind(0))
    got = []
    try:
        for mode in (1, 0, 1):
            L.dgpu_debug_set_fused_compress(mode, 0)
            (comp, sizes, _), names = kernels_launched(L, lambda: dg.compress_data(True, ts, False))
            assert ("k_float_compress_fused" in names) == (mode == 1), names
            got.append((comp.clone(), sizes.clone()))
    finally:
        L.dgpu_debug_set_fused_compress(-1, 0)
    for comp, sizes in got[1:]:
        assert torch.equal(sizes, got[0][1])
        n = int(sizes.max().item())
        mask = torch.arange(n, device=DEV)[None, :] < sizes[:, None]
        assert torch.equal(comp[:, :n] * mask, got[0][0][:, :n] * mask)
    comp, sizes = got[0]
    rows = [comp[i, : int(s)] for i, s in enumerate(sizes.cpu().tolist())]
    outs = [torch.empty_like(x) for x in ts]
    status = torch.zeros((256,), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, rows, outs, False, None, status)
    assert status.cpu().numpy().all() and all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ts, outs))


def test_batches_the_fused_path_does_not_take(dg, fused):
    # not whole tiles, or sizes that differ, or tensors that are not 16-byte aligned: the two-kernel path, same archives
    ft = O.BFLOAT16
    ws = [refgen.generate_floats(ft, TILE + 100), refgen.generate_floats(ft, TILE + 100)]
    compress_and_check(dg, ft, ws, 10, False, expect_fused=False)
    ws = [refgen.generate_floats(ft, TILE), refgen.generate_floats(ft, 2 * TILE)]
    compress_and_check(dg, ft, ws, 10, False, expect_fused=False)
    L = fused
    base = words_to_tensor(ft, refgen.generate_floats(ft, 2 * TILE + 8))
    ts = [base[1 : 1 + TILE], base[TILE + 1 : 2 * TILE + 1]]  # word 1: 2 bytes past a 16-byte boundary
    (comp, sizes, _), names = kernels_launched(L, lambda: dg.compress_data(True, ts, False))
    assert "k_float_compress_fused" not in names
    for i, t in enumerate(ts):
        want = O.float_compress(ft, tensor_to_words(ft, t), 10)
        n = int(sizes[i].item())
        assert n == want.size and (comp[i, :n].cpu().numpy() == want).all()
