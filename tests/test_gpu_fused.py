"""The fused histogram + encode kernel (k_ans_encode_fused: ONE read of the input)
against the oracle, byte for byte, and against the two-kernel path.  Uniform batches of whole tiles
take the fused path when it is switched on (it is opt-in: DESIGN.md section 4.3 has the measurements);
dgpu_debug_set_fused(0) forces the two-kernel path for the same call."""
import numpy as np
import pytest
import torch

import oracle as O
import refgen
from test_gpu_parity import DEV, dg, gpu_ans_decode, tensor_to_words, to_dev_bytes, words_to_tensor  # noqa: F401

pytestmark = pytest.mark.gpu
TILE = 8 * 4096


@pytest.fixture(autouse=True)
def fused_on(dg):
    # the kernel is only in builds made with -DDGPU_WITH_FUSED=1 (python tools/build_variant.py fused
    # -DDGPU_WITH_FUSED=1; run with DGPU_LIB=dietgpu_amd/lib/v_fused.so): the default library does not carry it
    if not dg.lib().dgpu_has_fused():
        pytest.skip("this build has no k_ans_encode_fused (DGPU_WITH_FUSED=0, the default)")
    # ... and there it is opt-in (DGPU_FUSED=1 / dgpu_debug_set_fused(1)): these tests force it
    dg.lib().dgpu_debug_set_fused(1)
    yield
    dg.lib().dgpu_debug_set_fused(-1)


def _float_rows(ft, B, n, seed, incompressible=False):
    rng = np.random.default_rng(seed)
    if incompressible:
        bits = 32 if ft == O.FLOAT32 else 16
        return rng.integers(0, 1 << bits, (B, n), dtype=np.uint64).astype(np.uint32 if bits == 32 else np.uint16)
    f = rng.standard_normal((B, n), dtype=np.float32) * np.exp(rng.standard_normal((B, 1), dtype=np.float32) * 3)
    if ft == O.FLOAT16:
        return f.astype(np.float16).view(np.uint16)
    if ft == O.BFLOAT16:
        return refgen.f32_to_bf16_rne(f)
    return f.view(np.uint32).copy()


def _encode_float(dg, ft, rows, prob_bits, checksum=False):
    ts = [words_to_tensor(ft, r) for r in rows]
    comp, sizes, _ = dg.compress_data(True, ts, checksum, prob_bits=prob_bits)
    return comp, sizes, ts


def _compare_float(dg, ft, rows, prob_bits, checksum=False):
    comp, sizes, ts = _encode_float(dg, ft, rows, prob_bits, checksum)
    want, wsz = O.float_compress_batch(ft, rows, prob_bits, threads=16) if not checksum else (None, None)
    hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
    for i in range(rows.shape[0]):
        w = want[i, : wsz[i]] if want is not None else O.float_compress(ft, rows[i], prob_bits, use_checksum=True)
        assert hs[i] == w.size, (i, hs[i], w.size)
        bad = np.nonzero(hc[i, : hs[i]] != w)[0]
        assert bad.size == 0, f"row {i}: first differing bytes {bad[:6]} of {w.size}"
    outs = [torch.empty_like(t) for t in ts]
    status = torch.zeros((len(ts),), dtype=torch.uint8, device=DEV)
    dg.decompress_data(True, [comp[i, : hs[i]] for i in range(len(ts))], outs, checksum, None, status, None,
                       prob_bits=prob_bits)
    assert status.cpu().numpy().all()
    for o, r in zip(outs, rows):
        assert (tensor_to_words(ft, o) == r).all()
    return hc, hs


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_fused_float_matches_oracle(dg, ft, prob_bits):
    for B, T in ((1, 1), (3, 2), (8, 4), (9, 5), (40, 16), (2, 32)):
        _compare_float(dg, ft, _float_rows(ft, B, T * TILE, 100 * B + T), prob_bits)


@pytest.mark.parametrize("prob_bits", [9, 10, 11])
def test_fused_raw_matches_oracle(dg, prob_bits):
    rng = np.random.default_rng(prob_bits)
    for B, T, lam in ((1, 1, 5.0), (5, 3, 20.0), (16, 8, 60.0), (33, 32, 200.0)):
        xs = (rng.exponential(lam, (B, T * TILE)) % 256).astype(np.uint8)
        for ck in (False, True):
            ts = [to_dev_bytes(x) for x in xs]
            comp, sizes, _ = dg.compress_data(False, ts, ck, prob_bits=prob_bits)
            hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
            for i in range(B):
                w = O.ans_encode(xs[i], prob_bits, use_checksum=ck)
                assert hs[i] == w.size and not (hc[i, : hs[i]] != w).any(), (B, T, i, ck)
            outs, status, _ = gpu_ans_decode(dg, [hc[i, : hs[i]] for i in range(B)], [T * TILE] * B, prob_bits, ck)
            assert status.all() and all((o == x).all() for o, x in zip(outs, xs))


@pytest.mark.parametrize("ft", [O.FLOAT16, O.BFLOAT16, O.FLOAT32])
def test_fused_incompressible_spills(dg, ft):
    # random bit patterns: the stage spills to temp memory; P = 11 blocks exceed even the reference's 5120-byte bound
    for P in (9, 11):
        _compare_float(dg, ft, _float_rows(ft, 6, 3 * TILE, 5, incompressible=True), P)


def test_fused_raw_incompressible(dg):
    rng = np.random.default_rng(3)
    xs = rng.integers(0, 256, (5, 4 * TILE), dtype=np.uint8)
    ts = [to_dev_bytes(x) for x in xs]
    for P in (9, 10, 11):
        comp, sizes, _ = dg.compress_data(False, ts, False, prob_bits=P)
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        for i in range(5):
            w = O.ans_encode(xs[i], P)
            assert hs[i] == w.size and not (hc[i, : hs[i]] != w).any(), (P, i)


def test_fused_equals_two_kernel_path_and_checksums(dg):
    L = dg.lib()
    rows = _float_rows(O.BFLOAT16, 24, 6 * TILE, 11)
    try:
        L.dgpu_debug_set_fused(0)
        c0, s0, _ = _encode_float(dg, O.BFLOAT16, rows, 10, True)
        L.dgpu_debug_set_fused(1)
        c1, s1, _ = _encode_float(dg, O.BFLOAT16, rows, 10, True)
    finally:
        L.dgpu_debug_set_fused(1)
    assert torch.equal(s0, s1)
    for i, n in enumerate(s0.cpu().tolist()):
        assert torch.equal(c0[i, :n], c1[i, :n]), i
    _compare_float(dg, O.BFLOAT16, rows[:4], 10, checksum=True)


@pytest.mark.parametrize("modulo", [2, 3, 7])
def test_fused_with_late_workgroups(dg, modulo):
    L = dg.lib()
    rows = _float_rows(O.BFLOAT16, 37, 16 * TILE, modulo)
    L.dgpu_debug_set_absent_workgroups(modulo)
    try:
        _compare_float(dg, O.BFLOAT16, rows, 10)
    finally:
        L.dgpu_debug_set_absent_workgroups(0)


def test_fused_back_to_back_calls_and_streams(dg):
    # library-owned state (tickets, arrival counters, ready epochs) across calls, batch shapes and streams
    shapes = [(7, 2), (64, 16), (3, 32), (64, 16), (1, 1), (20, 9)]
    data = [_float_rows(O.BFLOAT16, B, T * TILE, 7 * B + T) for B, T in shapes]
    want = [O.float_compress_batch(O.BFLOAT16, r, 10, threads=16) for r in data]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    results = []
    for rep in range(3):
        for k, rows in enumerate(data):
            st = streams[(rep + k) % 2]
            with torch.cuda.stream(st):
                comp, sizes, _ = _encode_float(dg, O.BFLOAT16, rows, 10)
                results.append((k, comp, sizes))
    torch.cuda.synchronize()
    for k, comp, sizes in results:
        hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
        w, wsz = want[k]
        assert (hs == wsz).all(), k
        for i in range(hs.size):
            assert not (hc[i, : hs[i]] != w[i, : hs[i]]).any(), (k, i)


def test_ragged_batches_take_the_two_kernel_path(dg):
    # not eligible (sizes differ / not whole tiles / unaligned): same results through the classic kernels
    rng = np.random.default_rng(1)
    ws = [refgen.generate_floats(O.BFLOAT16, n) for n in (TILE, TILE + 1, 3 * TILE, 17)]
    ts = [words_to_tensor(O.BFLOAT16, w) for w in ws]
    comp, sizes, _ = dg.compress_data(True, ts, False)
    hs, hc = sizes.cpu().numpy(), comp.cpu().numpy()
    for i, w in enumerate(ws):
        want = O.float_compress(O.BFLOAT16, w, 10)
        assert hs[i] == want.size and not (hc[i, : hs[i]] != want).any()
