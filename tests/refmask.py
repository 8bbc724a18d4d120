"""Bytes of an archive the reference leaves INDETERMINATE (they come from an
uninitialised stack struct): the oracle and the HIP path write zero there, so
comparisons with archives produced by the reference itself blank them first.

  ANS header  (GpuANSUtils.cuh:199-209): options bits 5..31 (setProbBits / setUseChecksum only
              touch bits 0..4 of an uninitialised word, :134-147), `checksum` when the checksum is off
              (GpuANSEncode.cuh:557-559), `unused0`, `unused1`.
  Float header (GpuFloatUtils.cuh:61-72): options bits 5..31, `checksum` when off.
Everything else -- pdf table, warp states, block table, block data and their padding, non-compressed
planes -- is compared as is (oracle/_ref runs the reference on zero-filled memory).
"""
import numpy as np


def _float_uncomp_data_size(ft, n):
    r16 = (n + 15) // 16 * 16
    if ft == 3:
        return 2 * ((n + 7) // 8 * 8) + r16
    return r16


def mask_ans(a, off=0):
    a = np.array(a, dtype=np.uint8, copy=True)
    opt = int.from_bytes(a[off + 16 : off + 20].tobytes(), "little")
    use_ck = (opt >> 4) & 1
    a[off + 16 : off + 20] = np.frombuffer((opt & 0x1F).to_bytes(4, "little"), np.uint8)
    if not use_ck:
        a[off + 20 : off + 24] = 0
    a[off + 24 : off + 32] = 0
    return a


def mask_float(a):
    a = np.array(a, dtype=np.uint8, copy=True)
    n = int.from_bytes(a[4:8].tobytes(), "little")
    opt = int.from_bytes(a[8:12].tobytes(), "little")
    ft = opt & 0xF
    use_ck = (opt >> 4) & 1
    a[8:12] = np.frombuffer((opt & 0x1F).to_bytes(4, "little"), np.uint8)
    if not use_ck:
        a[12:16] = 0
    return mask_ans(a, 16 + _float_uncomp_data_size(ft, n))
