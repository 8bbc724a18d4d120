"""Bytes of an archive the reference leaves INDETERMINATE (they come from an
uninitialised stack struct): the oracle and the HIP path write zero there, so
comparisons with archives produced by the reference itself blank them first.

  ANS header  (GpuANSUtils.cuh:199-209): options bits 5..31 (setProbBits / setUseChecksum only
              touch bits 0..4 of an uninitialised word, :134-147), `checksum` when the checksum is off
              (GpuANSEncode.cuh:557-559), `unused0`, `unused1`.
  Float header (GpuFloatUtils.cuh:61-72): options bits 5..31, `checksum` when off.
  Block padding (GpuANSEncode.cuh:618-627): the coalesce kernel copies roundUp(words, 8) words of every block out
              of the per-block scratch buffer, i.e. up to 7 words the encoder never wrote.  oracle/_ref runs the
              reference on zero-filled memory, so they are zero as long as the scratch region was not used for
              something else earlier in the same call; in large batches the reference's temp stack hands the
              encoder memory an earlier temporary has written (first seen: row 0 of an 8 x 1 MiB batch, word
              1358 of block 0), so the pad words are blanked too.
Everything else -- pdf table, warp states, block table, the blocks' own words, non-compressed planes -- is compared
as is.
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
    # pad words of every block
    nb = int.from_bytes(a[off + 4 : off + 8].tobytes(), "little")
    table = off + 32 + 512 + 128 * nb
    data = table + 8 * ((nb + 1) // 2 * 2)
    if nb and data <= a.size:
        bw = a[table : table + 8 * nb].view(np.uint32).reshape(nb, 2)
        words = (bw[:, 0] & 0xFFFF).astype(np.int64)
        start = bw[:, 1].astype(np.int64)
        for w, st in zip(words, start):
            lo, hi = data + 2 * (st + w), data + 2 * (st + (w + 7) // 8 * 8)
            if hi <= a.size:
                a[lo:hi] = 0
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
