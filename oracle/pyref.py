"""Second, independent restatement of the reference algorithm in pure Python.

TEST INFRASTRUCTURE ONLY (see oracle/dietgpu_oracle.h).  Slow: use on inputs of
a few KiB.  Written separately from dietgpu_oracle.c (different structure: it
simulates the 32 lanes with explicit ballot / prefix-popcount exactly as the
kernels do) so that two restatements must agree byte-for-byte before either is
trusted.  Citations are into /root/reference/dietgpu/.
"""
import struct

import numpy as np

BLOCK = 4096
START = 1 << 15


def normalize(counts, total, prob_bits):
    """ans/GpuANSStatistics.cuh:178-367. Returns list of (pdf, cdf, magic, shift)."""
    if total == 0:
        return [(0, 0, 0, 0)] * 256
    w = 1 << prob_bits
    q = []
    for c in counts:
        # :215 fp32 divide then fp32 multiply then truncate
        r = np.float32(c) / np.float32(total)
        v = int(np.float32(w) * r)
        if c > 0 and v == 0:  # :218
            v = 1
        q.append(v)
    keys = sorted(((q[s] << 16) | s for s in range(256)), reverse=True)  # :234-241
    sym = [k & 0xFFFF for k in keys]
    qq = [k >> 16 for k in keys]
    diff = w - sum(q)
    if diff > 0:  # :258-274
        while diff > 0:
            it = min(diff, 256)
            for r in range(256):
                if sym[r] < it:
                    qq[r] += 1
            diff -= it
    elif diff < 0:  # :275-315
        diff = -diff
        while diff > 0:
            n = sum(1 for v in qq if v > 1)
            it = min(diff, n)
            for r in range(n - it, n):
                qq[r] -= 1
            diff -= it
    pdf = [0] * 256
    for r in range(256):
        pdf[sym[r]] = qq[r]
    out, cdf = [], 0
    for s in range(256):
        p = pdf[s]
        if p == 0:
            out.append((0, cdf, 0, 0))
            continue
        shift = (p - 1).bit_length()  # 32 - clz(p - 1)
        magic = (((1 << 32) * ((1 << shift) - p)) // p + 1) & 0xFFFFFFFF  # :352-358
        out.append((p, cdf, magic, shift))
        cdf += p
    return out


def encode_block(data, prob_bits, table):
    """ans/GpuANSEncode.cuh:49-211, lane-parallel form (ballot + popc(lanemask_lt))."""
    n = len(data)
    state = [START] * 32
    words = []
    for row in range((n + 31) // 32):
        syms = [data[row * 32 + l] if row * 32 + l < n else None for l in range(32)]
        write = [
            s is not None and state[l] >= (table[s][0] << (31 - prob_bits))
            for l, s in enumerate(syms)
        ]
        base = len(words)
        words.extend([0] * sum(write))
        for l in range(32):
            if write[l]:
                prefix = sum(write[:l])  # popc(vote & lanemask_lt)
                words[base + prefix] = state[l] & 0xFFFF
                state[l] >>= 16
        for l, s in enumerate(syms):
            if s is None:
                continue
            pdf, cdf, magic, shift = table[s]
            t = (state[l] * magic) >> 32
            div = ((t + state[l]) & 0xFFFFFFFF) >> shift
            mod = state[l] - div * pdf
            state[l] = (div << prob_bits) + mod + cdf
    return words, state


def ans_encode(data, prob_bits=10, use_checksum=False):
    """ans/GpuANSEncode.cuh:515-628, 674-849."""
    data = bytes(data)
    size = len(data)
    counts = [0] * 256
    for b in data:
        counts[b] += 1
    table = normalize(counts, size, prob_bits)
    nb = (size + BLOCK - 1) // BLOCK
    states, bw, payload, start = [], [], b"", 0
    for b in range(nb):
        chunk = data[b * BLOCK : (b + 1) * BLOCK]
        words, st = encode_block(chunk, prob_bits, table)
        states.append(st)
        bw.append(((len(chunk) << 16) | len(words), start))
        pad = (-len(words)) % 8
        payload += struct.pack("<%dH" % (len(words) + pad), *(words + [0] * pad))
        start += len(words) + pad
    ck = 0
    if use_checksum:
        for b in data:
            ck ^= b
    hdr = struct.pack(
        "<8I", (0xD00D << 16) | 1, nb, size, start, prob_bits | (int(use_checksum) << 4), ck, 0, 0
    )
    out = hdr + struct.pack("<256H", *[t[0] for t in table])
    for st in states:
        out += struct.pack("<32I", *st)
    for x, y in bw:
        out += struct.pack("<2I", x, y)
    if nb % 2:
        out += struct.pack("<2I", 0, 0)
    return out + payload


def ans_decode(archive, prob_bits=10):
    """ans/GpuANSDecode.cuh:55-217, 299-476 (ballot + popc(lanemask_ge))."""
    a = bytes(archive)
    magic, nb, size, _total, opts, _ck, _, _ = struct.unpack_from("<8I", a, 0)
    assert magic == (0xD00D << 16) | 1 and (opts & 0xF) == prob_bits
    pdf = struct.unpack_from("<256H", a, 32)
    lut, cdf = [0] * (1 << prob_bits), 0
    for s in range(256):
        for j in range(pdf[s]):
            lut[cdf + j] = (j << 20) | (pdf[s] << 8) | s
        cdf += pdf[s]
    off_states = 32 + 512
    off_bw = off_states + 128 * nb
    off_data = off_bw + 8 * ((nb + 1) // 2 * 2)
    out = bytearray(size)
    mask = (1 << prob_bits) - 1
    for b in range(nb):
        state = list(struct.unpack_from("<32I", a, off_states + 128 * b))
        x, start = struct.unpack_from("<2I", a, off_bw + 8 * b)
        n, w = x >> 16, x & 0xFFFF
        words = struct.unpack_from("<%dH" % w, a, off_data + 2 * start)
        pos = w
        for row in range((n + 31) // 32 - 1, -1, -1):
            valid = [row * 32 + l < n for l in range(32)]
            read = [False] * 32
            for l in range(32):
                if not valid[l]:
                    continue
                e = lut[state[l] & mask]
                out[b * BLOCK + row * 32 + l] = e & 0xFF
                state[l] = ((e >> 8) & 0xFFF) * (state[l] >> prob_bits) + (e >> 20)
                read[l] = state[l] < START
            for l in range(32):
                if read[l]:
                    prefix = sum(read[l:])  # popc(vote & lanemask_ge)
                    state[l] = (state[l] << 16) + words[pos - prefix]
            pos -= sum(read)
        assert pos == 0 and all(s == START for s in state)
    return bytes(out)


def float_split_word(ft, w):
    """float/GpuFloatUtils.cuh:111-115, 141-147, 181-185. ft: 1 fp16, 2 bf16, 3 fp32."""
    if ft == 1:
        return w >> 8, w & 0xFF
    if ft == 2:
        v = (w * 65537) & 0xFFFFFFFF
        v = ((v << 1) | (v >> 31)) & 0xFFFFFFFF
        return v >> 24, v & 0xFF
    v = ((w << 1) | (w >> 31)) & 0xFFFFFFFF
    return v >> 24, v & 0xFFFFFF


def float_compress(ft, words, prob_bits=10, use_checksum=False):
    """float/GpuFloatCompress.cuh:280-365, 446-579."""
    words = [int(w) for w in words]
    n = len(words)
    comp, nc = [], []
    for w in words:
        c, r = float_split_word(ft, w)
        comp.append(c)
        nc.append(r)
    if ft == 3:
        n8, n16 = (n + 7) // 8 * 8, (n + 15) // 16 * 16
        plane = struct.pack("<%dH" % n8, *([r & 0xFFFF for r in nc] + [0] * (n8 - n)))
        plane += bytes([r >> 16 for r in nc] + [0] * (n16 - n))
        raw = struct.pack("<%dI" % n, *words)
    else:
        n16 = (n + 15) // 16 * 16
        plane = bytes(nc + [0] * (n16 - n))
        raw = struct.pack("<%dH" % n, *words)
    ck = 0
    if use_checksum:  # only the first n BYTES are covered (GpuFloatCompress.cuh:466-468)
        for b in raw[:n]:
            ck ^= b
    hdr = struct.pack("<4I", (0xF00F << 16) | 1, n, ft | (int(use_checksum) << 4), ck)
    return hdr + plane + ans_encode(bytes(comp), prob_bits, False)
