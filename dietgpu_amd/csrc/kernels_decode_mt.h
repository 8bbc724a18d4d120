// rANS decode of large raw-byte / fp32 elements: FOUR blocks per wavefront, two independent row chains.
//
// Why ("k_ans_decode_mt"): the row of k_ans_decode is one dependent chain -- LUT entry
// (LDS), state update, ballot, word address, word (LDS), renormalise: ~550 cycles per row and wavefront under load
// -- and a SIMD holds at most 8 wavefronts, so on inputs that are not bound by HBM (raw bytes: one rANS symbol per
// input byte) the VALU idles a third of the time waiting for chains.  The number of chains in flight is what LDS
// allows: 2 KiB of word ring per block.  Here
//   * the word ring of a block is 1 KiB (4 x 128-word chunks, refilled every FOUR rows; same protocol and proof as
//     kernels_decode.h with every constant halved), so a wavefront owns four blocks in the LDS two took;
//   * a wavefront decodes two block pairs A and B in lock step: the two chains are independent, the compiler
//     interleaves them, and the renormalisation reads of both are issued back to back under their ballots
//     (one asm statement: the compiler would otherwise sink each read into its own branch region and serialise them);
//   * ring maintenance is scalar: the unread-word counts, the lowest requested chunk and the pending flags of the
//     two halves live in SGPRs, the per-half execution masks come from scalar selects (inverse ballot), so a group
//     costs 2-3 VALU instructions instead of ~24;
//   * a workgroup (8 wavefronts) takes a SLICE of an element -- several 32-block rounds -- builds the LUT once and
//     fetches the next quad's descriptors and states while it decodes the current one.
// Format, validation and status semantics are k_ans_decode's (GpuANSDecode.cuh:55-403); quads that are not four
// full blocks (the tail of an element, malformed blocks) run kernels_decode.h's decodeBlock pair by pair.
#pragma once

#include "kernels_decode.h"

namespace dgpu {

constexpr uint32_t kMtBlocksPerWave = 4;
constexpr uint32_t kMtWaves = 8;
constexpr uint32_t kMtThreads = kMtWaves * 64u;
constexpr uint32_t kMtRoundBlocks = kMtWaves * kMtBlocksPerWave;  // 32 blocks per workgroup round
constexpr uint32_t kMtRingBytes = 1024;                            // per block
constexpr uint32_t kMtChunkWords = 128;
constexpr uint32_t kMtChunkBytes = 256;
constexpr uint32_t kMtGroupRows = 4;

__host__ __device__ constexpr uint32_t decMtLdsBytes(int P, uint32_t ft) {
  (void)ft;
  return kMtWaves * kMtBlocksPerWave * kMtRingBytes + decLutBytes(P, kDecBlocksPerTile);
}

typedef uint32_t u32x2m __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2m LdsU2m;

// Four full blocks: chain c = blocks (2c, 2c+1) of the quad, lower / upper half of the wave.
template <int P, uint32_t FT, bool kCompact>
__device__ __forceinline__ void decodeQuadFull(
    const uint8_t* __restrict__ dataBase,  // uniform: the ANS archive (header first)
    uint8_t* __restrict__ outBase,         // uniform: the element's output
    uint32_t (&state)[2],
    const uint32_t (&numWords)[2],   // per lane: compressed words of my block in chain c
    const uint32_t (&dataOff)[2],    // per lane: byte offset of my block's words from dataBase (16-byte aligned, < 2^31)
    const uint32_t (&outOff)[2],     // per lane: byte offset of my block's first row + hl from outBase
    const uint32_t (&ringBase)[2],   // per lane: LDS address of my block's 1 KiB ring (1 KiB aligned)
    const void* __restrict__ lutRaw,
    uint32_t hl, bool upper) {
  static_assert(FT == 0, "");
  constexpr uint32_t kMask = (1u << P) - 1u;
  auto lutAt = [&](uint32_t x) -> uint2 {
    if (kCompact) {
      const uint32_t e = ((const uint32_t*)lutRaw)[x];
      return make_uint2((e & 0xff000fffu), (e >> 12) & 0xfffu);
    }
    return ((const uint2*)lutRaw)[x];
  };

  // ---- initial fill: every chunk that intersects [numWords - 256, numWords) (at most three)
  uint32_t sLo[2], sHi[2];       // unread words (wave-uniform per half)
  // request threshold of a half: the next lower chunk k is requested as soon as position < (k + 3) * 128, i.e. as
  // soon as the resident words below the position come within two chunks.  It keeps falling once chunk 0 is
  // resident: the (at most two: position < 256, < 128) requests that follow fetch the 256 / 512 bytes BELOW the
  // block's data -- bytes of the same archive (its header and tables alone are >= 688 bytes), never used -- into
  // slots whose chunks have been consumed.  Cheaper than a second scalar compare per half and group.
  int needLo[2], needHi[2];
  int nextOff[2];                // per lane: byte offset within my block's data of the lowest requested chunk, + hl * 8
  uint2 pending[2];
  uint32_t pendSlot[2];
  uint64_t pendMask[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t nw = numWords[c];
    const uint32_t paddedBytes = roundUp(nw, kBlockAlignWords) * 2u;
    const int top = nw ? (int)((nw - 1u) / kMtChunkWords) : -1;
    const int stop = nw > 2u * kMtChunkWords ? (int)((nw - 2u * kMtChunkWords) / kMtChunkWords) : 0;
    uint2 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ch = top - k;
      const uint32_t off = (uint32_t)ch * kMtChunkBytes + hl * 8u;
      v[k] = make_uint2(0, 0);
      if (ch >= stop && off < paddedBytes) v[k] = *(const uint2*)(dataBase + (dataOff[c] + off));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ch = top - k;
      if (ch >= stop) *(LdsU2m*)(uintptr_t)(ringBase[c] | ((((uint32_t)ch & 3u) * kMtChunkBytes) + hl * 8u)) = u32x2m{v[k].x, v[k].y};
    }
    const int low = top < 0 ? 0 : stop;  // lowest resident chunk
    nextOff[c] = low * (int)kMtChunkBytes + (int)(hl * 8u);
    const int need = (low + 2) * (int)kMtChunkWords;
    sLo[c] = __builtin_amdgcn_readlane(nw, 0);
    sHi[c] = __builtin_amdgcn_readlane(nw, 32);
    needLo[c] = __builtin_amdgcn_readlane(need, 0);
    needHi[c] = __builtin_amdgcn_readlane(need, 32);
    pending[c] = make_uint2(0, 0);
    pendSlot[c] = ringBase[c];
    pendMask[c] = 0;
  }
  int upperSel = upper ? 1 : 0;
  asm volatile("" : "+v"(upperSel));  // a VGPR operand of the multiply-add, not a select to be folded into it
  uint32_t rowOff[2] = {outOff[0] + (kRowsPerBlock - kMtGroupRows) * 32u, outOff[1] + (kRowsPerBlock - kMtGroupRows) * 32u};
  (void)outBase;

#pragma unroll 1
  for (int g = (int)(kRowsPerBlock / kMtGroupRows) - 1; g >= 0; --g) {
    // ---- ring maintenance, scalar decisions per half
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (__builtin_amdgcn_inverse_ballot_w64(pendMask[c])) {
        *(LdsU2m*)(uintptr_t)pendSlot[c] = u32x2m{pending[c].x, pending[c].y};
      }
      // all ones when position < threshold, and the threshold moved down by one chunk then: four scalar
      // instructions per half (asm: the compiler turns the comparison into a lane mask and back through a VGPR)
      uint32_t mLo, mHi, tLo, tHi;
      asm("s_sub_i32 %0, %3, %2\n\ts_ashr_i32 %0, %0, 31\n\ts_lshl_b32 %1, %0, 7\n\ts_add_i32 %2, %2, %1"
          : "=&s"(mLo), "=&s"(tLo), "+s"(needLo[c]) : "s"(sLo[c]) : "scc");
      asm("s_sub_i32 %0, %3, %2\n\ts_ashr_i32 %0, %0, 31\n\ts_lshl_b32 %1, %0, 7\n\ts_add_i32 %2, %2, %1"
          : "=&s"(mHi), "=&s"(tHi), "+s"(needHi[c]) : "s"(sHi[c]) : "scc");
      (void)tLo;
      (void)tHi;
      const uint64_t req = (uint64_t)mLo | ((uint64_t)mHi << 32);
      if (__builtin_amdgcn_inverse_ballot_w64(req)) {
        nextOff[c] -= (int)kMtChunkBytes;
        // (a VALID block stops two chunks below its data, see above; the clamp bounds what corrupt states or
        // tables can make the decoder fetch: never more than 512 bytes below a block's data, i.e. inside the archive)
        const int at = nextOff[c] > (int)(hl * 8u) - 2 * (int)kMtChunkBytes ? nextOff[c] : (int)(hl * 8u) - 2 * (int)kMtChunkBytes;
        pending[c] = *(const uint2*)(dataBase + (uint32_t)((int)dataOff[c] + at));  // chunks below the top one are whole
        pendSlot[c] = ((uint32_t)nextOff[c] & (kMtRingBytes - 1u)) | ringBase[c];
      }
      pendMask[c] = req;
    }
    // ---- four rows, both chains in lock step
#pragma unroll
    for (int j = (int)kMtGroupRows - 1; j >= 0; --j) {
      uint2 e[2];
      uint64_t vote[2];
      uint32_t addr[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) e[c] = lutAt(state[c] & kMask);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        state[c] = __umul24(e[c].x, state[c] >> P) + e[c].y;
        vote[c] = __ballot(state[c] < kMinState);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t vLo = (uint32_t)vote[c], vHi = (uint32_t)(vote[c] >> 32);
        const uint32_t sLoOld = sLo[c];
        sLo[c] -= (uint32_t)__popc(vLo);
        sHi[c] -= (uint32_t)__popc(vHi);
        // readers below me in the wave: lower half = my rank, upper half = rank + readers of the lower half
        uint32_t t = __builtin_amdgcn_mbcnt_hi(vHi, __builtin_amdgcn_mbcnt_lo(vLo, 0u));
        t = (uint32_t)(__mul24(upperSel, (int)(sHi[c] - sLoOld)) + (int)t);
        asm volatile("" : "+v"(t));  // keep the scalar position in the add-shift below
        addr[c] = (((t + sLo[c]) << 1) & (kMtRingBytes - 1u)) | ringBase[c];
      }
      {
        // renormalisation of both chains: the two word reads go out back to back under their ballots
        uint32_t wA, wB;
        asm volatile(
            "s_mov_b64 exec, %[va]\n\t"
            "ds_read_u16 %[wa], %[aa]\n\t"
            "s_mov_b64 exec, %[vb]\n\t"
            "ds_read_u16 %[wb], %[ab]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_lshl_or_b32 %[sb], %[sb], 16, %[wb]\n\t"
            "s_mov_b64 exec, %[va]\n\t"
            "v_lshl_or_b32 %[sa], %[sa], 16, %[wa]\n\t"
            "s_mov_b64 exec, -1"
            : [sa] "+v"(state[0]), [sb] "+v"(state[1]), [wa] "=&v"(wA), [wb] "=&v"(wB)
            : [aa] "v"(addr[0]), [ab] "v"(addr[1]), [va] "s"(vote[0]), [vb] "s"(vote[1])
            : "memory");
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) (outBase + rowOff[c])[j * 32] = (uint8_t)(e[c].x >> 24);
    }
    rowOff[0] -= kMtGroupRows * 32u;
    rowOff[1] -= kMtGroupRows * 32u;
  }
}

// grid = (slices, B), 512 threads.  sliceBlocks: blocks of an element one workgroup decodes (multiple of 32).
template <int P, uint32_t FT>
__global__ __launch_bounds__(kMtThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ans_decode_mt(DecodeArgs a, uint32_t sliceBlocks) {
  static_assert(FT == 0, "raw bytes only (the per-row sinks of the float types need their non-compressed bytes)");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kRingArea = kMtWaves * kMtBlocksPerWave * kMtRingBytes;  // 32 KiB at LDS offset 0
  constexpr bool kCompact = decCompactLut(P, kDecBlocksPerTile);
  uint2* sLut = (uint2*)(smem + kRingArea);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t b = blockIdx.y;
  const uint32_t slice = blockIdx.x;

  const uint8_t* archive = a.in.ptr(b);
  const uint64_t inBytes = a.inBytes ? (uint64_t)a.inBytes[b] : ~0ull;
  if (inBytes < sizeof(AnsHeader)) {  // uniform: not even a header
    if (slice == 0 && tid == 0) {
      if (a.outSuccess) a.outSuccess[b] = 0;
      if (a.outSize) a.outSize[b] = 0;
    }
    return;
  }
  const uint8_t* ans = archive;
  const AnsHeader header = *(const AnsHeader*)ans;
  const uint32_t nb = header.numBlocks;
  const uint32_t total = header.totalUncompressedWords;
  const uint32_t totalWords = header.totalCompressedWords;

  // the checks of k_ans_decode (kernels_decode.h), in the same order
  bool success = a.out.size(b) >= total;
  success = success && header.magicAndVersion == ((kAnsMagic << 16) | kAnsVersion) && (header.options & 0xfu) == (uint32_t)P;
  success = success && nb == divUp(total, kBlockSize);
  success = success && (uint64_t)ansOverhead(nb) + 2ull * totalWords <= inBytes;
  if (!success) {  // uniform
    if (slice == 0 && tid == 0) {
      if (a.outSuccess) a.outSuccess[b] = 0;
      if (a.outSize) a.outSize[b] = total;
    }
    return;
  }
  const uint32_t sliceFirst = slice * sliceBlocks;
  if (sliceFirst >= nb && slice != 0) return;
  const uint32_t sliceEnd = sliceFirst + sliceBlocks < nb ? sliceFirst + sliceBlocks : nb;

  const uint2* blockWords = (const uint2*)(ans + ansBlockWordsOffset(nb));
  auto blockOk = [&](uint32_t i, uint2 bw) -> bool {
    const uint32_t want = (i + 1u < nb) ? kBlockSize : total - i * kBlockSize;
    const uint32_t words = bw.x & 0xffffu;
    return (bw.x >> 16) == want && (bw.y & (kBlockAlignWords - 1u)) == 0u &&
        (uint64_t)bw.y + roundUp(words, kBlockAlignWords) <= (uint64_t)totalWords;
  };
  // slice 0 vouches for the whole element (it alone writes outSuccess)
  bool allBlocksOk = true;
  if (slice == 0) {
    for (uint32_t i = tid; i < nb; i += kMtThreads) allBlocksOk = allBlocksOk && blockOk(i, blockWords[i]);
  }

  // ---- decode LUT (as k_ans_decode: wave 0 scans the pdfs, all threads fill the slots by binary search)
  uint32_t* sCdf = (uint32_t*)smem;
  uint32_t* sPdf = sCdf + kNumSymbols;
  uint32_t* sPdfSum = sPdf + kNumSymbols;
  uint32_t* sWaveBad = sPdfSum + 1;
  {
    const bool waveBad = __ballot(!allBlocksOk) != 0ull;
    if (lane == 0u) sWaveBad[wave] = waveBad ? 1u : 0u;
  }
  if (wave == 0) {
    const uint2 raw = ((const uint2*)(ans + sizeof(AnsHeader)))[lane];
    const uint32_t p0 = raw.x & 0xffffu, p1 = raw.x >> 16, p2 = raw.y & 0xffffu, p3 = raw.y >> 16;
    const uint32_t mine = p0 + p1 + p2 + p3;
    const uint32_t incl = waveInclusiveScan(mine, lane);
    const uint32_t base = incl - mine;
    ((uint4*)sCdf)[lane] = make_uint4(base, base + p0, base + p0 + p1, base + p0 + p1 + p2);
    ((uint4*)sPdf)[lane] = make_uint4(p0, p1, p2, p3);
    if (lane == 63u) *sPdfSum = incl;
  }
  __syncthreads();
  uint32_t anyBad = 0;
#pragma unroll
  for (uint32_t w = 0; w < kMtWaves; ++w) anyBad |= sWaveBad[w];
  const bool pdfOk = nb == 0u || *sPdfSum == (1u << P);
  if (slice == 0 && tid == 0) {
    if (a.outSuccess) a.outSuccess[b] = (anyBad == 0u && pdfOk) ? 1 : 0;
    if (a.outSize) a.outSize[b] = total;
  }
  if (!pdfOk || sliceFirst >= nb) return;  // uniform
  for (uint32_t x = tid; x < (1u << P); x += kMtThreads) {
    uint32_t lo = 0, hi = kNumSymbols;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const uint32_t mid = (lo + hi) >> 1;
      const bool le = sCdf[mid] <= x;
      lo = le ? mid : lo;
      hi = le ? hi : mid;
    }
    if (kCompact) ((uint32_t*)sLut)[x] = (sPdf[lo] & 0xfffu) | (((x - sCdf[lo]) & 0xfffu) << 12) | (lo << 24);
    else sLut[x] = make_uint2((sPdf[lo] & 0xfffu) | (lo << 24), (x - sCdf[lo]) & 0xfffu);
  }

  uint8_t* outBase = a.out.ptr(b);
  const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t waveRegion = ldsBase + wave * (kMtBlocksPerWave * kMtRingBytes);

  // descriptor + state of block q + 2 c + (upper ? 1 : 0), c = 0, 1
  struct Desc {
    uint2 bw[2];
    uint32_t state[2];
  };
  auto loadDesc = [&](uint32_t q) -> Desc {
    Desc d;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
      d.bw[c] = make_uint2(0, 0);
      d.state[c] = 0;
      if (blk < sliceEnd) {
        d.bw[c] = blockWords[blk];
        d.state[c] = ((const uint32_t*)(ans + ansStatesOffset()))[blk * 32u + hl];
      }
    }
    return d;
  };

  uint32_t q = sliceFirst + wave * kMtBlocksPerWave;
  Desc cur = loadDesc(q);
  __syncthreads();  // LUT visible to every wave, scratch free (from here on every wave's ring region is private)

  for (; q < sliceEnd; q += kMtRoundBlocks) {
    const Desc next = loadDesc(q + kMtRoundBlocks);  // blocks beyond the slice: zeros
    uint32_t n[2], numWords[2], dataOff[2], state[2];
    bool have[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
      // (malformed block: neither read nor written; byte offsets of block data fit 32 bits: compressed elements are
      // below 2 GiB, dgpu_ans_max_compressed_size)
      have[c] = blk < sliceEnd && blockOk(blk, cur.bw[c]) && cur.bw[c].y < 0x3fffffffu;
      n[c] = have[c] ? cur.bw[c].x >> 16 : 0u;
      numWords[c] = have[c] ? cur.bw[c].x & 0xffffu : 0u;
      dataOff[c] = ansOverhead(nb) + (have[c] ? 2u * cur.bw[c].y : 0u);
      state[c] = cur.state[c];
    }
    const bool quadFull = __ballot(n[0] != kBlockSize || n[1] != kBlockSize) == 0ull;  // uniform
    if (quadFull) {
      uint32_t ringBase[2], outOff[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
        outOff[c] = blk * kBlockSize + hl;
        ringBase[c] = waveRegion + (2u * (uint32_t)c + (upper ? 1u : 0u)) * kMtRingBytes;
      }
      decodeQuadFull<P, FT, kCompact>(ans, outBase, state, numWords, dataOff, outOff, ringBase, sLut, hl, upper);
    } else {
      // the tail of the element (or a malformed block): pair by pair through the general path, whose two 2 KiB
      // rings are this wave's 4 KiB
#pragma unroll 1
      for (int c = 0; c < 2; ++c) {
        const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
        const uint32_t nFirst = __shfl(n[c], 0, 64);
        const uint32_t nSecond = __shfl(n[c], 32, 64);
        const uint32_t maxN = nFirst > nSecond ? nFirst : nSecond;
        if (maxN == 0u) continue;  // uniform
        RowSink<FT> sink;
        sink.init(outBase, archive, 0u, (size_t)(have[c] ? blk : (blk & ~1u)) * kBlockSize, hl);
        decodeBlock<P, FT, false, false, false, kCompact>(
            0u, state[c], n[c], divUp(divUp(maxN, 32u), kGroupRows), ans + dataOff[c], numWords[c], smem,
            waveRegion + (upper ? kRingBytes : 0u), sLut, sink, hl, upper);
      }
    }
    cur = next;
  }
}

}  // namespace dgpu
