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

// Geometry of a variant: kChains row chains per wavefront (each = two blocks, lower / upper half), kRing bytes of
// word ring per block (4 chunks; a group = kRing / 256 rows consumes at most one chunk), kWaves wavefronts per
// workgroup.  Shipped: <2, 1024, 8>.
template <int kChains, uint32_t kRing, uint32_t kWaves>
struct MtGeom {
  static constexpr uint32_t kBlocksPerWave = 2u * (uint32_t)kChains;
  static constexpr uint32_t kThreads = kWaves * 64u;
  static constexpr uint32_t kRoundBlocks = kWaves * kBlocksPerWave;  // blocks per workgroup round
  static constexpr uint32_t kChunkBytes = kRing / 4u;
  static constexpr uint32_t kChunkWords = kChunkBytes / 2u;
  static constexpr uint32_t kGroupRows = kChunkWords / 32u;          // 4 (1 KiB ring) or 8 (2 KiB)
  static constexpr uint32_t kLaneBytes = kChunkBytes / 32u;          // 8 or 16 bytes per lane and chunk
  static constexpr uint32_t kWaveRegion = kBlocksPerWave * kRing;    // >= 4 KiB: the general path's two 2 KiB rings
  static_assert(kWaveRegion >= 2u * kRingBytes, "");
  static_assert(kRing == 1024u || kRing == 2048u, "");
  __host__ __device__ static constexpr uint32_t ldsBytes(int P) { return kWaves * kWaveRegion + decLutBytes(P, kDecBlocksPerTile); }
};

typedef uint32_t u32x2m __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2m LdsU2m;

// One chunk slice of a lane: 8 bytes (1 KiB ring) or 16 bytes (2 KiB ring)
template <uint32_t kBytes>
struct LaneChunk;
template <>
struct LaneChunk<8> {
  uint2 v;
  __device__ __forceinline__ void zero() { v = make_uint2(0, 0); }
  __device__ __forceinline__ void load(const uint8_t* p) { v = *(const uint2*)p; }
  __device__ __forceinline__ void toLds(uint32_t addr) const { *(LdsU2m*)(uintptr_t)addr = u32x2m{v.x, v.y}; }
};
template <>
struct LaneChunk<16> {
  uint4 v;
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void load(const uint8_t* p) { v = *(const uint4*)p; }
  __device__ __forceinline__ void toLds(uint32_t addr) const { *(LdsU4*)(uintptr_t)addr = u32x4{v.x, v.y, v.z, v.w}; }
};

// DGPU_DEC_MT_UNCOND: every lane reads a ring word (the address is always inside its ring) and keeps it by a
// v_cndmask, instead of the readers reading under their ballot as execution mask: one more VALU, two fewer scalar
// instructions per row and chain.
#ifndef DGPU_DEC_MT_UNCOND
#define DGPU_DEC_MT_UNCOND 0
#endif

// kChains x two full blocks: chain c = blocks (2c, 2c+1) of the wave's group, lower / upper half of the wave.
template <int P, uint32_t FT, bool kCompact, typename G, int kChains>
__device__ __forceinline__ void decodeFullChains(
    const uint8_t* __restrict__ dataBase,  // uniform: the ANS archive (header first)
    uint8_t* __restrict__ outBase,         // uniform: the element's output
    uint32_t (&state)[kChains],
    const uint32_t (&numWords)[kChains],   // per lane: compressed words of my block in chain c
    const uint32_t (&dataOff)[kChains],    // per lane: byte offset of my block's words from dataBase (16-byte aligned, < 2^31)
    const uint32_t (&outOff)[kChains],     // per lane: byte offset of my block's first row + hl from outBase
    const uint32_t (&ringBase)[kChains],   // per lane: LDS address of my block's ring (aligned to its size)
    const void* __restrict__ lutRaw,
    uint32_t hl, bool upper) {
  static_assert(FT == 0, "");
  constexpr uint32_t kMask = (1u << P) - 1u;
  constexpr uint32_t kRing = G::kChunkBytes * 4u;
  constexpr uint32_t kCW = G::kChunkWords, kCB = G::kChunkBytes, kLB = G::kLaneBytes;
  constexpr int kLog2CW = kCW == 128u ? 7 : 8;
  auto lutAt = [&](uint32_t x) -> uint2 {
    if (kCompact) {
      const uint32_t e = ((const uint32_t*)lutRaw)[x];
      return make_uint2((e & 0xff000fffu), (e >> 12) & 0xfffu);
    }
    return ((const uint2*)lutRaw)[x];
  };

  // ---- initial fill: every chunk that intersects [numWords - 2 chunks, numWords) (at most three)
  uint32_t sLo[kChains], sHi[kChains];  // unread words (wave-uniform per half)
  // request threshold of a half: the next lower chunk k is requested as soon as position < (k + 3) chunks, i.e. as
  // soon as the resident words below the position come within two chunks.  It keeps falling once chunk 0 is
  // resident: the (at most two: position < 2 chunks, < 1 chunk) requests that follow fetch the bytes BELOW the
  // block's data -- bytes of the same archive (its header and tables alone are >= 688 bytes; the address is
  // clamped to two chunks below the block for corrupt inputs), never used -- into slots whose chunks have been
  // consumed.  Cheaper than a second scalar compare per half and group.
  int needLo[kChains], needHi[kChains];
  int nextOff[kChains];  // per lane: byte offset within my block's data of the lowest requested chunk, + hl * kLB
  LaneChunk<kLB> pending[kChains];
  uint32_t pendSlot[kChains];
  uint64_t pendMask[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) {
    const uint32_t nw = numWords[c];
    const uint32_t paddedBytes = roundUp(nw, kBlockAlignWords) * 2u;
    const int top = nw ? (int)((nw - 1u) / kCW) : -1;
    const int stop = nw > 2u * kCW ? (int)((nw - 2u * kCW) / kCW) : 0;
    LaneChunk<kLB> v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ch = top - k;
      const uint32_t off = (uint32_t)ch * kCB + hl * kLB;
      v[k].zero();
      if (ch >= stop && off < paddedBytes) v[k].load(dataBase + (dataOff[c] + off));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ch = top - k;
      if (ch >= stop) v[k].toLds(ringBase[c] | ((((uint32_t)ch & 3u) * kCB) + hl * kLB));
    }
    const int low = top < 0 ? 0 : stop;  // lowest resident chunk
    nextOff[c] = low * (int)kCB + (int)(hl * kLB);
    const int need = (low + 2) * (int)kCW;
    sLo[c] = __builtin_amdgcn_readlane(nw, 0);
    sHi[c] = __builtin_amdgcn_readlane(nw, 32);
    needLo[c] = __builtin_amdgcn_readlane(need, 0);
    needHi[c] = __builtin_amdgcn_readlane(need, 32);
    pending[c].zero();
    pendSlot[c] = ringBase[c];
    pendMask[c] = 0;
  }
  int upperSel = upper ? 1 : 0;
  asm volatile("" : "+v"(upperSel));  // a VGPR operand of the multiply-add, not a select to be folded into it
  uint32_t rowOff[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) rowOff[c] = outOff[c] + (kRowsPerBlock - G::kGroupRows) * 32u;

#pragma unroll 1
  for (int g = (int)(kRowsPerBlock / G::kGroupRows) - 1; g >= 0; --g) {
    // ---- ring maintenance, scalar decisions per half
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      if (__builtin_amdgcn_inverse_ballot_w64(pendMask[c])) pending[c].toLds(pendSlot[c]);
      // all ones when position < threshold, and the threshold moved down by one chunk then: four scalar
      // instructions per half (asm: the compiler turns the comparison into a lane mask and back through a VGPR)
      uint32_t mLo, mHi, tLo, tHi;
      asm("s_sub_i32 %0, %3, %2\n\ts_ashr_i32 %0, %0, 31\n\ts_lshl_b32 %1, %0, %4\n\ts_add_i32 %2, %2, %1"
          : "=&s"(mLo), "=&s"(tLo), "+s"(needLo[c]) : "s"(sLo[c]), "n"(kLog2CW) : "scc");
      asm("s_sub_i32 %0, %3, %2\n\ts_ashr_i32 %0, %0, 31\n\ts_lshl_b32 %1, %0, %4\n\ts_add_i32 %2, %2, %1"
          : "=&s"(mHi), "=&s"(tHi), "+s"(needHi[c]) : "s"(sHi[c]), "n"(kLog2CW) : "scc");
      (void)tLo;
      (void)tHi;
      const uint64_t req = (uint64_t)mLo | ((uint64_t)mHi << 32);
      if (__builtin_amdgcn_inverse_ballot_w64(req)) {
        nextOff[c] -= (int)kCB;
        const int floorOff = (int)(hl * kLB) - 2 * (int)kCB;
        const int at = nextOff[c] > floorOff ? nextOff[c] : floorOff;
        pending[c].load(dataBase + (uint32_t)((int)dataOff[c] + at));  // chunks below the top one are whole
        pendSlot[c] = ((uint32_t)nextOff[c] & (kRing - 1u)) | ringBase[c];
      }
      pendMask[c] = req;
    }
    // ---- the rows of the group, all chains in lock step
#pragma unroll
    for (int j = (int)G::kGroupRows - 1; j >= 0; --j) {
      uint2 e[kChains];
      uint64_t vote[kChains];
      uint32_t addr[kChains];
#pragma unroll
      for (int c = 0; c < kChains; ++c) e[c] = lutAt(state[c] & kMask);
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        state[c] = __umul24(e[c].x, state[c] >> P) + e[c].y;
        vote[c] = __ballot(state[c] < kMinState);
      }
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        const uint32_t vLo = (uint32_t)vote[c], vHi = (uint32_t)(vote[c] >> 32);
        const uint32_t sLoOld = sLo[c];
        sLo[c] -= (uint32_t)__popc(vLo);
        sHi[c] -= (uint32_t)__popc(vHi);
        // readers below me in the wave: lower half = my rank, upper half = rank + readers of the lower half
        uint32_t t = __builtin_amdgcn_mbcnt_hi(vHi, __builtin_amdgcn_mbcnt_lo(vLo, 0u));
        t = (uint32_t)(__mul24(upperSel, (int)(sHi[c] - sLoOld)) + (int)t);
        asm volatile("" : "+v"(t));  // keep the scalar position in the add-shift below
        addr[c] = (((t + sLo[c]) << 1) & (kRing - 1u)) | ringBase[c];
      }
      if (DGPU_DEC_MT_UNCOND) {
        uint32_t w[kChains];
#pragma unroll
        for (int c = 0; c < kChains; ++c) w[c] = *(const volatile LdsU16*)(uintptr_t)addr[c];  // (volatile: not sunk into a branch)
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
          const uint32_t renorm = (state[c] << kEncodedBits) | w[c];
          state[c] = (state[c] < kMinState) ? renorm : state[c];
        }
      } else if constexpr (kChains == 2) {
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
      } else {
        uint32_t wA;
        asm volatile(
            "s_mov_b64 exec, %[va]\n\t"
            "ds_read_u16 %[wa], %[aa]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_lshl_or_b32 %[sa], %[sa], 16, %[wa]\n\t"
            "s_mov_b64 exec, -1"
            : [sa] "+v"(state[0]), [wa] "=&v"(wA)
            : [aa] "v"(addr[0]), [va] "s"(vote[0])
            : "memory");
      }
#pragma unroll
      for (int c = 0; c < kChains; ++c) (outBase + rowOff[c])[j * 32] = (uint8_t)(e[c].x >> 24);
    }
#pragma unroll
    for (int c = 0; c < kChains; ++c) rowOff[c] -= G::kGroupRows * 32u;
  }
}

// grid = (slices, B), kWaves * 64 threads.  sliceBlocks: blocks of an element one workgroup decodes (a multiple of
// the workgroup round).
template <int P, uint32_t FT, int kChains, uint32_t kRing, uint32_t kWaves>
__global__ __launch_bounds__(kWaves * 64u) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ans_decode_mt(DecodeArgs a, uint32_t sliceBlocks) {
  static_assert(FT == 0, "raw bytes only (the per-row sinks of the float types need their non-compressed bytes)");
  typedef MtGeom<kChains, kRing, kWaves> G;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kRingArea = kWaves * G::kWaveRegion;  // at LDS offset 0
  constexpr bool kCompact = decCompactLut(P, kDecBlocksPerTile);
  constexpr uint32_t kThreads = G::kThreads;
  uint2* sLut = (uint2*)(smem + kRingArea);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t b = blockIdx.y;
  const uint32_t slice = blockIdx.x;

  const uint8_t* archive = a.in.ptr(b);
  const uint64_t inBytes = decodeInBytes(a, b);
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
    for (uint32_t i = tid; i < nb; i += kThreads) allBlocksOk = allBlocksOk && blockOk(i, blockWords[i]);
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
  for (uint32_t w = 0; w < kWaves; ++w) anyBad |= sWaveBad[w];
  const bool pdfOk = nb == 0u || *sPdfSum == (1u << P);
  if (slice == 0 && tid == 0) {
    if (a.outSuccess) a.outSuccess[b] = (anyBad == 0u && pdfOk) ? 1 : 0;
    if (a.outSize) a.outSize[b] = total;
  }
  if (!pdfOk || sliceFirst >= nb) return;  // uniform
  for (uint32_t x = tid; x < (1u << P); x += kThreads) {
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
  const uint32_t waveRegion = ldsBase + wave * G::kWaveRegion;

  // descriptor + state of block q + 2 c + (upper ? 1 : 0), c = 0 .. kChains - 1
  struct Desc {
    uint2 bw[kChains];
    uint32_t state[kChains];
  };
  auto loadDesc = [&](uint32_t q) -> Desc {
    Desc d;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
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

  uint32_t q = sliceFirst + wave * G::kBlocksPerWave;
  Desc cur = loadDesc(q);
  __syncthreads();  // LUT visible to every wave, scratch free (from here on every wave's ring region is private)

  for (; q < sliceEnd; q += G::kRoundBlocks) {
    const Desc next = loadDesc(q + G::kRoundBlocks);  // blocks beyond the slice: zeros
    uint32_t n[kChains], numWords[kChains], dataOff[kChains], state[kChains];
    bool have[kChains];
    bool notFull = false;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
      // (malformed block: neither read nor written; byte offsets of block data fit 32 bits: compressed elements are
      // below 2 GiB, dgpu_ans_max_compressed_size)
      have[c] = blk < sliceEnd && blockOk(blk, cur.bw[c]) && cur.bw[c].y < 0x3fffffffu;
      n[c] = have[c] ? cur.bw[c].x >> 16 : 0u;
      numWords[c] = have[c] ? cur.bw[c].x & 0xffffu : 0u;
      dataOff[c] = ansOverhead(nb) + (have[c] ? 2u * cur.bw[c].y : 0u);
      state[c] = cur.state[c];
      notFull = notFull || n[c] != kBlockSize;
    }
    const bool allFull = __ballot(notFull) == 0ull;  // uniform
    if (allFull) {
      uint32_t ringBase[kChains], outOff[kChains];
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        const uint32_t blk = q + 2u * (uint32_t)c + (upper ? 1u : 0u);
        outOff[c] = blk * kBlockSize + hl;
        ringBase[c] = waveRegion + (2u * (uint32_t)c + (upper ? 1u : 0u)) * kRing;
      }
      decodeFullChains<P, FT, kCompact, G, kChains>(ans, outBase, state, numWords, dataOff, outOff, ringBase, sLut, hl, upper);
    } else {
      // the tail of the element (or a malformed block): pair by pair through the general path, whose two 2 KiB
      // rings are the first 4 KiB of this wave's region
#pragma unroll 1
      for (int c = 0; c < kChains; ++c) {
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
