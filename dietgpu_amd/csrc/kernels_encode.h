// rANS encode for gfx950, single pass.
//
// What it computes is the reference's ansEncodeBatchFull/Partial +
// batchExclusivePrefixSum + ansEncodeCoalesceBatch
// (dietgpu/ans/GpuANSEncode.cuh:49-211, 301-672); how it computes it is
// different:
//
//   * The wire format interleaves 32 rANS states per 4 KiB block.  A wave64
//     therefore encodes TWO blocks at once: lanes 0-31 own block 2w, lanes
//     32-63 own block 2w+1.  One 64-bit ballot per row serves both halves;
//     each half takes its own 32-bit slice for the prefix popcount.
//   * Each half-wave emits its u16 words into an LDS stage (no scattered 2-byte
//     global stores, no 5248-byte-per-block scratch in HBM).
//   * A workgroup (4 waves = 8 blocks = one "tile") publishes the padded word
//     count of its tile and obtains its archive offset with a decoupled
//     look-back over the preceding tiles of the same batch element, then
//     copies the stage to its final place with 16-byte stores.  The input is
//     read once, the archive written once; there is no coalesce pass.
//   * Tiles are handed out by an atomic ticket in (element, tile) order, so a
//     tile only ever waits on tiles that have already started: the look-back
//     cannot deadlock whatever order the hardware dispatches workgroups in.
//   * Hand-off words are single 8-byte {status, value} granules written and
//     polled with relaxed agent-scope atomics (write-through sc1 stores /
//     L1-bypassing loads), the placement-independent form for gfx950's
//     non-coherent per-XCD L2s.
#pragma once

#include "format.h"
#include "kernels_stats.h"

namespace dgpu {

// Upper bound of u16 words one block can emit: per lane, 128 symbols of at
// most P bits each plus the 16-bit start/end slack and the sub-bit rounding
// slop of the state update => 8 * P + 1 words per lane.
#ifdef DGPU_EXPERIMENT_STAGE_WORDS
__host__ __device__ constexpr uint32_t encStageWords(int) { return DGPU_EXPERIMENT_STAGE_WORDS; }
#else
__host__ __device__ constexpr uint32_t encStageWords(int P) { return 32u * (8u * (uint32_t)P + 1u); }
#endif
__host__ __device__ constexpr uint32_t encLdsBytes(int P) {
  return 4096u                                   // packed symbol table
      + kBlocksPerTile * encStageWords(P) * 2u   // bitstream stage per half-wave
      + kBlocksPerTile * 512u                    // input ring, 16 rows per half-wave
      + 128u;                                    // tile bookkeeping
}

constexpr uint64_t kDescAggregate = 1ull << 62;
constexpr uint64_t kDescInclusive = 2ull << 62;
constexpr uint64_t kDescValueMask = (1ull << 62) - 1;

struct EncodeArgs {
  BatchView in;              // bytes to encode (raw input or the float comp plane)
  BatchView out;             // archive base pointers
  BatchView sizes;           // size(b) = number of symbols of element b
  uint32_t floatType;        // != 0: ANS archive embedded in a float archive
  const uint4* encTable;     // [B][256] from k_normalize
  uint32_t maxTiles;         // tiles per element the ticket space is laid out for
  uint64_t* tileDesc;        // [B][maxTiles], zeroed before launch
  uint32_t* ticket;          // zeroed before launch
  uint32_t* outSize;         // [B] nullable
};

struct TileShared {
  uint32_t ticket;
  uint32_t tileBase;          // exclusive prefix (u16 words) of this tile in the element
  uint32_t words[kBlocksPerTile];
  uint32_t localOff[kBlocksPerTile];
};

template <int P, bool kFull>
__device__ __forceinline__ uint32_t encodeRows(
    const uint8_t* __restrict__ inBlock,  // this half's block (global)
    uint32_t n,                           // symbols in this half's block (0 = idle half)
    uint32_t maxRows,                     // wave-uniform row count
    const uint4* __restrict__ table,      // LDS
    uint16_t* __restrict__ stage,         // LDS, this half's stage
    uint8_t* __restrict__ ring,           // LDS, this half's 512-byte input ring
    uint32_t hl,
    bool upper,
    uint32_t& stateOut) {
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  uint32_t state = kStartState;
  uint32_t outOff = 0;

  auto step = [&](const uint4 e, bool valid) {
    const bool write = valid && (state >= e.x);
    const uint64_t vote = __ballot(write);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    if (write) {
      stage[outOff + __popc(vh & laneMaskLt)] = (uint16_t)(state & 0xffffu);
      state >>= kEncodedBits;
    }
    // state = ((state / pdf) << P) + state % pdf + cdf
    //       = state + cdf + (state / pdf) * (2^P - pdf)
    const uint32_t t = __umulhi(state, e.y);
    const uint32_t div = (t + state) >> (e.w >> 24);
    const uint32_t next = __umul24(div, e.w) + state + e.z;
    state = valid ? next : state;
    outOff += __popc(vh);
  };

  if (kFull) {
    // 8 chunks of 16 rows; chunk c+1 is in flight in registers while chunk c is
    // consumed from the LDS ring (same wave writes and reads it: LDS ops of one
    // wave execute in order, no barrier needed).  Neither the symbol fetch nor
    // the table lookup depends on the rANS state, so both are issued ahead of
    // the dependent chain: all 16 symbols at the chunk start, table entries
    // kAhead rows ahead.
    constexpr int kAhead = 4;
    const uint4* src = (const uint4*)inBlock + hl;
    uint4 cur = src[0];
#pragma unroll 1
    for (uint32_t c = 0; c < kRowsPerBlock / 16; ++c) {
      *(uint4*)(ring + hl * 16u) = cur;
      if (c + 1 < kRowsPerBlock / 16) cur = src[(c + 1) * 32u];
      uint32_t sym[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sym[r] = ring[r * 32 + hl];
      uint4 e[kAhead];
#pragma unroll
      for (int r = 0; r < kAhead; ++r) e[r] = table[sym[r]];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint4 cur_e = e[r % kAhead];
        if (r + kAhead < 16) e[r % kAhead] = table[sym[r + kAhead]];
        step(cur_e, true);
      }
    }
  } else {
#pragma unroll 1
    for (uint32_t row = 0; row < maxRows; ++row) {
      const uint32_t i = row * 32u + hl;
      const bool valid = i < n;
      const uint32_t sym = valid ? inBlock[i] : 0u;
      step(table[sym], valid);
    }
  }
  stateOut = state;
  return outOff;
}

template <int P>
__global__ __launch_bounds__(256) void k_ans_encode(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // bookkeeping sits BELOW the stages so that a stage overrun (only possible
  // with a caller-supplied histogram that does not match the data) can never
  // reach it
  uint4* sTable = (uint4*)smem;
  TileShared* sh = (TileShared*)(smem + 4096);
  uint16_t* sStage = (uint16_t*)(smem + 4096 + 128);
  uint8_t* sRing = smem + 4096 + 128 + kBlocksPerTile * encStageWords(P) * 2u;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);

  if (tid == 0) {
    sh->ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const uint32_t ticket = sh->ticket;
  const uint32_t b = ticket / a.maxTiles;
  const uint32_t tile = ticket - b * a.maxTiles;

  const uint32_t size = a.sizes.size(b);
  const uint32_t nb = divUp(size, kBlockSize);
  const uint32_t numTiles = divUp(nb, kBlocksPerTile);
  if (tile >= numTiles) return;  // uniform for the workgroup

  sTable[tid] = a.encTable[b * kNumSymbols + tid];
  __syncthreads();

  const uint8_t* in = a.in.ptr(b);
  uint8_t* ans = a.out.ptr(b) + ansOffsetInArchive(a.floatType, size);

  const uint32_t block = tile * kBlocksPerTile + hw;
  const bool haveBlock = block < nb;
  uint32_t n = 0;
  if (haveBlock) {
    const uint32_t begin = block * kBlockSize;
    n = size - begin < kBlockSize ? size - begin : kBlockSize;
  }
  // wave-uniform: are both halves full blocks?
  const uint32_t firstBlockOfWave = tile * kBlocksPerTile + wave * 2u;
  const bool waveFull = (uint64_t)(firstBlockOfWave + 2u) * kBlockSize <= (uint64_t)size &&
      (((uintptr_t)in & 15u) == 0);

  uint16_t* stage = sStage + hw * encStageWords(P);
  uint32_t state;
  uint32_t words;
  const uint8_t* inBlock = in + (size_t)block * kBlockSize;
  if (waveFull) {
    words = encodeRows<P, true>(inBlock, n, kRowsPerBlock, sTable, stage, sRing + hw * 512u, hl, upper, state);
  } else {
    // rows needed by the larger of the two halves (uniform)
    uint32_t nA = 0;
    if (firstBlockOfWave < nb) {
      uint32_t beginA = firstBlockOfWave * kBlockSize;
      nA = size - beginA < kBlockSize ? size - beginA : kBlockSize;  // first half is never smaller than the second
    }
    words = encodeRows<P, false>(inBlock, n, divUp(nA, 32u), sTable, stage, nullptr, hl, upper, state);
  }

  if (haveBlock) {
    // final lane states, 128 contiguous bytes per block (GpuANSEncode.cuh:207, :584-590)
    ((uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl] = state;
    // zero the pad up to the 16-byte boundary
    const uint32_t padded = roundUp(words, kBlockAlignWords);
    if (words + hl < padded) stage[words + hl] = 0;
  }
  words = words < encStageWords(P) ? words : encStageWords(P);
  if (hl == 0) sh->words[hw] = haveBlock ? words : 0u;
  __syncthreads();

  if (wave == 0) {
    // local exclusive scan of the padded sizes of the tile's 8 blocks
    uint32_t myPadded = (lane < kBlocksPerTile) ? roundUp(sh->words[lane], kBlockAlignWords) : 0u;
    uint32_t incl = waveInclusiveScan(myPadded, lane);
    const uint32_t aggregate = __shfl(incl, kBlocksPerTile - 1, 64);
    if (lane < kBlocksPerTile) sh->localOff[lane] = incl - myPadded;

    uint64_t* desc = a.tileDesc + (size_t)b * a.maxTiles;
    if (lane == 0) {
      __hip_atomic_store(&desc[tile], kDescAggregate | (uint64_t)aggregate, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }

    // decoupled look-back, 64 predecessors per step
    uint32_t exclusive = 0;
    int base = (int)tile - 1;
    while (base >= 0) {
      const int idx = base - (int)lane;
      uint64_t d = kDescInclusive;  // virtual tile -1: inclusive prefix 0
      if (idx >= 0) {
        do {
          d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((d >> 62) == 0) __builtin_amdgcn_s_sleep(1);
        } while ((d >> 62) == 0);
      }
      const uint64_t inclMask = __ballot((d >> 62) == 2);
      const int firstIncl = inclMask ? (__ffsll((unsigned long long)inclMask) - 1) : 64;
      const uint32_t v = ((int)lane <= firstIncl) ? (uint32_t)(d & kDescValueMask) : 0u;
      exclusive += waveReduceSum(v);
      if (firstIncl < 64) break;
      base -= 64;
    }

    const uint32_t inclusive = exclusive + aggregate;
    if (lane == 0) {
      __hip_atomic_store(&desc[tile], kDescInclusive | (uint64_t)inclusive, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      sh->tileBase = exclusive;
      if (tile == numTiles - 1) {
        // complete the header (GpuANSEncode.cuh:533-566)
        ((AnsHeader*)ans)->totalCompressedWords = inclusive;
        if (a.outSize) {
          a.outSize[b] = ansOffsetInArchive(a.floatType, size) + ansOverhead(nb) + 2u * inclusive;
        }
      }
    }
    // per-block word counts and start offsets (GpuANSEncode.cuh:595-608)
    uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(nb));
    const uint32_t blk = tile * kBlocksPerTile + lane;
    if (lane < kBlocksPerTile && blk < nb) {
      const uint32_t begin = blk * kBlockSize;
      const uint32_t bn = size - begin < kBlockSize ? size - begin : kBlockSize;
      blockWords[blk] = make_uint2((bn << 16) | sh->words[lane], exclusive + (incl - myPadded));
    }
    if (tile == numTiles - 1 && (nb & 1u) && lane == kBlocksPerTile) {
      blockWords[nb] = make_uint2(0u, 0u);  // alignment pad entry
    }
  }
  __syncthreads();

  if (haveBlock) {
    const uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
    const uint4* src = (const uint4*)stage;
    uint4* dst = (uint4*)(ans + ansOverhead(nb) + 2u * (size_t)(sh->tileBase + sh->localOff[hw]));
    for (uint32_t i = hl; i < vecs; i += 32u) dst[i] = src[i];
  }
}

}  // namespace dgpu
