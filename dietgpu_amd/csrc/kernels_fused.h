// Float compress in ONE kernel and with ONE read of the input: split + count + normalise + rANS encode + ordered
// compaction (k_float_compress_fused).
//
// What it computes is what k_float_histogram followed by k_ans_encode compute -- the reference's splitFloat with its
// fused histogram (dietgpu/float/GpuFloatCompress.cuh:144, 280-365), ansCalcWeights and the three encode kernels
// (dietgpu/ans/GpuANSEncode.cuh:49-211, 301-672) -- byte for byte.  The two-kernel path reads every float word twice:
// the table of an element needs the histogram of the whole element before its first symbol can be coded.  Here a
// tile's compressed (exponent) bytes stay in LDS between the two:
//
//   A  a workgroup (4 wavefronts = 8 blocks = one tile of 32 Ki words) requests its whole 64 KiB of input at once
//      (16 x 16 bytes per lane in flight: with three workgroups per CU the registers are there), splits the words
//      (ChunkSource<FT>::splitStore: the non-compressed bytes go straight into the archive), keeps the tile's 32 KiB
//      of compressed bytes in LDS and counts them into LDS bins (16 lane slots per bin, in the space the bitstream
//      stage takes later);
//   B  it publishes the tile's 256 counts (write-through; elements of many tiles: atomics into 256 counters) and bumps
//      the element's arrival counter; the workgroup that sees the element's last arrival normalises it on the spot
//      (normalizeElement, the code the histogram kernel ends with), stores the encoder table write-through and raises
//      the element's ready bit; everybody else polls that bit;
//   C  table into LDS, then the row loops of k_ans_encode (encodeRows, kLdsSrc) over the kept bytes -- no global load
//      in the loop at all;
//   D  ordered compaction exactly as in k_ans_encode: aggregate descriptor, decoupled look-back, 16-byte copy-out.
//
// 53 KiB of LDS per workgroup = three workgroups per CU: measured on MI355X the float encoder loses 2 % at that
// residency (profiles/r06_ab_encoder_residency_bf16.txt: 87.2 us at six workgroups per CU, 88.7 at three), the
// histogram kernel's 46 us and its 268 MB of reads are what is saved.
//
// Tickets are ELEMENT-major (ticket = element * T + tile) on a static persistent map, so the T tiles of an element are
// in flight together -- the host takes this path only when T <= the resident workgroups.
//
// Progress whatever is resident (another kernel, or another process, may hold CUs for as long as it likes -- e.g. a
// second fused kernel whose workgroups wait in B themselves):
//   * tile claim words as in k_ans_encode: a workgroup makes sure the previous tile of its element is claimed and
//     processes it first if it is not, so a look-back only ever waits on tiles of running workgroups;
//   * COUNT claim words: whoever counts a tile has claimed its count first.  A workgroup that has waited in B for
//     helpAfterPolls polls looks for tiles of the element whose count nobody has claimed, claims one and counts it
//     (a read-only pass over that tile's input, nothing kept) -- so every tile of an element somebody waits for is
//     being counted by a running workgroup, the last arrival normalises, and B ends.  The late owner of such a tile
//     finds its count claimed and only splits and keeps.
// All hand-off words (claims, descriptors, arrival counters) are library-owned and zero at rest: the last workgroup
// to leave the kernel (an exit counter) puts them back to zero.
#pragma once

#include "kernels_encode.h"

namespace dgpu {

constexpr uint32_t kFusedReady = 0x80000000u;  // bit of arrive[b]: the element's table is in place
constexpr uint32_t kFusedNone = 0xffffffffu;
// u32 words between the arrival words of two elements: every workgroup that waits for an element's table polls that
// element's word, and the arrivals and the ready bit have to get through those polls -- with the words of 16 elements
// in one 64-byte line, 512 waiting workgroups took 58 us for what 256 did in 33 (profiles/r06_fused_small_calls.txt)
constexpr uint32_t kFusedArriveStride = 32;
constexpr uint32_t kFusedHistSlots = 16;
constexpr uint32_t kFusedMaxPartials = 64;  // elements of more tiles than this meet in atomic counters (FusedArgs::histAcc)

__host__ __device__ constexpr uint32_t fusedStageBytes(int P, uint32_t ft) {
  const uint32_t stage = kBlocksPerTile * encStageCap(P, true, ft) * 2u, bins = kNumSymbols * kFusedHistSlots * 4u;
  return stage > bins ? stage : bins;
}
__host__ __device__ constexpr uint32_t fusedLdsBytes(int P, uint32_t ft) {
  return 4096u                          // encoder table (normalisation scratch while no table is loaded)
      + 128u                            // tile bookkeeping
      + fusedStageBytes(P, ft)          // bitstream stage of the 8 blocks (phase A: the histogram bins)
      + kBlocksPerTile * kBlockSize;    // the tile's compressed bytes
}

struct FusedArgs {
  BatchView in;              // float words; every element has `size` of them
  BatchView out;             // archive base pointers
  uint32_t numInBatch;       // B
  uint32_t tiles;            // T: tiles per element (size == T * 8 * 4096)
  uint32_t size;             // words per element
  uint32_t numTickets;       // B * T
  uint4* encTable;           // [B][256] temp: written by the normalising workgroup, read by the element's tiles
  uint32_t* histParts;       // [B][T][256] temp: per-tile counts (T <= kFusedMaxPartials), else null
  uint32_t* histAcc;         // [B][accSets][256] library-owned, zero at rest (T > kFusedMaxPartials), else null
  uint32_t accSets;          // power of two: tile t adds into set t % accSets
  uint32_t* arrive;          // [B] x kFusedArriveStride words, library-owned, zero at rest
  uint32_t* claims;          // [numTickets]   "
  uint32_t* countClaims;     // [numTickets]   "
  uint64_t* tileDesc;        // [numTickets]   "
  uint32_t* exitCount;       // [1]            "
  uint16_t* spill;           // [gridDim.x][8][encSpillSlotWords(P)] temp
  uint32_t* outSize;         // [B] nullable
  uint32_t outCapacity;      // see EncodeArgs
  uint32_t useChecksum;      // float header only
  const uint32_t* checksum;  // [B] nullable
  uint32_t absentModulo;     // test hook (see EncodeArgs)
  uint32_t helpAfterPolls;   // polls of the ready bit before a waiting workgroup starts counting other tiles
  NormalizeArgs norm;        // hist = histParts, histParts = T (or histAcc), tableInKernel = 1, no tileDesc / claims
};

struct FusedShared {
  uint32_t tileLo;
  uint32_t tileBase;
  uint32_t flag;
  uint32_t help;
  uint32_t words[kBlocksPerTile];
  uint32_t localOff[kBlocksPerTile];
};
static_assert(sizeof(FusedShared) <= 128, "");

template <int P, uint32_t FT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_float_compress_fused(FusedArgs a) {
  static_assert(FT != 0u, "float types only: a raw-byte tile's worst-case stage leaves no room to keep its symbols");
  constexpr uint32_t kTB = kBlocksPerTile;
  constexpr uint32_t kCap = encStageCap(P, true, FT);
  constexpr uint32_t S = kFusedHistSlots;
  using Src = ChunkSource<FT>;
  constexpr uint32_t kChunkRows = Src::kRows, kChunks = kRowsPerBlock / kChunkRows, kCR = Src::kCompRegs;
  constexpr uint32_t kChunkSyms = kChunkRows * 32u;
  constexpr uint32_t kGroup = 8;  // chunks requested at once: 16 x 16 bytes in flight per lane
  static_assert(kChunks % kGroup == 0 && kCR * 4u * 32u == kChunkSyms, "");
  static_assert(kNormScratchWords * 4u <= 4096u, "the normalisation's scratch lives where the table goes");

  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint4* sTable = (uint4*)smem;
  FusedShared* sh = (FusedShared*)(smem + 4096);
  uint16_t* sStage = (uint16_t*)(smem + 4096 + 128);
  uint32_t* sBins = (uint32_t*)(smem + 4096 + 128);
  uint8_t* sKeep = smem + 4096 + 128 + fusedStageBytes(P, FT);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);

  uint16_t* stage = sStage + hw * kCap;
  const uint32_t stageLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)stage;
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)sTable;
  uint8_t* keep = sKeep + hw * kBlockSize;
  uint32_t* myBins = histMine<S>(sBins, tid);
  uint16_t* spillSlot = a.spill + ((size_t)blockIdx.x * kTB + hw) * encSpillSlotWords(P);

  const uint32_t B = a.numInBatch, T = a.tiles, size = a.size;
  const uint32_t nb = T * kTB;
  const uint32_t me = blockIdx.x + 1u;
  (void)B;

  if (a.absentModulo && blockIdx.x % a.absentModulo == 1u) {
    for (int i = 0; i < 150; ++i) __builtin_amdgcn_s_sleep(127);  // test hook: resident ~0.5 ms late
  }

  auto claimLoad = [&](const uint32_t* w) -> uint32_t { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto claimTry = [&](uint32_t* w) -> bool {
    uint32_t expected = 0;
    return __hip_atomic_compare_exchange_strong(w, &expected, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ---- A: the lane's share of tile (b, tile): split, keep, count.  kCountOnly: a read-only pass for another
  //      workgroup's tile (nothing stored, nothing kept)
  auto splitKeepCount = [&](uint32_t b, uint32_t tile, auto countOnly) {
    constexpr bool kCountOnly = decltype(countOnly)::value;
    Src src;
    src.init(a.in.ptr(b), a.out.ptr(b), size, tile * kTB + hw);
#pragma unroll 1
    for (uint32_t g = 0; g < kChunks; g += kGroup) {
      typename Src::Raw raw[kGroup];
#pragma unroll
      for (uint32_t j = 0; j < kGroup; ++j) raw[j] = src.load(g + j, hl);
#pragma unroll
      for (uint32_t j = 0; j < kGroup; ++j) {
        uint32_t comp[kCR];
        src.splitStore(raw[j], g + j, hl, comp, !kCountOnly);
        if constexpr (!kCountOnly) {
          uint8_t* dst = keep + (g + j) * kChunkSyms + hl * (kCR * 4u);
          if constexpr (kCR == 4) *(uint4*)dst = make_uint4(comp[0], comp[1], comp[2], comp[3]);
          else *(uint2*)dst = make_uint2(comp[0], comp[1]);
        }
#pragma unroll
        for (uint32_t q = 0; q < kCR; ++q) histAdd4<S>(myBins, comp[q]);
      }
    }
  };

  // ---- B: publish the counts of tile (b, tile) and arrive; the element's last arrival normalises it
  auto publishCounts = [&](uint32_t b, uint32_t tile) {
    __syncthreads();  // the LDS atomics of all four waves have been performed
    const uint32_t sum = histFold<S>(sBins, tid);
    if (a.histParts) {
      __hip_atomic_store(a.histParts + ((size_t)b * T + tile) * kNumSymbols + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (sum) {
      __hip_atomic_fetch_add(a.histAcc + ((size_t)b * a.accSets + (tile & (a.accSets - 1u))) * kNumSymbols + tid, sum, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and performed
    __syncthreads();
    if (tid == 0) {
      const uint32_t prev = __hip_atomic_fetch_add(a.arrive + (size_t)b * kFusedArriveStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh->flag = (prev + 1u == T) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t last = sh->flag;
    __syncthreads();
    if (last) {  // uniform
      normalizeElement<true>(a.norm, b, (uint32_t*)sTable);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table, header and pdf (write-through where read in this kernel) performed
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.arrive + (size_t)b * kFusedArriveStride, kFusedReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // ---- a tile of element b whose count nobody has claimed: claims it (wave 0; result through sh->help)
  auto claimUncountedTile = [&](uint32_t b) {
    if (wave == 0) {
      uint32_t found = kFusedNone;
      for (uint32_t base = 0; base < T && found == kFusedNone; base += 64u) {
        const uint32_t j = base + lane;
        const uint32_t c = j < T ? claimLoad(a.countClaims + (size_t)b * T + j) : 1u;
        uint64_t m = __ballot(c == 0u);
        while (m != 0ull && found == kFusedNone) {
          const uint32_t l = (uint32_t)__ffsll((unsigned long long)m) - 1u;
          m &= m - 1ull;
          bool ok = false;
          if (lane == l) ok = claimTry(a.countClaims + (size_t)b * T + base + l);
          if (__ballot(ok) != 0ull) found = base + l;
        }
      }
      if (lane == 0) sh->help = found;
    }
    __syncthreads();
    const uint32_t j = sh->help;
    __syncthreads();
    return j;
  };

  // ---- wait for the table of element b (counting unclaimed tiles of b when it takes long) and fetch it
  auto waitAndFetchTable = [&](uint32_t b) {
    uint32_t polls = 0, pause = 1;  // pause between two polls: 1, 2, 4, ... 16 x 512 cycles (0.2 us ... 3.4 us)
    bool helping = true;
    for (;;) {
      if (tid == 0) sh->flag = __hip_atomic_load(a.arrive + (size_t)b * kFusedArriveStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const uint32_t v = sh->flag;
      __syncthreads();
      if (v & kFusedReady) break;
      if (helping && ++polls >= a.helpAfterPolls) {
        const uint32_t j = claimUncountedTile(b);
        if (j != kFusedNone) {
          histZero<S>(sBins, tid);
          __syncthreads();
          splitKeepCount(b, j, std::true_type{});
          publishCounts(b, j);
          continue;
        }
        helping = false;  // every count of this element is claimed: its owners are running
      }
      for (uint32_t q = 0; q < pause; ++q) __builtin_amdgcn_s_sleep(8);
      pause = pause < 16u ? pause * 2u : 16u;
    }
    const uint64_t* src = (const uint64_t*)(a.encTable + (size_t)b * kNumSymbols + tid);
    const uint64_t lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sTable[tid] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    __syncthreads();
  };

  // ---- tickets: the static map with claim words of k_ans_encode, element-major
  bool firstOwned = false;
  if (tid == 0 && blockIdx.x < a.numTickets) firstOwned = claimTry(a.claims + blockIdx.x);
  for (uint32_t t = blockIdx.x + (tid + 1u) * gridDim.x; t < a.numTickets; t += 256u * gridDim.x) (void)claimTry(a.claims + t);

  for (uint32_t ticket0 = blockIdx.x; ticket0 < a.numTickets; ticket0 += gridDim.x) {
    const uint32_t b = ticket0 / T;
    const uint32_t tile0 = ticket0 - b * T;
    if (tid == 0) {
      const bool mine = (ticket0 == blockIdx.x) ? firstOwned : (claimLoad(a.claims + ticket0) == me);
      uint32_t lo = tile0 + 1u;  // empty range: the tile was taken over by somebody else
      if (mine) {
        lo = tile0;
        while (lo > 0u) {
          uint32_t* w = a.claims + (ticket0 - tile0 + (lo - 1u));
          uint32_t p = claimLoad(w);
          for (int spin = 0; p == 0u && spin < 4; ++spin) {  // give a running owner's claim time to land
            __builtin_amdgcn_s_sleep(32);
            p = claimLoad(w);
          }
          if (p != 0u) break;
          if (!claimTry(w)) break;
          --lo;
        }
      }
      sh->tileLo = lo;
    }
    __syncthreads();
    const uint32_t tileLo = sh->tileLo;
    __syncthreads();
    if (tileLo > tile0) continue;

    for (uint32_t tile = tileLo; tile <= tile0; ++tile) {
      const uint32_t ticket = ticket0 - tile0 + tile;
      // (the copy-out of the previous tile has read the stage: every wave is past it at this barrier)
      if (tid == 0) sh->flag = claimTry(a.countClaims + ticket) ? 1u : 0u;
      histZero<S>(sBins, tid);
      __syncthreads();
      const bool iCount = sh->flag != 0u;
      uint8_t* archive = a.out.ptr(b);
      uint8_t* ans = archive + ansOffsetInArchive(FT, size);

      splitKeepCount(b, tile, std::false_type{});
      if (iCount) publishCounts(b, tile);  // (else: somebody counted this tile while it had no owner; the bins are dropped)
      else __syncthreads();

      if (tile == 0 && tid == 0) {
        // GpuFloatHeader (GpuFloatCompress.cuh:325-337); `size` is a whole number of tiles: the planes need no padding
        FloatHeader h;
        h.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
        h.size = size;
        h.options = FT | (a.useChecksum ? 0x10u : 0u);
        h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
        *(FloatHeader*)archive = h;
      }

      waitAndFetchTable(b);

      // ---- C: the row loops over the kept bytes
      const uint32_t block = tile * kTB + hw;
      Src src;
      src.init(a.in.ptr(b), archive, size, block);
      uint32_t state, spilled = 0;
      bool overrun = false;
      const uint32_t words = encodeRows<P, FT, true, true, false, false, false, true>(
          src, kBlockSize, kRowsPerBlock, tableLds, stageLds, keep, hl, upper, spillSlot, spilled, state, overrun);

      // ---- D: as k_ans_encode
      ((uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl] = state;
      {
        const uint32_t padded = roundUp(words, kBlockAlignWords);
        if (words + hl < padded) stage[words + hl] = 0;
      }
      if (hl == 0) sh->words[hw] = (spilled + words) | (overrun ? 0x80000000u : 0u);
      ldsBarrier();

      if (wave == 0) {
        const uint32_t mine = (lane < kTB) ? sh->words[lane] : 0u;
        const bool tileFailed = __ballot((mine & 0x80000000u) != 0u) != 0ull;
        const uint32_t myWords = mine & 0x7fffffffu;
        const uint32_t myPadded = roundUp(myWords, kBlockAlignWords);
        const uint32_t incl = waveInclusiveScan(myPadded, lane);
        const uint32_t aggregate = __shfl(incl, kTB - 1, 64);
        if (lane < kTB) sh->localOff[lane] = incl - myPadded;

        uint64_t* desc = a.tileDesc + (size_t)(ticket0 - tile0);
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], kDescAggregate | (tileFailed ? kDescFailed : 0ull) | (uint64_t)aggregate,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool failed = tileFailed;
        const uint32_t exclusive = lookBackExclusive(desc, tile, lane, failed, T > 16u);
        const uint32_t inclusive = exclusive + aggregate;
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], kDescInclusive | (failed ? kDescFailed : 0ull) | (uint64_t)inclusive,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sh->tileBase = exclusive;
          if (tile == T - 1u) {
            // complete the header (GpuANSEncode.cuh:533-566)
            ((AnsHeader*)ans)->totalCompressedWords = failed ? 0u : inclusive;
            if (failed) ((AnsHeader*)ans)->magicAndVersion = 0u;
            if (a.outSize) a.outSize[b] = failed ? 0u : ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2u * inclusive;
          }
        }
        // per-block word counts and start offsets (GpuANSEncode.cuh:595-608); nb is even: no pad entry
        uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(nb));
        if (lane < kTB) blockWords[tile * kTB + lane] = make_uint2((kBlockSize << 16) | myWords, exclusive + (incl - myPadded));
      }
      ldsBarrier();

      {
        const uint64_t dataOff = (uint64_t)ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2ull * (sh->tileBase + sh->localOff[hw]);
        uint4* dst = (uint4*)(archive + dataOff);
        const uint64_t room = (uint64_t)a.outCapacity > dataOff ? ((uint64_t)a.outCapacity - dataOff) / 16u : 0u;
        uint32_t fit = room > 0xffffffffull ? 0xffffffffu : (uint32_t)room;
        if (spilled) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own spill stores
          const uint4* sp = (const uint4*)spillSlot;
          const uint32_t sv = spilled / kBlockAlignWords;
          const uint32_t svFit = sv < fit ? sv : fit;
          for (uint32_t i = hl; i < svFit; i += 32u) streamStore<kNtEncStores>(&dst[i], sp[i]);
          dst += sv;
          fit -= svFit;
        }
        uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
        vecs = vecs < fit ? vecs : fit;
        const uint4* s4 = (const uint4*)stage;
        for (uint32_t i = hl; i < vecs; i += 32u) streamStore<kNtEncStores>(&dst[i], s4[i]);
      }
      __syncthreads();  // stage and kept bytes are free again
    }
  }

  // ---- the last workgroup out puts the hand-off words back to zero
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(a.exitCount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh->flag = (prev + 1u == gridDim.x) ? 1u : 0u;
  }
  __syncthreads();
  if (sh->flag) {  // uniform
    for (uint32_t i = tid; i < a.numTickets; i += 256u) {
      __hip_atomic_store(a.tileDesc + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.claims + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.countClaims + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t i = tid; i < a.numInBatch; i += 256u) __hip_atomic_store(a.arrive + (size_t)i * kFusedArriveStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) __hip_atomic_store(a.exitCount, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace dgpu
