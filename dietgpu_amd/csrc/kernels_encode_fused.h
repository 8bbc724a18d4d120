// Fused statistics + rANS encode: ONE read of the input.
//
// The two-kernel path (k_histogram / k_float_histogram, then k_ans_encode) reads
// every input byte twice: once to count symbols, once to code them -- the table
// of an element needs the histogram of the whole element first.  The reference
// has the same structure for raw bytes and writes + re-reads a whole exponent
// plane for floats (GpuFloatCompress.cuh:144,352-364).  Here the symbols of a
// tile stay ON CHIP between the two passes:
//
//   * a tile = 8 blocks = 32 Ki symbols; each of the workgroup's 4 wavefronts owns
//     two blocks = 8192 symbols = 32 VGPRs per lane.  The register file of a CU
//     is 512 KiB -- three times its LDS -- and the coder's working set leaves
//     room for those 32 registers at 4 wavefronts per SIMD.
//   * software pipeline per workgroup: while tile i is being encoded from its
//     registers (chunk by chunk through the same 512-byte LDS ring as in
//     k_ans_encode), the input of tile i+1 is loaded, split (floats: the
//     non-compressed bytes go straight to the archive), counted into LDS bins and
//     parked in the registers tile i has just vacated.  HBM latency and the
//     LDS atomics run under the VALU-bound row steps of the same wavefront.
//   * after its chunk loop the workgroup publishes the partial histogram of tile
//     i+1 (write-through) and bumps the element's arrival counter; the workgroup
//     that sees the last arrival normalises the element on the spot
//     (normalizeElement, the same code the histogram kernel ends with) and raises
//     the element's ready flag.  Before encoding a tile a workgroup waits for
//     that flag and fetches the 4 KiB table.
//   * tickets are drawn dynamically, ELEMENT-major (all tiles of an element are
//     drawn back to back, from one of up to 8 counters = classes of elements), so
//     the tiles of an element are in flight together and a workgroup only ever
//     waits for tiles that running workgroups hold: see DESIGN.md section 4.3 for
//     the progress argument (it needs >= `tiles` resident workgroups per class,
//     which is why the host only takes this path for elements of <= 32 tiles).
//
// The host takes this path for uniform batches (all elements the same size, a
// whole number of tiles, 16-byte aligned inputs, no caller-supplied histogram);
// everything else runs the two-kernel path.  Archives are byte-identical.
#pragma once

#include "kernels_encode.h"

namespace dgpu {

constexpr uint32_t kFusedMaxTiles = 32;     // tiles per element the fused path accepts (progress argument)
constexpr uint32_t kFusedMaxClasses = 8;    // ticket counters (one per XCD: workgroup w runs on XCD w % 8)
constexpr uint32_t kFusedTicketStride = 32; // u32 words between counters (128 bytes)
constexpr uint32_t kFusedNone = 0xffffffffu;

// LDS histogram slots per bin (see kernels_stats.h): 8 for the float types; raw bytes need a larger
// bitstream stage, 4 slots keep the workgroup at 40 KiB = 4 workgroups per CU
__host__ __device__ constexpr uint32_t fusedHistSlots(uint32_t ft) { return ft ? 8u : 4u; }
// Stage words per block.  Raw bytes: Zipf-like data produces ~1350 words per block and the flush check
// wants 256 words of headroom.
__host__ __device__ constexpr uint32_t fusedStageCap(int P, uint32_t ft) {
  return ft == 0 ? 1664u : encStageCap(P, true, ft);
}
__host__ __device__ constexpr uint32_t fusedLdsBytes(int P, uint32_t ft) {
  return 4096u + 128u + kBlocksPerTile * fusedStageCap(P, ft) * 2u + kBlocksPerTile * 512u + 512u +
      kNumSymbols * fusedHistSlots(ft) * 4u;
}

struct FusedArgs {
  BatchView in;            // raw bytes or float words; every element has `size` symbols
  BatchView out;           // archive base pointers
  uint32_t numInBatch;     // B
  uint32_t tiles;          // T: tiles per element (size == T * 8 * 4096)
  uint32_t size;           // symbols per element
  uint32_t numClasses;     // C <= kFusedMaxClasses ticket counters; element e belongs to class e % C
  uint4* encTable;         // [B][256] temp: written by the normalising workgroup, read by the element's tiles
  uint32_t* histParts;     // [B][T][256] temp: per-tile partial histograms
  uint64_t* tileDesc;      // [B][T] temp: look-back descriptors (each tile resets its own before it arrives)
  uint32_t* arrive;        // [B] library-owned, zero at rest
  uint32_t* ready;         // [B] library-owned: == epoch once the element's table is in place
  uint32_t epoch;          // this call's value for `ready`
  uint32_t* tickets;       // library-owned, zero at rest: kFusedMaxClasses counters + the exit counter
  uint16_t* spill;         // [gridDim.x][8][encSpillSlotWords(P)]
  uint32_t* outSize;       // [B] nullable
  uint32_t useChecksum;    // float header only
  const uint32_t* checksum;  // [B] nullable (float header only)
  uint32_t absentModulo;   // test hook (see EncodeArgs)
  uint32_t staggerSleeps;  // start delay, in s_sleep(127) units, per dispatch round of the grid (blockIdx / 256)
  NormalizeArgs norm;      // for normalizeElement: hist = histParts, histParts = T, inKernelConsumer = 1
};

struct FusedShared {
  uint32_t ticket;
  uint32_t flag;
  uint32_t tileBase;
  uint32_t pad;
  uint32_t words[kBlocksPerTile];
  uint32_t localOff[kBlocksPerTile];
};
static_assert(sizeof(FusedShared) <= 128, "");

template <int P, uint32_t FT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ans_encode_fused(FusedArgs a) {
  constexpr uint32_t kTB = kBlocksPerTile;
  constexpr uint32_t kCap = fusedStageCap(P, FT);
  constexpr uint32_t S = fusedHistSlots(FT);
  using Src = ChunkSource<FT>;
  constexpr uint32_t kChunkRows = Src::kRows;
  constexpr uint32_t kChunks = kRowsPerBlock / kChunkRows;
  constexpr uint32_t kCR = Src::kCompRegs;
  static_assert(kNormScratchWords * 4u <= kBlocksPerTile * 512u && kChunks * kCR == 32, "a wavefront retains its two blocks in 32 registers per lane");

  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint4* sTable = (uint4*)smem;
  FusedShared* sh = (FusedShared*)(smem + 4096);
  uint16_t* sStage = (uint16_t*)(smem + 4096 + 128);
  uint8_t* sRing = smem + 4096 + 128 + kTB * kCap * 2u;
  uint32_t* sBins = (uint32_t*)(sRing + kTB * 512u + 512u);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);
  const uint32_t laneMaskLt = (1u << hl) - 1u;

  uint16_t* stage = sStage + hw * kCap;
  const uint32_t stageBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)stage;
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)sTable;
  const uint32_t dummyAddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(sRing + kTB * 512u) + tid * 2u;
  uint8_t* ring = sRing + hw * 512u;
  uint16_t* spillSlot = a.spill + ((size_t)blockIdx.x * kTB + hw) * encSpillSlotWords(P);
  uint32_t* myBins = histMine<S>(sBins, tid);

  const uint32_t B = a.numInBatch, T = a.tiles, C = a.numClasses;
  const uint32_t size = a.size;
  const uint32_t nb = T * kTB;

  if (a.absentModulo && blockIdx.x % a.absentModulo == 1u) {
    for (int i = 0; i < 150; ++i) __builtin_amdgcn_s_sleep(127);  // test hook: becomes resident ~0.5 ms late
  }

  // Stagger: the workgroups that share a CU (blockIdx i, i + 256, i + 512, ...) start a fraction of a
  // tile time apart, so that they sit in different phases: while one waits for a table or a look-back
  // the others keep the SIMDs busy.
  {
    // bit 31 of the knob: phase class from a hash of the workgroup index instead of blockIdx / 256
    const uint32_t q = (a.staggerSleeps >> 31) ? ((blockIdx.x * 2654435761u) >> 30) : (blockIdx.x >> 8);
    for (uint32_t i = 0, n = q * (a.staggerSleeps & 0xffffu); i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  histZero<S>(sBins, tid);

  // ---- tickets: element-major, one counter per class of elements -------------------------
  uint32_t myClass = blockIdx.x % C;  // thread 0's state
  uint32_t classesTried = 0;
  auto drawBroadcast = [&]() -> uint32_t {
    if (tid == 0) {
      uint32_t t = kFusedNone;
      while (classesTried < C) {
        const uint32_t k = __hip_atomic_fetch_add(a.tickets + myClass * kFusedTicketStride, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t elemsInClass = (B - myClass + C - 1u) / C;
        if (k < elemsInClass * T) {
          const uint32_t e = k / T;
          t = (myClass + C * e) * T + (k - e * T);
          break;
        }
        myClass = (myClass + 1u) % C;  // this class is exhausted: help the next one
        ++classesTried;
      }
      sh->ticket = t;
    }
    ldsBarrier();
    const uint32_t t = sh->ticket;
    ldsBarrier();  // everybody has read it before it is written again
    return t;
  };

  // ---- histogram of the lane's symbols of one chunk ---------------------------------------
  auto histChunk = [&](const uint32_t (&comp)[kCR]) {
#pragma unroll
    for (uint32_t j = 0; j < kCR; ++j) histAdd4<S>(myBins, comp[j]);
  };

  // ---- publish the partial histogram of tile (b, t); the last tile of the element to get
  //      here normalises it and raises its ready flag --------------------------------------
  auto publishHistogram = [&](uint32_t b, uint32_t t) {
    ldsBarrier();  // the LDS atomics of all four waves have been performed
    const uint32_t sum = histFold<S>(sBins, tid);
#pragma unroll
    for (uint32_t k = 0; k < S / 4u; ++k) ((uint4*)(sBins + tid * S))[k] = make_uint4(0, 0, 0, 0);  // bins back to zero
    __hip_atomic_store(a.histParts + ((size_t)b * T + t) * kNumSymbols + tid, sum, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // write-through
    // this tile's look-back descriptor back to "nothing published" (temp memory is reused between calls);
    // the tiles that read it do so after the element's ready flag, which is after this arrival
    if (tid == 0) __hip_atomic_store(a.tileDesc + (size_t)b * T + t, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and performed
    __syncthreads();
    if (tid == 0) {
      const uint32_t prev = __hip_atomic_fetch_add(a.arrive + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh->flag = (prev + 1u == T) ? 1u : 0u;
    }
    ldsBarrier();
    const uint32_t last = sh->flag;
    ldsBarrier();
    if (last) {  // uniform
      if (tid == 0) __hip_atomic_store(a.arrive + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero at rest
      normalizeElement<true>(a.norm, b, (uint32_t*)sRing);  // the rings are idle between chunk loops
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table (write-through) performed
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.ready + b, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // ---- wait for the table of element b and fetch it ------------------------------------------
  auto fetchTable = [&](uint32_t b) {
    if (tid == 0) {
      uint32_t spins = 0;
      while (__hip_atomic_load(a.ready + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) __builtin_trap();  // ~seconds: the progress argument has been violated
      }
    }
    ldsBarrier();
    const uint64_t* src = (const uint64_t*)(a.encTable + (size_t)b * kNumSymbols + tid);
    const uint64_t lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sTable[tid] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    ldsBarrier();
  };

  // retained symbols: R[c] = this lane's symbols of chunk c of its block (tile being / about to be encoded)
  uint32_t R[kChunks][kCR];

  // ================= prologue: the first tile is loaded with nothing to overlap =================
  uint32_t nxt = drawBroadcast();
  if (nxt != kFusedNone) {
    const uint32_t b = nxt / T, t = nxt - b * T;
    Src src;
    src.init(a.in.ptr(b), a.out.ptr(b), size, t * kTB + hw);
    typename Src::Raw raw = src.load(0, hl);
#pragma unroll
    for (uint32_t c = 0; c < kChunks; ++c) {
      typename Src::Raw rawNext = raw;
      if (c + 1 < kChunks) rawNext = src.load(c + 1, hl);
      src.splitStore(raw, c, hl, R[c]);
      histChunk(R[c]);
      raw = rawNext;
    }
    publishHistogram(b, t);
  }

  // ================= steady state ================================================================
  while (nxt != kFusedNone) {
    const uint32_t b = nxt / T, tile = nxt - b * T;
#ifdef DGPU_PHASE_TIMING
    const uint32_t phaseSlot = nxt;
    if (threadIdx.x == 0 && g_phaseBuf) g_phaseBuf[(size_t)phaseSlot * 8 + 7] = blockIdx.x;
#endif
    DGPU_PHASE(0);
    fetchTable(b);
    DGPU_PHASE(1);
    nxt = drawBroadcast();
    const bool haveNext = nxt != kFusedNone;
    const uint32_t bn = haveNext ? nxt / T : 0u, tn = haveNext ? nxt - bn * T : 0u;

    uint8_t* archive = a.out.ptr(b);
    uint8_t* ans = archive + ansOffsetInArchive(FT, size);
    const uint32_t block = tile * kTB + hw;

    if (FT != 0 && tile == 0 && tid == 0) {
      // GpuFloatHeader (GpuFloatCompress.cuh:325-337); `size` is a multiple of 32768: no plane padding to zero
      FloatHeader h;
      h.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
      h.size = size;
      h.options = FT | (a.useChecksum ? 0x10u : 0u);
      h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
      *(FloatHeader*)archive = h;
    }

    Src srcN;
    srcN.init(a.in.ptr(bn), a.out.ptr(bn), size, tn * kTB + hw);
    typename Src::Raw raw;
    if (haveNext) raw = srcN.load(0, hl);

    // ---- encode tile (b, tile) from R while tile (bn, tn) takes its place ----
    uint32_t state = kStartState;
    uint32_t outOff = 0;
    uint32_t spilled = 0;

    auto makeRoom = [&]() {
      const uint32_t o0 = __builtin_amdgcn_readlane(outOff, 0);
      const uint32_t o1 = __builtin_amdgcn_readlane(outOff, 32);
      if ((o0 > o1 ? o0 : o1) + kFlushRows * 32u <= kCap) return;  // wave-uniform
      uint32_t nvec = outOff >> 3;
      if (spilled + nvec * 8u > encSpillSlotWords(P)) nvec = 0;
      uint4* dst = (uint4*)(spillSlot + spilled);
      for (uint32_t i = hl; i < nvec; i += 32u) {
        const u32x4e v = *(const LdsU4e*)(uintptr_t)(stageBase + 16u * i);
        dst[i] = make_uint4(v.x, v.y, v.z, v.w);
      }
      const uint32_t rem = outOff & 7u;
      uint16_t tmp = 0;
      if (hl < rem) tmp = *(const LdsU16e*)(uintptr_t)(stageBase + 2u * (nvec * 8u + hl));
      if (hl < rem) *(LdsU16e*)(uintptr_t)(stageBase + 2u * hl) = tmp;
      spilled += nvec * 8u;
      outOff = rem;
    };
    // the row step of k_ans_encode (encodeRows, full blocks): branch-free, see kernels_encode.h
    auto stepFull = [&](const uint4 e) {
      const bool write = state >= e.x;
      const uint64_t vote = __ballot(write);
      const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
      const uint32_t idx = outOff + __popc(vh & laneMaskLt);
      const uint32_t addr = write ? stageBase + 2u * idx : dummyAddr;
      *(LdsU16e*)(uintptr_t)addr = (uint16_t)state;
      state = write ? (state >> kEncodedBits) : state;
      const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
      state = __umul24(div, e.w) + state + e.z;
      outOff += __popc(vh);
    };

#pragma unroll
    for (uint32_t c = 0; c < kChunks; ++c) {
      // the symbols of chunk c go to the ring ...
      if constexpr (kCR == 4) *(uint4*)(ring + hl * 16u) = make_uint4(R[c][0], R[c][1], R[c][2], R[c][3]);
      else *(uint2*)(ring + hl * 8u) = make_uint2(R[c][0], R[c][1]);
      // ... and their registers take chunk c of the next tile
      if (haveNext) {  // uniform
        typename Src::Raw rawNext = raw;
        if (c + 1 < kChunks) rawNext = srcN.load(c + 1, hl);
        srcN.splitStore(raw, c, hl, R[c]);
        histChunk(R[c]);
        raw = rawNext;
      }
      constexpr int kAhead = DGPU_ENC_AHEAD;
      constexpr int kSymAhead = DGPU_ENC_SYM_AHEAD;
      auto symAddr = [&](int r) -> uint32_t {
        uint32_t t = tableLds + ((uint32_t)ring[r * 32 + hl] << 4);
        asm volatile("" : "+v"(t));
        return t;
      };
      uint32_t toff[kSymAhead];
#pragma unroll
      for (int r = 0; r < kSymAhead; ++r) toff[r] = symAddr(r);
      uint4 e[kAhead];
#pragma unroll
      for (int r = 0; r < kAhead; ++r) e[r] = ldsTableEntry(toff[r]);
#pragma unroll
      for (int r = 0; r < (int)kChunkRows; ++r) {
        if (r % kFlushRows == 0) makeRoom();
        const uint4 cur_e = e[r % kAhead];
        if (r + kAhead < (int)kChunkRows) e[r % kAhead] = ldsTableEntry(toff[(r + kAhead) % kSymAhead]);
        if (r + kSymAhead < (int)kChunkRows) toff[r % kSymAhead] = symAddr(r + kSymAhead);
        stepFull(cur_e);
      }
    }
    uint32_t words = outOff;
    DGPU_PHASE(2);

    // ---- finish tile (b, tile), part 1: states, block sizes, and the tile's AGGREGATE descriptor --
    // published before anything that can take long (the histogram hand-off below may include the
    // normalisation of a whole element): the tiles behind this one only need the aggregate
    {
      ((uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl] = state;
      const uint32_t padded = roundUp(words, kBlockAlignWords);
      if (words + hl < padded) stage[words + hl] = 0;
    }
    if (hl == 0) sh->words[hw] = spilled + words;
    ldsBarrier();
    uint32_t myPadded = 0, incl = 0, aggregate = 0;  // wave 0
    uint64_t* desc = a.tileDesc + (size_t)b * T;
    if (wave == 0) {
      myPadded = (lane < kTB) ? roundUp(sh->words[lane], kBlockAlignWords) : 0u;
      incl = waveInclusiveScan(myPadded, lane);
      aggregate = __shfl(incl, kTB - 1, 64);
      if (lane < kTB) sh->localOff[lane] = incl - myPadded;
      if (lane == 0) {
        __hip_atomic_store(&desc[tile], kDescAggregate | (uint64_t)aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    if (haveNext) publishHistogram(bn, tn);
    DGPU_PHASE(3);

    // ---- part 2: ordered compaction (look-back over the preceding tiles) and copy-out, as in k_ans_encode ----
    if (wave == 0) {
      uint32_t exclusive = 0;
      int base = (int)tile - 1;
      while (base >= 0) {
        const int idx = base - (int)lane;
        uint64_t d = kDescInclusive;
        if (idx >= 0) {
          uint32_t spins = 0;
          do {
            d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((d >> 62) == 0) {
              __builtin_amdgcn_s_sleep(1);
              if (++spins > (1u << 26)) __builtin_trap();
            }
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
        __hip_atomic_store(&desc[tile], kDescInclusive | (uint64_t)inclusive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh->tileBase = exclusive;
        if (tile == T - 1) {
          ((AnsHeader*)ans)->totalCompressedWords = inclusive;
          if (a.outSize) a.outSize[b] = ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2u * inclusive;
        }
      }
      uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(nb));
      const uint32_t blk = tile * kTB + lane;
      if (lane < kTB) blockWords[blk] = make_uint2((kBlockSize << 16) | sh->words[lane], exclusive + (incl - myPadded));
      // nb is a multiple of 8: no odd blockWords pad entry
    }
    ldsBarrier();
    DGPU_PHASE(4);
    {
      uint4* dst = (uint4*)(ans + ansOverhead(nb) + 2u * (size_t)(sh->tileBase + sh->localOff[hw]));
      if (spilled) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4* sp = (const uint4*)spillSlot;
        const uint32_t sv = spilled / kBlockAlignWords;
        for (uint32_t i = hl; i < sv; i += 32u) streamStore<DGPU_NT_ENC_STORES != 0>(&dst[i], sp[i]);
        dst += sv;
      }
      const uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
      const uint4* s4 = (const uint4*)stage;
      for (uint32_t i = hl; i < vecs; i += 32u) streamStore<DGPU_NT_ENC_STORES != 0>(&dst[i], s4[i]);
    }
    DGPU_PHASE(5);
    ldsBarrier();  // sh->words / stage are written again by the next tile
  }

  // ---- the last workgroup out puts the ticket counters back to zero --------------------------
  if (tid == 0) {
    uint32_t* done = a.tickets + kFusedMaxClasses * kFusedTicketStride;
    const uint32_t prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == gridDim.x) {
      for (uint32_t c = 0; c < kFusedMaxClasses; ++c) {
        __hip_atomic_store(a.tickets + c * kFusedTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace dgpu
