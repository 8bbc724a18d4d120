// dietgpu_amd: batches of SINGLE-block elements (every element <= 4096 symbols) -- two elements per wavefront.
//
// The format interleaves 32 rANS states per 4 KiB block (GpuANSUtils.cuh:62-65); the coders put two blocks in a
// wave64, lanes 0-31 and 32-63.  In a batch whose elements have one block each, the general kernels can only fill
// one half of a wave (the other half has no block of ITS element to take) and every instruction is issued for 32
// useful lanes: measured on 32768 x 4 Ki bf16 both coders were instruction-issue bound (PMC: encode 101 M
// wave-instructions = 165 of its 193 us at one instruction per 4 cycles per SIMD).  These two kernels give the
// upper half of the wave the NEXT element of the batch instead: element 2p in lanes 0-31, element 2p+1 in lanes
// 32-63, each half with its own table / LUT, stage, ring and archive.  Everything that is wave-uniform in the
// general kernels (element, sizes, pointers, validity) is per-half here; the row loops (encodeRows, decodeBlock)
// are the general kernels' own.  A workgroup is one wavefront: no barriers, no look-back (an element is one tile).
//
// Reference semantics: ansEncodeBatch / ansDecodeBatch on elements of one block (GpuANSEncode.cuh:429-672,
// GpuANSDecode.cuh:299-403); archives are byte-identical to the general kernels' (tests/test_gpu_parity.py::
// test_ragged_batches_of_small_elements, test_whole_block_elements_in_every_tile_variant).
#pragma once

#include "kernels_decode.h"
#include "kernels_encode.h"

namespace dgpu {

// inclusive scan over each HALF of the wavefront (rows 0-1 and rows 2-3 separately)
__device__ __forceinline__ uint32_t halfInclusiveScanDpp(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
  return v;
}
__device__ __forceinline__ uint32_t halfInclusiveMaxScanDpp(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
  return v;
}

// LDS fence for a single-wavefront workgroup: its LDS operations execute in order, this only stops the compiler
// from moving accesses made through differently typed pointers across the phase boundary
__device__ __forceinline__ void pairLdsFence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------
// Statistics of single-block elements: ONE WAVEFRONT counts and normalises an element (k_histogram /
// k_float_histogram + normalizeElement give it a 256-thread workgroup: six barriers and a quarter of the elements in
// flight per CU for 8 KiB of input).  The wave loads the whole element with all its loads in flight, counts into its
// own 8-slot bins, then lane l folds and normalises symbols 4 l .. 4 l + 3 -- the layout normDeficitTrips works on --
// and writes the pdf table and the static header fields into the archive (what normalizeElement writes; the encoder
// table is k_ans_encode_pair's business).  No barriers: the four waves of a workgroup are four elements.
// Semantics: ansHistogramBatch + ansCalcWeights (GpuANSStatistics.cuh:43-367) per element.  Raw bytes and 16-bit floats
// (float32 -- half as many symbols per byte of input -- measured slower this way and keeps the workgroup).
constexpr uint32_t kSingleStatSlots = 8;
constexpr uint32_t kSingleStatWaves = 4;  // elements (= wavefronts) per workgroup
__host__ __device__ constexpr uint32_t statSingleLdsBytes() { return kSingleStatWaves * kNumSymbols * kSingleStatSlots * 4u; }

template <uint32_t FT, bool kNt>
// `elemMap` (nullable): the elements to work on, when they are a subset of the batch -- the single-block elements of a batch
// that also holds larger ones, which the host sends to the kernels of their own size class (capi.hip, EncodeClass).
__global__ __launch_bounds__(64 * kSingleStatWaves) void k_stats_single(BatchView in, NormalizeArgs a, const uint32_t* elemMap, uint32_t numElems) {
  static_assert(FT == 0 || FT == kFloat16 || FT == kBFloat16, "float32 batches keep the workgroup per element (capi.hip, encodeCommon)");
  constexpr uint32_t S = kSingleStatSlots;
  __shared__ __attribute__((aligned(16))) uint32_t sBins[kSingleStatWaves * kNumSymbols * S];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t slot = blockIdx.x * kSingleStatWaves + wave;
  if (slot >= numElems) return;  // wave-uniform; no barriers in this kernel
  const uint32_t b = elemMap ? elemMap[slot] : slot;

  uint32_t* bins = sBins + wave * kNumSymbols * S;
#pragma unroll
  for (uint32_t i = 0; i < kNumSymbols * S / 4u / 64u; ++i) ((uint4*)bins)[i * 64u + lane] = make_uint4(0, 0, 0, 0);
  uint32_t* myBins = histMine<S>(bins, lane);

  const uint32_t n = in.size(b);  // symbols (bytes or float words), <= 4096: the host's guarantee
  const uint8_t* p = in.ptr(b);
  auto addWord = [&](uint32_t x) {  // the symbols of one 32-bit word of input
    if (FT == 0) {
      histAdd4<S>(myBins, x);
    } else {
      constexpr uint32_t kShift = FT == kFloat16 ? 8u : 7u;
      histAdd<S>(myBins, (x >> kShift) & 0xffu);
      histAdd<S>(myBins, (x >> (16u + kShift)) & 0xffu);
    }
  };
  auto addOne = [&](uint32_t i) {  // symbol i on its own (heads, tails, unaligned elements)
    uint32_t c;
    if (FT == 0) c = p[i];
    else c = ((uint32_t)((const uint16_t*)p)[i] >> (FT == kFloat16 ? 8u : 7u)) & 0xffu;
    histAdd<S>(myBins, c);
  };
  constexpr uint32_t kSymBytes = FT == 0 ? 1u : 2u;
  constexpr uint32_t kSymPerVec = 16u / kSymBytes;
  // symbols before the first 16-byte boundary (raw bytes come at any alignment; float words that are not aligned to
  // their own size cannot reach a boundary: the whole element then goes symbol by symbol)
  const uint32_t mis = (uint32_t)((uintptr_t)p & 15u);
  uint32_t head = ((16u - mis) & 15u) / kSymBytes;
  if ((mis % kSymBytes) != 0u || head > n) head = n;
  const uint32_t numVec = (n - head) / kSymPerVec;  // <= 256 (bytes), 512 (16-bit words)
  const uint4* pv = (const uint4*)(p + (size_t)head * kSymBytes);
  constexpr uint32_t kVecPerLane = 4096u / kSymPerVec / 64u;  // 4 / 8: all of them in flight at once
  pairLdsFence();  // bins zeroed (this wave's LDS operations execute in order; see pairLdsFence)
  {
    uint4 x[kVecPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kVecPerLane; ++k) {
      x[k] = make_uint4(0, 0, 0, 0);
      if (k * 64u + lane < numVec) x[k] = streamLoad<kNt>(&pv[k * 64u + lane]);
    }
#pragma unroll
    for (uint32_t k = 0; k < kVecPerLane; ++k) {
      if (k * 64u + lane < numVec) {
        addWord(x[k].x);
        addWord(x[k].y);
        addWord(x[k].z);
        addWord(x[k].w);
      }
    }
  }
  for (uint32_t i = lane; i < head; i += 64u) addOne(i);
  for (uint32_t i = head + numVec * kSymPerVec + lane; i < n; i += 64u) addOne(i);
  pairLdsFence();  // every count is in the bins

  // ---- normalisation (normalizeElement's arithmetic; lane l = symbols 4 l .. 4 l + 3)
  const int P = a.probBits;
  const uint32_t W = 1u << P;
  uint32_t pdf[4] = {0, 0, 0, 0};
  if (n != 0u) {
    uint32_t qSumMine = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
      const uint32_t count = histFold<S>(bins, 4u * lane + j);
      // GpuANSStatistics.cuh:215-218, round-to-nearest divide and multiply as in normalizeElement
      const float ratio = __fdiv_rn(__uint2float_rn(count), __uint2float_rn(n));
      uint32_t q = __float2uint_rz(__fmul_rn(__uint2float_rn(W), ratio));
      q = (count > 0u && q == 0u) ? 1u : q;
      pdf[j] = q;
      qSumMine += q;
    }
    const uint32_t qSum = __builtin_amdgcn_readlane(waveInclusiveScanDpp(qSumMine), 63);
    const int diff = (int)W - (int)qSum;  // :256
    if (diff >= 0) {  // uniform: the closed form of :258-274
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) pdf[j] += (uint32_t)diff / 256u + ((4u * lane + j < ((uint32_t)diff % 256u)) ? 1u : 0u);
    } else {
      normDeficitTrips(pdf, (uint32_t)(-diff), W);
    }
  }

  uint8_t* ans = a.out.ptr(b) + ansOffsetInArchive(a.floatType, n);
  // pdf[4 l .. 4 l + 3] as four u16 (GpuANSEncode.cuh:568-573)
  ((uint2*)(ans + sizeof(AnsHeader)))[lane] = make_uint2(pdf[0] | (pdf[1] << 16), pdf[2] | (pdf[3] << 16));
  if (lane == 0u) normWriteHeader(a, b, n, ans);
}

// ---------------------------------------------------------------------------
// Encoder.  LDS: two 4 KiB tables, two stages, two 512-byte symbol rings.  The two stages belong to different
// ELEMENTS: the non-spilling ones are guarded (encodeRows, kGuard) and have kEncGuardSlackWords of room behind them.
__host__ __device__ constexpr uint32_t encPairStageWords(int P, bool spill, uint32_t ft) {
  return encStageCap(P, spill, ft) + (spill ? 0u : kEncGuardSlackWords);
}
__host__ __device__ constexpr uint32_t encPairLdsBytes(int P, bool spill, uint32_t ft) {
  return 2u * 4096u + 2u * encPairStageWords(P, spill, ft) * 2u + 2u * 512u;
}

// grid = one workgroup (one wavefront) per pair of elements: the hardware dispatches them as slots free up, which
// balances the kernel's serial phases far better than persistent workgroups that own 5 or 6 pairs each (32768 x 4 Ki
// bf16: 133.7 -> 95.7 us, profiles/r05_ab_pair_encoder_hw_dispatch.txt).  Spill slots (kSpill) come from a.spill as a
// POOL handed out through a.spillFlags (SpillPool, kernels_encode.h).  The loop form also serves a smaller grid.
// The host guarantees size(b) <= 4096 for every element (encTileBlocksFor).
template <int P, uint32_t FT, bool kSpill>
__global__ __launch_bounds__(64) void k_ans_encode_pair(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kCap = encPairStageWords(P, kSpill, FT);
  const uint32_t lane = threadIdx.x;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t half = upper ? 1u : 0u;

  uint4* sTable = (uint4*)smem + half * kNumSymbols;
  uint16_t* stage = (uint16_t*)(smem + 8192u) + half * kCap;
  uint8_t* ring = smem + 8192u + 4u * kCap + half * 512u;
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)sTable;
  const uint32_t stageLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)stage;
  SpillPool pool;
  pool.base = a.spill;
  pool.flags = a.spillFlags;
  pool.pairs = a.spillPairs;
  pool.pair = kNoSpillPair;

  // (a.workMap: the elements to pair up, a.numTickets of them, when they are a subset of the batch -- see k_stats_single)
  const uint32_t B = a.workMap ? a.numTickets : a.numInBatch;
  const uint32_t numPairs = (B + 1u) >> 1;
#pragma unroll 1
  for (uint32_t p = blockIdx.x; p < numPairs; p += gridDim.x) {
    const bool haveHi = 2u * p + 1u < B;
    const uint32_t bLo = a.workMap ? a.workMap[2u * p] : 2u * p;
    const uint32_t bHi = a.workMap ? (haveHi ? a.workMap[2u * p + 1u] : bLo) : 2u * p + 1u;
    const uint32_t sLo = a.in.size(bLo);
    const uint32_t sHi = haveHi ? a.in.size(bHi) : 0u;
    if ((sLo | sHi) == 0u) continue;  // uniform: two empty elements (their headers are the normalisation's)
    // a half without symbols (empty element, or none at all) follows its neighbour's element: it reads that
    // element's first word, keeps nothing and writes nothing
    const uint32_t n = upper ? sHi : sLo;
    const bool have = n != 0u;
    const uint32_t b = have ? (upper ? bHi : bLo) : (sLo ? bLo : bHi);
    const uint32_t esize = have ? n : (sLo ? sLo : sHi);

    const uint8_t* in = a.in.ptr(b);
    uint8_t* archive = a.out.ptr(b);
    uint8_t* ans = archive + ansOffsetInArchive(FT, esize);

    // This half's encoder table, from the 512-byte pdf table the normalisation left in the archive header.  A
    // [B][256] x 16-byte table in HBM between the two kernels would be as many bytes as the exponent plane of a 4 Ki
    // element, written once and read once.  Lane hl builds the entries of symbols hl, hl + 32, ... (a row of 32
    // symbols per step: cdf = the earlier rows' total + a 32-lane scan), so that a step's 16-byte entries go to
    // CONSECUTIVE LDS addresses across the lanes -- eight consecutive symbols per lane put every lane of a half on the
    // same four banks, 16 passes per store (the conflict k_ans_decode_pair's LUT build had, docs/HISTORY.md section 4.4).
    {
      const uint16_t* pdfTable = (const uint16_t*)(ans + sizeof(AnsHeader));
      uint32_t pdf[8];
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) pdf[j] = pdfTable[j * 32u + hl];
      uint32_t before = 0;  // probability mass of the earlier rows (uniform per half)
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t incl = halfInclusiveScanDpp(pdf[j]);
        sTable[j * 32u + hl] = encTableEntry(pdf[j], before + incl - pdf[j], P);
        before += upper ? (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) : (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
      }
    }

    if (FT != 0 && have) {
      // GpuFloatHeader (GpuFloatCompress.cuh:325-337) and the zero padding of the non-comp plane(s) up to 16 bytes
      if (hl == 0) {
        FloatHeader h;
        h.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
        h.size = n;
        h.options = FT | (a.useChecksum ? 0x10u : 0u);
        h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
        *(FloatHeader*)archive = h;
      }
      if (FT == kFloat32) {
        uint16_t* nc2 = (uint16_t*)(archive + 16u);
        uint8_t* nc1 = archive + 16u + 2u * (size_t)roundUp(n, 8u);
        if (n + hl < roundUp(n, 8u)) nc2[n + hl] = 0;
        if (n + hl < roundUp(n, 16u)) nc1[n + hl] = 0;
      } else {
        uint8_t* nc = archive + 16u;
        if (n + hl < roundUp(n, 16u)) nc[n + hl] = 0;
      }
    }
    pairLdsFence();  // tables in place

    const bool alignedMe = encVectorLoadsOk<FT>(in, n);  // (vector loads take any word-aligned address)
    const bool fullMe = have && n == kBlockSize && alignedMe;
    const bool bothFull = __ballot(fullMe) == ~0ull;
    const bool bothAligned = __ballot(have && !alignedMe) == 0ull;  // (a half without an element loads and stores nothing)

    ChunkSource<FT> src;
    src.init(in, archive, esize, 0u);

    uint32_t state;
    uint32_t words;
    uint32_t spilled = 0;
    bool overrun = false;  // see encodeRows: only with a caller-supplied histogram that does not cover the data
    if (bothFull) {
      words = encodeRows<P, FT, true, kSpill, !kSpill, kSpill>(src, n, kRowsPerBlock, tableLds, stageLds, ring, hl, upper, nullptr,
                                                        spilled, state, overrun, &pool);
    } else if (bothAligned) {
      // elements of any size below a block on word-aligned inputs: the chunked path bounded by n (encodeRows, kTail)
      const uint32_t nMax = sLo > sHi ? sLo : sHi;
      words = encodeRows<P, FT, true, kSpill, !kSpill, kSpill, true>(src, n, divUp(nMax, 32u), tableLds, stageLds, ring, hl, upper,
                                                                     nullptr, spilled, state, overrun, &pool);
    } else {
      const uint32_t nMax = sLo > sHi ? sLo : sHi;
      words = encodeRows<P, FT, false, kSpill, !kSpill, kSpill>(src, n, divUp(nMax, 32u), tableLds, stageLds, nullptr, hl, upper,
                                                         nullptr, spilled, state, overrun, &pool);
    }
    pairLdsFence();  // stage complete

    if (have) {
      // final lane states (GpuANSEncode.cuh:207, :584-590)
      ((uint32_t*)(ans + ansStatesOffset()))[hl] = state;
      // zero the pad up to the 16-byte boundary (the spilled part is whole vectors)
      const uint32_t padded = roundUp(words, kBlockAlignWords);
      if (words + hl < padded) stage[words + hl] = 0;
      const uint32_t total = spilled + words;
      const uint32_t totalPadded = spilled + padded;
      if (hl == 0) {
        // complete the header (GpuANSEncode.cuh:533-566) and the block descriptors (:595-608): one block at offset 0
        ((AnsHeader*)ans)->totalCompressedWords = overrun ? 0u : totalPadded;
        if (overrun) ((AnsHeader*)ans)->magicAndVersion = 0u;  // failed element: no decoder will follow this archive
        if (a.outSize) a.outSize[b] = overrun ? 0u : ansOffsetInArchive(FT, n) + ansOverhead(1u) + 2u * totalPadded;
        uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(1u));
        blockWords[0] = make_uint2((n << 16) | total, 0u);
        blockWords[1] = make_uint2(0u, 0u);  // alignment pad entry
      }
    }
    pairLdsFence();
    if (have) {
      uint4* dst = (uint4*)(ans + ansOverhead(1u));
      if (kSpill && spilled) {
        // spilled vectors first (written by this wave through agent-scope stores; they must have been performed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4* sp = (const uint4*)(pool.base + ((size_t)pool.pair * 2u + half) * encSpillSlotWords(P));
        const uint32_t sv = spilled / kBlockAlignWords;
        for (uint32_t i = hl; i < sv; i += 32u) streamStore<kNtEncStores>(&dst[i], coherentLoad16(&sp[i]));
        dst += sv;
      }
      const uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
      const uint4* s4 = (const uint4*)stage;
      for (uint32_t i = hl; i < vecs; i += 32u) streamStore<kNtEncStores>(&dst[i], s4[i]);
    }
    if (kSpill && pool.pair != kNoSpillPair) spillRelease(pool);  // wave-uniform
    pairLdsFence();  // the next pair overwrites tables and stages
  }
}

// ---------------------------------------------------------------------------
// Decoder.  LDS: two 2 KiB rings (2 KiB aligned, at offset 0), two compact LUTs, two store buffers.
__host__ __device__ constexpr uint32_t decPairLdsBytes(int P, uint32_t ft) {
  return 2u * kRingBytes + 2u * (4u << P) + 2u * decXposeBytes(P, ft, 1u);
}

// grid = ceil(B / 2) workgroups of one wavefront.  The host guarantees that every output capacity is <= 4096
// symbols, so a valid archive has at most one block.  Validation and the reporting through outSuccess / outSize are
// k_ans_decode's, evaluated per half.
template <int P, uint32_t FT>
__global__ __launch_bounds__(64) void k_ans_decode_pair(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kLutBytes = 4u << P;
  constexpr uint32_t kXpose = decXposeBytes(P, FT, 1u);
  const uint32_t lane = threadIdx.x;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t half = upper ? 1u : 0u;
  // (a.workMap, kDecOrderMap: the elements to pair up, a.numListed of them, when they are a subset of the batch)
  const bool mapped = a.order == kDecOrderMap && a.workMap;
  const uint32_t B = mapped ? a.numListed : a.numInBatch;
  const uint32_t slot = 2u * blockIdx.x + half;
  const uint32_t b = mapped ? a.workMap[slot < B ? slot : B - 1u] : slot;

  uint32_t* sLut = (uint32_t*)(smem + 2u * kRingBytes + half * kLutBytes);
  // cdf / pdf of the 256 symbols: in this half's ring, which is not in use before the LUT is complete
  uint32_t* sCdf = (uint32_t*)(smem + half * kRingBytes);
  uint32_t* sPdf = sCdf + kNumSymbols;
  // 2^P symbol marks in the tail of this half's LUT (read back into registers before the first LUT store)
  uint8_t* sMark = (uint8_t*)sLut + kLutBytes - (1u << P);

  bool live = slot < B;  // this half still has an element to decode
  const uint8_t* archive = nullptr;
  const uint8_t* ans = nullptr;
  uint32_t floatSize = 0, total = 0, nb = 0, totalWords = 0;
  auto fail = [&](uint32_t reportedSize) {
    if (hl == 0) {
      if (a.outSuccess) a.outSuccess[b] = 0;
      if (a.outSize) a.outSize[b] = reportedSize;
    }
    live = false;
  };
  uint64_t inBytes = ~0ull;
  if (live) {
    archive = a.in.ptr(b);
    inBytes = decodeInBytes(a, b);
    if (inBytes < (FT ? sizeof(FloatHeader) : sizeof(AnsHeader))) fail(0u);  // not even a header
  }
  if (FT && live) {
    // the float header locates the ANS archive: check it before following it
    const FloatHeader fh = *(const FloatHeader*)archive;
    const bool fhOk = fh.magicAndVersion == ((kFloatMagic << 16) | kFloatVersion) && (fh.options & 0xfu) == FT &&
        fh.size <= a.out.size(b) &&
        (uint64_t)sizeof(FloatHeader) + floatUncompDataSize(FT, fh.size) + sizeof(AnsHeader) <= inBytes;
    floatSize = fh.size;
    if (!fhOk) fail(fh.size);
  }
  // Everything at a fixed offset from the start of the ANS archive is requested in ONE round trip: the header, the
  // pdf table (8 probabilities per lane) and the descriptor and lane states of the element's block -- a valid archive
  // of a non-empty element has exactly one block here (capacity <= 4096).  A part is requested early only where it is
  // known to lie inside the caller's buffer: the caller said how many bytes there are, or (archives of unknown
  // extent) a checked float header says that an ANS archive follows -- and, for descriptor and states, that the element
  // is not empty.  Raw ANS archives of unknown extent fetch pdf, descriptor and states after the header was checked,
  // still in one round trip.  A load that is not to be made yet reads the first bytes of the header again.  (Before:
  // header -> {descriptor, pdf} -> [LUT build] -> {states, words} = four dependent round trips per element with three
  // wavefronts per SIMD to hide them.)
  uint2 bw = make_uint2(0u, 0u);
  uint4 rawPdf = make_uint4(0u, 0u, 0u, 0u);
  uint32_t state = 0;
  if (live) {
    const uint32_t ansOff = ansOffsetInArchive(FT, floatSize);
    ans = archive + ansOff;
    const bool known = inBytes != ~0ull;
    const bool pdfEarly = known ? (uint64_t)ansOff + ansOverhead(0u) <= inBytes : FT != 0u;
    const bool early = known ? (uint64_t)ansOff + ansOverhead(1u) <= inBytes : (FT != 0u && floatSize != 0u);
    const AnsHeader header = *(const AnsHeader*)ans;
    rawPdf = *(const uint4*)(pdfEarly ? ans + sizeof(AnsHeader) + 16u * hl : ans);  // pdf[8 hl .. 8 hl + 7]
    bw = *(const uint2*)(early ? ans + ansBlockWordsOffset(1u) : ans);
    state = *(const uint32_t*)(early ? ans + ansStatesOffset() + 4u * hl : ans);
    nb = header.numBlocks;
    total = header.totalUncompressedWords;
    totalWords = header.totalCompressedWords;
    bool success = a.out.size(b) >= total;
    success = success && header.magicAndVersion == ((kAnsMagic << 16) | kAnsVersion) &&
        (header.options & 0xfu) == (uint32_t)P;
    if (FT) success = success && floatSize == total;
    success = success && nb == divUp(total, kBlockSize);
    success = success && (uint64_t)ansOffsetInArchive(FT, total) + ansOverhead(nb) + 2ull * totalWords <= inBytes;
    if (!success) fail(total);
    // (a checked header: everything it describes lies inside the archive)
    if (live && !pdfEarly) rawPdf = ((const uint4*)(ans + sizeof(AnsHeader)))[hl];
    if (live && nb && !early) {
      bw = *(const uint2*)(ans + ansBlockWordsOffset(nb));
      state = ((const uint32_t*)(ans + ansStatesOffset()))[hl];
    }
    if (nb == 0u) bw = make_uint2(0u, 0u);
  }
  if (!live) rawPdf = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t pdf[8] = {rawPdf.x & 0xffffu, rawPdf.x >> 16, rawPdf.y & 0xffffu, rawPdf.y >> 16,
                           rawPdf.z & 0xffffu, rawPdf.z >> 16, rawPdf.w & 0xffffu, rawPdf.w >> 16};
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) mine += pdf[j];
  const uint32_t incl = halfInclusiveScanDpp(mine);
  const uint32_t pdfSum = __shfl(incl, (int)(half * 32u + 31u), 64);
  bool decodeMe = false;
  if (live) {
    // (nb <= 1 here: total <= capacity <= 4096)
    bool blocksOk = true;
    if (nb) {
      const uint32_t w = bw.x & 0xffffu;
      blocksOk = (bw.x >> 16) == total && (bw.y & (kBlockAlignWords - 1u)) == 0u &&
          (uint64_t)bw.y + roundUp(w, kBlockAlignWords) <= (uint64_t)totalWords;
    }
    // the probabilities of a non-empty element sum to 2^P (GpuANSStatistics.cuh:256-316)
    const bool pdfOk = nb == 0u || pdfSum == (1u << P);
    if (hl == 0) {
      if (a.outSuccess) a.outSuccess[b] = (blocksOk && pdfOk) ? 1 : 0;
      if (a.outSize) a.outSize[b] = total;
    }
    decodeMe = nb != 0u && blocksOk && pdfOk;
  }
  if (__ballot(decodeMe) == 0ull) return;  // uniform

  // The set-up of the row loop does not depend on the LUT: it comes first, so that the block's compressed words and
  // its first non-compressed bytes are requested (decodePrefetch) BEFORE the LUT is built and land during the build.
  uint32_t n = 0, numWords = 0;
  const uint8_t* gwords = nullptr;
  uint8_t* outPtr = nullptr;
  if (decodeMe) {
    n = bw.x >> 16;
    numWords = bw.x & 0xffffu;
    gwords = ans + ansOverhead(nb) + 2u * (size_t)bw.y;
    outPtr = a.out.ptr(b);
  } else {
    state = 0;
  }
  // a half with nothing to decode follows its neighbour's element (its prefetches must stay inside an archive)
  {
    const int other = (int)(lane ^ 32u);
    const uint64_t oArchive = __shfl((unsigned long long)(uintptr_t)archive, other, 64);
    const uint64_t oOut = __shfl((unsigned long long)(uintptr_t)outPtr, other, 64);
    const uint64_t oWords = __shfl((unsigned long long)(uintptr_t)gwords, other, 64);
    const uint32_t oFloatSize = __shfl(floatSize, other, 64);
    if (!decodeMe) {
      typedef __attribute__((address_space(1))) uint8_t* GlobalBytes;
      archive = (const uint8_t*)(GlobalBytes)(uintptr_t)oArchive;
      outPtr = (uint8_t*)(GlobalBytes)(uintptr_t)oOut;
      gwords = (const uint8_t*)(GlobalBytes)(uintptr_t)oWords;
      floatSize = oFloatSize;
    }
  }

  RowSink<FT> sink;
  sink.init(outPtr, archive, floatSize, 0, hl);
  const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t ringBase = ldsBase + half * kRingBytes;
  const uint32_t xpose = ldsBase + 2u * kRingBytes + 2u * kLutBytes + half * kXpose;
  const uint32_t nLo = __shfl(n, 0, 64);
  const uint32_t nHi = __shfl(n, 32, 64);
  const uint32_t wLo = __shfl(numWords, 0, 64);
  const uint32_t wHi = __shfl(numWords, 32, 64);
  // (the wide stores cover whole 8-row groups of words that exist; they take any word-aligned output element)
  const bool wide = kXpose != 0 && __ballot((((uintptr_t)outPtr) & (decOutWordBytes(FT) - 1u)) != 0) == 0ull;
  const bool fullBoth = nLo == kBlockSize && nHi == kBlockSize;
  // both blocks small enough to be staged whole (every exponent block of N(0,1) bf16 has ~650 words): the cheaper
  // row loop of decodeBlock (kNoRing), fed by the prefetch
  const bool noRing = wLo <= kRingBytes / 2u && wHi <= kRingBytes / 2u;
  DecodePre pre;
  if (fullBoth && noRing) decodePrefetch<FT>(pre, gwords, numWords, sink, hl, wide);

  if (decodeMe) {
    // cdf / pdf scratch and the symbol marks: mark[cdf[s]] = s for every present symbol
    uint32_t c = incl - mine;
    uint32_t cdf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cdf[j] = c;
      c += pdf[j];
    }
    ((uint4*)sCdf)[2u * hl] = make_uint4(cdf[0], cdf[1], cdf[2], cdf[3]);
    ((uint4*)sCdf)[2u * hl + 1u] = make_uint4(cdf[4], cdf[5], cdf[6], cdf[7]);
    ((uint4*)sPdf)[2u * hl] = make_uint4(pdf[0], pdf[1], pdf[2], pdf[3]);
    ((uint4*)sPdf)[2u * hl + 1u] = make_uint4(pdf[4], pdf[5], pdf[6], pdf[7]);
    for (uint32_t i = hl; i < (1u << P) / 4u; i += 32u) ((uint32_t*)sMark)[i] = 0u;
    pairLdsFence();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (pdf[j]) sMark[cdf[j]] = (uint8_t)(8u * hl + (uint32_t)j);
    }
    pairLdsFence();
    // sym(x) = the largest mark at or below x (cdfs ascend with the symbol; slot 0 belongs to the first present
    // symbol, so "no mark" = 0 never surfaces).  128 slots per step: a lane owns FOUR consecutive slots of the step --
    // one dword of marks in, one 16-byte vector of entries out, both at consecutive addresses across the lanes, so
    // neither conflicts in LDS (a lane owning 2^P / 32 CONSECUTIVE slots wrote its entries at a 32-dword stride: 32
    // stores per element with all lanes of a half on one bank -- half of the kernel's per-element cost,
    // profiles/r04_single_block_fixed_cost.txt) -- a max-scan across the half's lanes per step, and the running maximum
    // of the earlier steps carried along.  The marks live in the tail of the LUT they are replaced by: the entries of
    // step t overwrite the marks of steps 4 (t - 3 S / 4) .. + 3 <= t (S steps), all read by then.
    constexpr uint32_t kSteps = (1u << P) / 128u;
    uint32_t carry = 0;  // largest mark of the earlier steps (uniform per half)
#pragma unroll 1
    for (uint32_t t = 0; t < kSteps; ++t) {
      const uint32_t m4 = ((const uint32_t*)sMark)[t * 32u + hl];
      uint32_t run[4];
      run[0] = m4 & 0xffu;
      run[1] = (m4 >> 8) & 0xffu;
      run[2] = (m4 >> 16) & 0xffu;
      run[3] = m4 >> 24;
#pragma unroll
      for (int j = 1; j < 4; ++j) run[j] = run[j] > run[j - 1] ? run[j] : run[j - 1];
      const uint32_t inclMax = halfInclusiveMaxScanDpp(run[3]);
      uint32_t excl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inclMax, 0x138, 0xf, 0xf, true);  // wave_shr:1
      if (hl == 0u) excl = 0u;  // (lane 32 would see lane 31's maximum)
      excl = excl > carry ? excl : carry;
      uint32_t e[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t x = t * 128u + hl * 4u + j;
        const uint32_t sym = run[j] > excl ? run[j] : excl;
        e[j] = (sPdf[sym] & 0xfffu) | (((x - sCdf[sym]) & 0xfffu) << 12) | (sym << 24);
      }
      const uint32_t last = upper ? (uint32_t)__builtin_amdgcn_readlane((int)inclMax, 63) : (uint32_t)__builtin_amdgcn_readlane((int)inclMax, 31);
      carry = carry > last ? carry : last;
      pairLdsFence();  // this step's marks (and the table reads) are in registers: its entries may replace marks
      ((uint4*)sLut)[t * 32u + hl] = make_uint4(e[0], e[1], e[2], e[3]);
    }
  }
  pairLdsFence();  // LUTs complete, scratch (= the rings) free

#define DGPU_PAIR_DECODE_FULL(WIDE, NORING) \
  decodeBlock<P, FT, true, WIDE, false, true, NORING, NORING>(xpose, state, n, kRowsPerBlock / kGroupRows, gwords, numWords, ringBase, sLut, sink, hl, upper, &pre)
  if (fullBoth) {
    if (wide) {
      if (noRing) DGPU_PAIR_DECODE_FULL(true, true); else DGPU_PAIR_DECODE_FULL(true, false);
    } else {
      if (noRing) DGPU_PAIR_DECODE_FULL(false, true); else DGPU_PAIR_DECODE_FULL(false, false);
    }
#undef DGPU_PAIR_DECODE_FULL
  } else {
    // Elements below a block: the whole 8-row groups BOTH halves fill run the straight-line step, the rows above them
    // (each element's partial last row, the rows only the longer element has) the predicated one (decodeBlock, kTail)
    const uint32_t maxN = nLo > nHi ? nLo : nHi, minN = nLo < nHi ? nLo : nHi;
    const uint32_t maxRows = divUp(maxN, 32u);
    const uint32_t groups = (minN / 32u) / kGroupRows;
    const uint32_t topRows = maxRows - groups * kGroupRows;
    if (groups != 0u) {  // (minN == 0: a half without an element -> the predicated path)
#define DGPU_PAIR_DECODE_TAIL(WIDE, NORING) \
  decodeBlock<P, FT, true, WIDE, false, true, NORING, false, true>(xpose, state, n, groups, gwords, numWords, ringBase, sLut, sink, hl, upper, nullptr, topRows)
      if (wide) {
        if (noRing) DGPU_PAIR_DECODE_TAIL(true, true); else DGPU_PAIR_DECODE_TAIL(true, false);
      } else {
        if (noRing) DGPU_PAIR_DECODE_TAIL(false, true); else DGPU_PAIR_DECODE_TAIL(false, false);
      }
#undef DGPU_PAIR_DECODE_TAIL
    } else {
      decodeBlock<P, FT, false, false, false, true>(xpose, state, n, divUp(maxRows, kGroupRows), gwords, numWords, ringBase, sLut, sink, hl, upper);
    }
  }
}

}  // namespace dgpu
