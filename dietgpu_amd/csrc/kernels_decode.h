// rANS decode for gfx950, with the float join fused into the write-out.
//
// Behavioural contract: ansDecodeTable / ansDecodeKernel / decodeOneWarp
// (dietgpu/ans/GpuANSDecode.cuh:34-476) and JoinFloatWriter / FloatOutProvider
// (dietgpu/float/GpuFloatDecompress.cuh:320-521).  Organisation for wave64:
//
//   * one wave64 decodes TWO 4 KiB blocks (lanes 0-31 / 32-63), one 64-bit
//     ballot per row, each half counting its own 32-bit slice from the top
//     lane down (the mirror of the encoder's ascending emission order);
//   * the 2^P-entry decode LUT lives in LDS; decoded symbols go to a 4 KiB LDS
//     stage per block instead of 1-byte global stores;
//   * after the block is decoded the half-wave streams it out with 16-byte
//     vectors: plain copy for raw bytes, or -- float codec -- joins each
//     exponent byte with its non-compressed byte(s) read as 16-byte vectors
//     from the archive and stores whole float words.
#pragma once

#include "format.h"
#include "kernels_stats.h"

namespace dgpu {

// Locates the ANS archive of batch element b (skipping the float header and
// non-comp plane for float archives, FloatANSProvider,
// GpuFloatDecompress.cuh:320-351).
__device__ __forceinline__ const uint8_t* locateAns(const uint8_t* archive, uint32_t ft, uint32_t* floatSize) {
  if (!ft) return archive;
  const FloatHeader fh = *(const FloatHeader*)archive;
  if (floatSize) *floatSize = fh.size;
  return archive + 16u + floatUncompDataSize(ft, fh.size);
}

// ---------------------------------------------------------------------------
// Decode LUT: lut[b][x] = (x - cdf[sym]) << 20 | pdf[sym] << 8 | sym for
// x in [0, 2^P)  (packDecodeLookup, GpuANSDecode.cuh:34-41).  grid = B, 256
// threads; every slot finds its symbol by binary search over the cdf in LDS.
__global__ __launch_bounds__(256) void k_decode_table(
    BatchView in, uint32_t floatType, int probBits, uint32_t* __restrict__ lut) {
  __shared__ uint32_t sCdf[kNumSymbols];
  __shared__ uint32_t sPdf[kNumSymbols];
  __shared__ uint32_t sWave[4];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t b = blockIdx.x;

  const uint8_t* ans = locateAns(in.ptr(b), floatType, nullptr);
  const AnsHeader* h = (const AnsHeader*)ans;
  if (h->magicAndVersion != ((kAnsMagic << 16) | kAnsVersion)) return;
  if (h->totalUncompressedWords == 0) return;  // GpuANSDecode.cuh:424-427

  const uint32_t pdf = ((const uint16_t*)(ans + sizeof(AnsHeader)))[tid];
  uint32_t incl = waveInclusiveScan(pdf, lane);
  if (lane == 63) sWave[wave] = incl;
  __syncthreads();
  uint32_t waveBase = 0;
  for (uint32_t w = 0; w < wave; ++w) waveBase += sWave[w];
  sCdf[tid] = waveBase + incl - pdf;
  sPdf[tid] = pdf;
  __syncthreads();

  const uint32_t slots = 1u << probBits;
  uint32_t* out = lut + (size_t)b * slots;
  for (uint32_t x = tid; x < slots; x += 256u) {
    // last symbol s with cdf[s] <= x (zero-pdf symbols share the cdf of their
    // successor and are skipped by taking the last one)
    uint32_t lo = 0, hi = kNumSymbols;  // invariant: cdf[lo] <= x, answer in [lo, hi)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      uint32_t mid = (lo + hi) >> 1;
      bool le = sCdf[mid] <= x;
      lo = le ? mid : lo;
      hi = le ? hi : mid;
    }
    out[x] = ((x - sCdf[lo]) << 20) | (sPdf[lo] << 8) | lo;
  }
}

// ---------------------------------------------------------------------------
struct DecodeArgs {
  BatchView in;            // archive pointers
  BatchView out;           // output pointers + capacities (bytes for raw, float words for float)
  uint32_t floatType;      // must equal the template FT
  const uint32_t* lut;     // [B][1 << P]
  uint8_t* outSuccess;     // [B] nullable
  uint32_t* outSize;       // [B] nullable
};

// float join, FloatTypeInfo<FT>::join (GpuFloatUtils.cuh:117-119,149-159,187-190)
__device__ __forceinline__ uint32_t joinF16(uint32_t comp, uint32_t nc) { return (comp << 8) | nc; }
__device__ __forceinline__ uint32_t joinBF16(uint32_t comp, uint32_t nc) {
  return (((comp << 8) | nc) >> 1) | ((nc & 1u) << 15);
}
__device__ __forceinline__ uint32_t joinF32(uint32_t comp, uint32_t nc24) {
  uint32_t v = (comp << 24) | nc24;
  return (v >> 1) | (v << 31);
}

template <int P, bool kFull>
__device__ __forceinline__ void decodeRows(
    uint32_t state,
    uint32_t n,
    uint32_t maxRows,
    const uint16_t* __restrict__ words,  // this half's compressed words (global)
    uint32_t numWords,
    const uint32_t* __restrict__ lut,    // LDS
    uint8_t* __restrict__ stage,         // LDS, this half's 4 KiB symbol stage
    uint32_t hl,
    bool upper) {
  constexpr uint32_t kMask = (1u << P) - 1u;
  const uint32_t laneMaskGe = ~((1u << hl) - 1u);
  uint32_t pos = numWords;

  auto step = [&](uint32_t row, bool valid) {
    const uint32_t e = lut[state & kMask];
    if (valid) {
      stage[row * 32u + hl] = (uint8_t)(e & 0xffu);
      state = __umul24((e >> 8) & 0xfffu, state >> P) + (e >> 20);
    }
    const bool read = valid && (state < kMinState);
    const uint64_t vote = __ballot(read);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    if (read) {
      const uint32_t v = words[pos - __popc(vh & laneMaskGe)];
      state = (state << kEncodedBits) | v;
    }
    pos -= __popc(vh);
  };

  if (kFull) {
#pragma unroll 8
    for (int row = (int)kRowsPerBlock - 1; row >= 0; --row) step((uint32_t)row, true);
  } else {
#pragma unroll 1
    for (int row = (int)maxRows - 1; row >= 0; --row) {
      step((uint32_t)row, (uint32_t)row * 32u + hl < n);
    }
  }
}

// Streams one decoded block out of its LDS stage.  FT == 0: raw bytes.
template <uint32_t FT>
__device__ __forceinline__ void writeBlock(
    const uint8_t* __restrict__ stage,  // LDS, n symbols
    uint32_t n,
    uint32_t block,
    uint8_t* __restrict__ outBase,      // element output base
    const uint8_t* __restrict__ archive,// float archive base (FT != 0)
    uint32_t floatSize,
    uint32_t hl) {
  const size_t first = (size_t)block * kBlockSize;  // first symbol of the block
  if (FT == 0) {
    uint8_t* dst = outBase + first;
    if (n == kBlockSize && (((uintptr_t)dst & 15u) == 0)) {
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) ((uint4*)dst)[hl + 32u * k] = ((const uint4*)stage)[hl + 32u * k];
    } else {
      for (uint32_t i = hl; i < n; i += 32u) dst[i] = stage[i];
    }
  } else if (FT == kFloat16 || FT == kBFloat16) {
    const uint8_t* nc = archive + 16u + first;
    uint16_t* dst = (uint16_t*)outBase + first;
    if (n == kBlockSize && (((uintptr_t)dst & 15u) == 0)) {
      // 16 elements per lane per step: 16 B comp (LDS) + 16 B non-comp -> 2 x 16 B out
#pragma unroll 2
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t v = hl + 32u * k;
        const uint4 c = ((const uint4*)stage)[v];
        const uint4 r = ((const uint4*)nc)[v];
        const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
        const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t w[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t cb = (cw[j] >> (8 * q)) & 0xffu;
            const uint32_t rb = (rw[j] >> (8 * q)) & 0xffu;
            w[q] = FT == kFloat16 ? joinF16(cb, rb) : joinBF16(cb, rb);
          }
          o[2 * j] = w[0] | (w[1] << 16);
          o[2 * j + 1] = w[2] | (w[3] << 16);
        }
        ((uint4*)dst)[2 * v] = make_uint4(o[0], o[1], o[2], o[3]);
        ((uint4*)dst)[2 * v + 1] = make_uint4(o[4], o[5], o[6], o[7]);
      }
    } else {
      for (uint32_t i = hl; i < n; i += 32u) {
        const uint32_t cb = stage[i], rb = nc[i];
        dst[i] = (uint16_t)(FT == kFloat16 ? joinF16(cb, rb) : joinBF16(cb, rb));
      }
    }
  } else {  // kFloat32: u16 plane of roundUp(size, 8) entries, then the high-byte plane
    const uint16_t* nc2 = (const uint16_t*)(archive + 16u) + first;
    const uint8_t* nc1 = archive + 16u + 2u * (size_t)roundUp(floatSize, 8u) + first;
    uint32_t* dst = (uint32_t*)outBase + first;
    if (n == kBlockSize && (((uintptr_t)dst & 15u) == 0)) {
      // 4 elements per lane per step
#pragma unroll 4
      for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t v = hl + 32u * k;
        const uint32_t c = ((const uint32_t*)stage)[v];
        const uint2 lo = ((const uint2*)nc2)[v];
        const uint32_t hi = ((const uint32_t*)nc1)[v];
        uint4 o;
        o.x = joinF32(c & 0xffu, ((hi & 0xffu) << 16) | (lo.x & 0xffffu));
        o.y = joinF32((c >> 8) & 0xffu, (((hi >> 8) & 0xffu) << 16) | (lo.x >> 16));
        o.z = joinF32((c >> 16) & 0xffu, (((hi >> 16) & 0xffu) << 16) | (lo.y & 0xffffu));
        o.w = joinF32(c >> 24, ((hi >> 24) << 16) | (lo.y >> 16));
        ((uint4*)dst)[v] = o;
      }
    } else {
      for (uint32_t i = hl; i < n; i += 32u) {
        dst[i] = joinF32(stage[i], ((uint32_t)nc1[i] << 16) | nc2[i]);
      }
    }
  }
}

// grid = (maxTiles, B), 256 threads, LDS = LUT + 8 x 4 KiB stages.
template <int P, uint32_t FT>
__global__ __launch_bounds__(256) void k_ans_decode(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* sLut = (uint32_t*)smem;
  uint8_t* sStage = smem + (4u << P);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);
  const uint32_t b = blockIdx.y;
  const uint32_t tile = blockIdx.x;

  const uint8_t* archive = a.in.ptr(b);
  uint32_t floatSize = 0;
  const uint8_t* ans = locateAns(archive, FT, &floatSize);
  const AnsHeader header = *(const AnsHeader*)ans;
  const uint32_t nb = header.numBlocks;
  const uint32_t total = header.totalUncompressedWords;

  // Is there room for the output?  Capacity and size are in the API's units
  // (bytes for raw ANS, float words for the float codec); GpuANSDecode.cuh:325-341
  bool success = a.out.size(b) >= total;
  success = success && header.magicAndVersion == ((kAnsMagic << 16) | kAnsVersion) &&
      (header.options & 0xfu) == (uint32_t)P;
  if (FT) success = success && floatSize == total;
  if (tile == 0 && tid == 0) {
    if (a.outSuccess) a.outSuccess[b] = success ? 1 : 0;
    if (a.outSize) a.outSize[b] = total;
  }
  if (!success || tile * kBlocksPerTile >= nb) return;

  {
    const uint4* src = (const uint4*)(a.lut + ((size_t)b << P));
    uint4* dst = (uint4*)sLut;
    for (uint32_t i = tid; i < (1u << P) / 4u; i += 256u) dst[i] = src[i];
  }
  __syncthreads();

  const uint32_t block = tile * kBlocksPerTile + hw;
  const bool haveBlock = block < nb;

  uint32_t state = 0, n = 0, numWords = 0, start = 0;
  if (haveBlock) {
    state = ((const uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl];
    const uint2 bw = ((const uint2*)(ans + ansBlockWordsOffset(nb)))[block];
    n = bw.x >> 16;
    numWords = bw.x & 0xffffu;
    start = bw.y;
  }
  const uint16_t* words = (const uint16_t*)(ans + ansOverhead(nb)) + start;
  uint8_t* stage = sStage + hw * kBlockSize;

  // uniform per wave: both halves hold full blocks?
  const uint32_t nFirst = __shfl(n, 0, 64);
  const uint32_t nSecond = __shfl(n, 32, 64);
  if (nFirst == kBlockSize && nSecond == kBlockSize) {
    decodeRows<P, true>(state, n, kRowsPerBlock, words, numWords, sLut, stage, hl, upper);
  } else {
    const uint32_t maxN = nFirst > nSecond ? nFirst : nSecond;
    decodeRows<P, false>(state, n, divUp(maxN, 32u), words, numWords, sLut, stage, hl, upper);
  }

  if (haveBlock) {
    writeBlock<FT>(stage, n, block, a.out.ptr(b), archive, floatSize, hl);
  }
}

}  // namespace dgpu
