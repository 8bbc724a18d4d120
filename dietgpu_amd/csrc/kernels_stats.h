// Symbol statistics for the rANS coder: batched byte histogram, probability
// normalisation to 2^probBits, checksum and header-info kernels.
//
// Behavioural contract = dietgpu/ans/GpuANSStatistics.cuh:21-430,
// dietgpu/ans/GpuChecksum.cuh:26-133, dietgpu/ans/GpuANSInfo.cuh:17-37,
// dietgpu/float/GpuFloatInfo.cuh:18-41.  The code is organised for wave64:
// 256-thread workgroups = 4 wavefronts, DPP/shuffle reductions, LDS bins.
#pragma once

#include "format.h"

namespace dgpu {

__device__ __forceinline__ uint32_t waveReduceSum(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ uint32_t waveReduceXor(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v ^= __shfl_xor(v, m, 64);
  return v;
}
// inclusive scan across the 64 lanes of a wavefront
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

// inclusive scan across the wavefront with DPP row shifts / broadcasts (6 VALU instructions, no LDS
// crossbar round trips as with ds_bpermute shuffles)
__device__ __forceinline__ uint32_t waveInclusiveScanDpp(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
  return v;
}

// ---------------------------------------------------------------------------
// Probability normalisation (GpuANSStatistics.cuh:178-367), one 256-thread
// workgroup per batch element.  Produces
//   encTable[b][sym] = {thresh = pdf << (31-P), magic, cdf, (2^P - pdf) | shift << 24}
//     -- the encoder's packed form: state' = x + cdf + div * (2^P - pdf) with
//        div = (mulhi(x, magic) + x) >> shift; the low 24 bits feed
//        v_mad_u32_u24 directly;
//   refTable[b][sym] = {pdf, cdf, magic, shift} (reference layout, optional);
// and, when `out` is given, the static part of the ANS archive header plus the
// u16 pdf table (the fields ansEncodeCoalesce writes at
// GpuANSEncode.cuh:553-573); totalCompressedWords and the reported size are
// completed by the encode kernel's last tile (or here for an empty input).
//
// The reference sorts (q << 16 | sym) with a CUB block radix sort and then walks the sorted order.  Neither
// branch needs the sort here: the surplus branch adds by SYMBOL index (closed form); the deficit branch
// subtracts 1 from "the iter smallest keys among the entries with q > 1", trip after trip, which one
// wavefront (4 symbols per lane, in symbol order) selects with ballots: a trip that covers every q > 1 entry
// is applied t times at once, the final partial trip finds the threshold value v* by bisection over
// c(v) = #{1 < q <= v} and takes the first entries with q == v* in symbol order (= ascending key).  ~300
// wave-instructions instead of the 2300 of a 256 x 256 rank-by-counting (round 1), which made batches of
// many small elements normalisation-bound.  The block scan is wave64 DPP.

// minimum over the 64 lanes of a wavefront (DPP row shifts / broadcasts; every lane receives it)
__device__ __forceinline__ uint32_t waveMinDpp(uint32_t v) {
  auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
  const int kId = (int)0xffffffffu;  // lanes without a source keep the identity
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
  v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(kId, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The deficit branch of the normalisation (GpuANSStatistics.cuh:275-315) on ONE wavefront: lane l holds the quantised
// probabilities of symbols 4 l .. 4 l + 3 in qq[]; subtracts 1 from the smallest entries that are still > 1, trip
// after trip, until `d` has been taken off the sum (see the header comment above for the selection by ballots).
__device__ __forceinline__ void normDeficitTrips(uint32_t (&qq)[4], uint32_t d, const uint32_t W) {
  // c(v) = number of entries with 1 < q <= v  (wave-uniform)
  auto countUpTo = [&](uint32_t v) -> uint32_t {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (uint32_t)__popcll(__ballot((qq[j] - 2u) <= (v - 2u)));
    return c;
  };
  // smallest v >= 2 with c(v) >= target (1 <= target <= number of q > 1 entries)
  auto threshold = [&](uint32_t target) -> uint32_t {
    uint32_t lo = 2u, hi = W;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (countUpTo(mid) >= target) hi = mid;
      else lo = mid + 1u;
    }
    return lo;
  };
  while (d > 0u) {
    const uint32_t n = countUpTo(W);  // entries with q > 1 (q <= W always)
    if (n == 0u) break;               // cannot happen: the sum would be <= 256 <= W
    if (d >= n) {
      // full trips: every q > 1 entry loses 1 per trip until the smallest of them reaches 1
      // (threshold(1) is the smallest q > 1: one wave minimum instead of a bisection)
      uint32_t least = 0xffffffffu;
#pragma unroll
      for (int j = 0; j < 4; ++j) least = (qq[j] > 1u && qq[j] < least) ? qq[j] : least;
      const uint32_t m = waveMinDpp(least) - 1u;
      const uint32_t t = (d / n) < m ? (d / n) : m;
#pragma unroll
      for (int j = 0; j < 4; ++j) qq[j] -= (qq[j] > 1u) ? t : 0u;
      d -= t * n;
    } else {
      // last, partial trip: the d smallest keys (q, then symbol) among the q > 1 entries
      const uint32_t vs = threshold(d);
      const uint32_t below = vs > 2u ? countUpTo(vs - 1u) : 0u;
      const uint32_t need = d - below;  // of the entries with q == vs, the first `need` in symbol order
      uint32_t run = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t m64 = __ballot(qq[j] == vs);
        run += __builtin_amdgcn_mbcnt_hi((uint32_t)(m64 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m64, 0u));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool eq = qq[j] == vs;
        const bool dec = (qq[j] > 1u && qq[j] < vs) || (eq && run < need);
        run += eq ? 1u : 0u;
        qq[j] -= dec ? 1u : 0u;
      }
      d = 0u;
    }
  }
}

// Encoder table entry of one symbol.  The encoder needs q = floor(x / pdf) for states x < 2^31
// (x >= 16 always): q = umulhi(x, m) >> sh with a 32-bit m, no add-back step
// (the 33-bit "round-up" magic of the reference, GpuANSStatistics.cuh:349-358, is only needed
// for full 32-bit dividends):
//   pdf = 2^k, k >= 1: m = 2^(32-k), sh = 0                 (exact shift)
//   otherwise, L = ceil(log2 pdf): m = floor(2^(31+L) / pdf) + 1, sh = L - 1
//     (m < 2^32 because pdf > 2^(L-1); error term x * e / (pdf * 2^(31+L)) with
//      e <= pdf < 2^L and x < 2^31 stays below 1 / pdf: the floor is exact)
//   pdf = 1: m = 2^32 - 1 gives q = x - 1; the missing (2^P - 1) is added to
//     the entry's cdf term instead
// state' = x + cdf + q * (2^P - pdf).  The double division is exact enough:
// the quotient is at least 2^-11 away from an integer, its error is < 2^-20.
__device__ __forceinline__ uint4 encTableEntry(uint32_t pdf, uint32_t cdf, int P) {
  const uint32_t W = 1u << P;
  uint32_t m = 0, sh = 0, cdfTerm = cdf;
  if (pdf == 1u) {
    m = 0xffffffffu;
    cdfTerm = cdf + (W - 1u);
  } else if (pdf > 1u) {
    const uint32_t L = 32u - (uint32_t)__clz((int)(pdf - 1u));  // ceil(log2 pdf)
    if ((pdf & (pdf - 1u)) == 0u) {
      m = 1u << (32u - L);
    } else {
      sh = L - 1u;
      m = (uint32_t)(unsigned long long)floor(ldexp(1.0, 31 + (int)L) / (double)pdf) + 1u;
    }
  }
  uint4 e;
  e.x = pdf << (kStateBits - P);
  e.y = m;
  e.z = cdfTerm;
  e.w = ((W - pdf) & 0xffffffu) | (sh << 24);
  return e;
}

struct NormalizeArgs {
  BatchView sizes;           // only size(b) is used
  const uint32_t* hist;      // [B][histParts][256]: per-workgroup partial histograms, summed here
  uint32_t* histAcc;         // nullable: [B][256] counts accumulated by atomics (library-owned, zero at rest);
                             // used instead of `hist`, read and put back to zero here
  uint32_t histParts;
  int probBits;
  uint4* encTable;           // [B][256] nullable
  uint4* refTable;           // [B][256] nullable
  BatchView out;             // archive base pointers (ptr(b)); valid iff writeHeader
  uint32_t writeHeader;
  uint32_t floatType;        // != 0: ANS archive is embedded in a float archive
  uint32_t useChecksum;
  const uint32_t* checksum;  // [B] nullable
  uint32_t* outSize;         // [B] nullable
  uint32_t floatUseChecksum; // float archives: checksum flag for the header of an EMPTY element
  // encode hand-off state cleared here for the encode kernel that follows on the stream
  uint64_t* tileDesc;        // [B][maxTiles] nullable
  uint32_t maxTiles;
  uint32_t* claims;          // [maxTiles][numInBatch] nullable (encoder's tile claim words)
  uint32_t numInBatch;
  // batches whose elements differ widely in size (EncodeArgs::workMap): descriptors and claim words exist for the tiles
  // that exist only, element by element; element b's begin at tileBase[b], it has ceil(size / tileSymbols) of them
  const uint32_t* tileBase;  // nullable: [numInBatch]
  uint32_t tileSymbols;
  // second level of the encoder's look-back (EncodeArgs::groupWords), cleared here too; null / 0: none
  uint64_t* groupWords;      // [numInBatch][groupWordsPerElement]
  uint32_t groupWordsPerElement;
};

// The static part of the ANS archive header of element b (the fields ansEncodeCoalesce writes at
// GpuANSEncode.cuh:553-566; totalCompressedWords and the reported size are completed by the encode kernel) and, for
// an EMPTY element -- no encode tile will run for it -- its reported size and float header.  One lane.
__device__ __forceinline__ void normWriteHeader(const NormalizeArgs& a, uint32_t b, uint32_t total, uint8_t* ans) {
  const uint32_t nb = divUp(total, kBlockSize);
  AnsHeader h;
  h.magicAndVersion = (kAnsMagic << 16) | kAnsVersion;
  h.numBlocks = nb;
  h.totalUncompressedWords = total;
  h.totalCompressedWords = 0;  // completed by k_ans_encode's last tile
  h.options = (uint32_t)a.probBits | (a.useChecksum ? 0x10u : 0u);
  h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
  h.unused0 = 0;
  h.unused1 = 0;
  *(AnsHeader*)ans = h;
  if (nb == 0) {
    if (a.outSize) a.outSize[b] = ansOffsetInArchive(a.floatType, total) + ansOverhead(0);
    if (a.floatType) {
      FloatHeader fh;
      fh.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
      fh.size = 0;
      fh.options = a.floatType | (a.floatUseChecksum ? 0x10u : 0u);
      fh.checksum = 0;
      *(FloatHeader*)a.out.ptr(b) = fh;
    }
  }
}

// One 256-thread workgroup normalises batch element b.  kCoherent: the partial
// histograms were written by other workgroups of the SAME kernel (write-through
// agent-scope stores) and are read with agent-scope loads.
// `scratch`: kNormScratchWords u32 of LDS, 16-byte aligned, free for the duration of the call.
constexpr uint32_t kNormScratchWords = 3u * kNumSymbols + 4u;
// `direct` (nullable): this thread's count of symbol tid, when the calling workgroup has counted the whole
// element itself (one histogram workgroup per element: no partial histograms, no arrival counter).
template <bool kCoherent>
// `partials` / `numPartials` (nullable): where the element's partial histograms lie when they are not at
// a.hist[b][a.histParts] (HistFuse::workMap).
__device__ __forceinline__ void normalizeElement(const NormalizeArgs& a, const uint32_t b, uint32_t* scratch,
                                                 const uint32_t* direct = nullptr, const uint32_t* partials = nullptr,
                                                 uint32_t numPartials = 0) {
  uint32_t* sKeys = scratch;  // q per symbol
  uint32_t* sPdf = scratch + 2u * kNumSymbols;
  uint32_t* sWave = scratch + 3u * kNumSymbols;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t total = a.sizes.size(b);
  const int P = a.probBits;
  const uint32_t W = 1u << P;

  uint8_t* ans = nullptr;
  if (a.writeHeader) ans = a.out.ptr(b) + ansOffsetInArchive(a.floatType, total);

  uint32_t pdf = 0, cdf = 0;

  if (a.tileDesc && a.tileBase) {
    const uint32_t first = a.tileBase[b], count = divUp(total, a.tileSymbols);
    for (uint32_t i = tid; i < count; i += 256u) {
      a.tileDesc[(size_t)first + i] = 0;
      if (a.claims) a.claims[(size_t)first + i] = 0;
    }
  } else if (a.tileDesc) {
    for (uint32_t i = tid; i < a.maxTiles; i += 256u) {
      a.tileDesc[(size_t)b * a.maxTiles + i] = 0;
      if (a.claims) a.claims[(size_t)i * a.numInBatch + b] = 0;
    }
    for (uint32_t i = tid; i < a.groupWordsPerElement; i += 256u) a.groupWords[(size_t)b * a.groupWordsPerElement + i] = 0;
  }

  if (total != 0) {
    uint32_t count = 0;
    if (direct) {
      count = *direct;
    } else if (a.histAcc) {
      // counts accumulated with atomics by the histogram workgroups of this element
      uint32_t* acc = a.histAcc + (size_t)b * kNumSymbols + tid;
      count = __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(acc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero at rest
    } else {
      // sum of the per-workgroup partial histograms, up to 16 loads in flight
      const uint32_t* hp = (partials ? partials : a.hist + (size_t)b * a.histParts * kNumSymbols) + tid;
      const uint32_t histParts = partials ? numPartials : a.histParts;
      auto part = [&](uint32_t x) -> uint32_t {
        const uint32_t* q = hp + (size_t)x * kNumSymbols;
        return kCoherent ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
      };
      uint32_t x = 0;
      for (; x + 16u <= histParts; x += 16u) {
        uint32_t c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = part(x + k);
#pragma unroll
        for (int k = 0; k < 16; ++k) count += c[k];
      }
      for (; x + 4u <= histParts; x += 4u) {
        uint32_t c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = part(x + k);
        count += (c[0] + c[1]) + (c[2] + c[3]);
      }
      for (; x < histParts; ++x) count += part(x);
    }
    // :215  qProb = kProbWeight * ((float)count / (float)totalNum), truncated.
    // Explicit round-to-nearest divide and multiply: no fma contraction, no
    // approximate reciprocal.
    float ratio = __fdiv_rn(__uint2float_rn(count), __uint2float_rn(total));
    uint32_t q = __float2uint_rz(__fmul_rn(__uint2float_rn(W), ratio));
    q = (count > 0 && q == 0) ? 1u : q;  // :218

    const uint32_t qScan = waveInclusiveScanDpp(q);
    if (lane == 63) sWave[wave] = qScan;
    sKeys[tid] = q;
    __syncthreads();
    const int qSum = (int)(sWave[0] + sWave[1] + sWave[2] + sWave[3]);

    int diff = (int)W - qSum;  // :256
    if (diff >= 0) {  // uniform
      // :258-274.  Each loop trip of the reference adds 1 to every entry whose
      // SYMBOL index is < min(diff, 256): the result does not depend on the sorted
      // order at all.  Closed form of the loop:
      pdf = q + (uint32_t)diff / 256u + ((tid < ((uint32_t)diff % 256u)) ? 1u : 0u);
    } else {
      // :275-315  subtract 1 from the smallest entries that are still > 1, until the sum fits
      if (wave == 0) {
        const uint4 v4 = ((const uint4*)sKeys)[lane];  // symbols 4 lane .. 4 lane + 3
        uint32_t qq[4] = {v4.x, v4.y, v4.z, v4.w};
        normDeficitTrips(qq, (uint32_t)(-diff), W);
        ((uint4*)sPdf)[lane] = make_uint4(qq[0], qq[1], qq[2], qq[3]);
      }
      __syncthreads();
      pdf = sPdf[tid];
    }

    // exclusive scan -> cdf  (:336-341)
    const uint32_t incl = waveInclusiveScanDpp(pdf);
    __syncthreads();  // sWave reuse
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0;
    for (uint32_t w = 0; w < wave; ++w) waveBase += sWave[w];
    cdf = waveBase + incl - pdf;

  }

  if (a.encTable) a.encTable[b * kNumSymbols + tid] = encTableEntry(pdf, cdf, P);
  if (a.refTable) {
    // the reference's table (pdf, cdf, 33-bit magic, shift), :349-358
    uint32_t magic = 0, shift = 0;
    if (pdf > 0) {  // undefined upstream for pdf == 0, never looked up
      shift = 32u - (uint32_t)__clz((int)(pdf - 1u));  // __clz(0) == 32
      const uint64_t one = 1;
      uint64_t magic64 = ((one << 32) * ((one << shift) - (uint64_t)pdf)) / (uint64_t)pdf + 1;
      magic = (uint32_t)magic64;
    }
    a.refTable[b * kNumSymbols + tid] = make_uint4(pdf, cdf, magic, shift);
  }

  if (ans) {
    ((uint16_t*)(ans + sizeof(AnsHeader)))[tid] = (uint16_t)pdf;
    if (tid == 0) normWriteHeader(a, b, total, ans);
  }
}

// Stand-alone normalisation (caller-supplied histograms, dgpu_ans_calc_weights).  grid = B.
__global__ __launch_bounds__(256) void k_normalize(NormalizeArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t sScratch[kNormScratchWords];
  normalizeElement<false>(a, blockIdx.x, sScratch);
}

// ---------------------------------------------------------------------------
// Histogram -> normalisation hand-off inside one kernel.  Every histogram
// workgroup of element b publishes its partial counts (write-through stores),
// waits for them to be performed, and bumps arrive[b]; the workgroup that
// observes the last arrival normalises the element on the spot, so the table is
// being built for early elements while late ones are still being counted, and
// there is no separate kernel (launch gap + ramp + 256 mostly idle CUs) on the
// critical path.  arrive[] is library-owned, zero at rest: the normalising
// workgroup resets it.  arrive == nullptr: plain histogram.
struct HistFuse {
  uint32_t* arrive;  // [B], zero at launch
  uint32_t* acc;     // nullable: [B][256] accumulate-by-atomics mode (few, large elements), zero at launch
  NormalizeArgs norm;
  // Batches whose elements differ widely in size (the host knows the sizes): a 1-D grid with one workgroup per
  // (element, part) listed in workMap -- element-major, entry = element << 16 | part -- and every element cut into
  // parts of partBytes, instead of a (parts, B) rectangle laid out for the largest element.  nullptr: the rectangle.
  const uint32_t* workMap;
  uint32_t partBytes;
};

// Which (element, part of how many) a histogram workgroup counts, and where its partial histogram goes.
struct HistWork {
  uint32_t b, part, parts, slot;
};
__device__ __forceinline__ HistWork histWorkOf(const HistFuse& f, const BatchView& in, uint32_t wordBytes) {
  HistWork w;
  if (f.workMap) {
    const uint32_t m = f.workMap[blockIdx.x];
    w.b = m >> 16;
    w.part = m & 0xffffu;
    const uint64_t bytes = (uint64_t)in.size(w.b) * wordBytes;
    const uint32_t parts = (uint32_t)((bytes + f.partBytes - 1u) / f.partBytes);
    w.parts = parts ? parts : 1u;
    w.slot = blockIdx.x;  // (the parts of an element are consecutive workgroups)
  } else {
    w.b = blockIdx.y;
    w.part = blockIdx.x;
    w.parts = gridDim.x;
    w.slot = blockIdx.y * gridDim.x + blockIdx.x;
  }
  return w;
}

// ---------------------------------------------------------------------------
// Histogram bins in LDS.  Entropy-coder inputs are skewed (one exponent value
// can be a third of the data) and ds_add_u32 serialises lanes of one
// instruction that hit the same address or bank, which made a per-wave
// privatised layout run at ~2 lane-updates per clock per CU.  Layout used
// instead: bin-major with 32 lane slots per bin,
//     word(bin, lane) = bin * 32 + (lane & 31),  bank = lane & 31
// so within a 32-lane LDS pass every lane has a bank of its own whatever the
// data: the pass is conflict-free by construction (16 slots allow 2-way
// conflicts: raw bytes 68 -> 58 us, exponents 50.5 -> 49 us).  The four
// wavefronts of a workgroup share the same 32 KiB (ds_add is atomic; different
// waves are different instructions and merely interleave).
// Workgroups that see little data (many small elements, <= 64 KiB per workgroup) use 16 slots instead: zeroing and
// folding 32 KiB of bins costs more than the conflicts of 16 (measured on MI355X, k_float_histogram,
// profiles/r05_ab_histogram_small_slots.txt: 16384 x 8 Ki bf16 62.4 us with 8 slots, 59.0 with 16, 84.5 with 32; sparse
// fp16 at probBits 11 -- half the symbols are one value -- 72.0 / 58.7 / 87.2; 4 slots 75.5).
constexpr uint32_t kHistSlotsLarge = 32;
constexpr uint32_t kHistSlotsSmall = 16;

template <uint32_t S>
__device__ __forceinline__ void histZero(uint32_t* bins, uint32_t tid) {
  for (uint32_t i = tid; i < kNumSymbols * S / 4u; i += 256u) ((uint4*)bins)[i] = make_uint4(0, 0, 0, 0);
}
// this lane's slot column; bin c lives at mine[c * S]
template <uint32_t S>
__device__ __forceinline__ uint32_t* histMine(uint32_t* bins, uint32_t tid) {
  return bins + (tid & (S - 1u));
}
template <uint32_t S>
__device__ __forceinline__ void histAdd(uint32_t* mine, uint32_t c) { atomicAdd(&mine[c * S], 1u); }
// total of bin `tid` over all slots
template <uint32_t S>
__device__ __forceinline__ uint32_t histFold(const uint32_t* bins, uint32_t tid) {
  const uint4* p = (const uint4*)(bins + tid * S);
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < S / 4u; ++k) {
    const uint4 v = p[k];
    sum += v.x + v.y + v.z + v.w;
  }
  return sum;
}

// Histogram kernel: 16-byte loads with a byte-wise head/tail so any start
// alignment works (the reference test uses stride size+11,
// ANSStatisticsTest.cu:52-57).  grid = (xBlocks, B), 256 threads.
template <uint32_t S>
__device__ __forceinline__ void histAdd4(uint32_t* mine, uint32_t x) {
  histAdd<S>(mine, x & 0xff);
  histAdd<S>(mine, (x >> 8) & 0xff);
  histAdd<S>(mine, (x >> 16) & 0xff);
  histAdd<S>(mine, x >> 24);
}

// `partial` != 0: this workgroup stores its 256 totals to hist[(b * gridDim.x + blockIdx.x) * 256 + bin]
// (no atomics, no zero-initialisation needed; the normalisation adds the parts up).  Otherwise it adds
// them atomically into hist[b][256], which must have been zeroed.  With f.arrive set (partial mode
// only) the last workgroup of the element to get here also normalises it.
__device__ __forceinline__ void histStore(uint32_t* __restrict__ hist, uint32_t partial, const HistFuse& f, const HistWork& w, uint32_t tid, uint32_t sum) {
  const uint32_t b = w.b;
  if (!partial) {
    if (sum) atomicAdd(&hist[b * kNumSymbols + tid], sum);
    return;
  }
  uint32_t* slot = &hist[(size_t)w.slot * kNumSymbols + tid];
  if (!f.arrive) {
    *slot = sum;
    return;
  }
  __shared__ uint32_t sLast;
  __shared__ __attribute__((aligned(16))) uint32_t sScratch[kNormScratchWords];
  if (w.parts == 1u) {
    // the only histogram workgroup of its element (batches of small elements): it holds the complete counts in
    // registers and normalises right away -- no partial histogram through memory, no arrival counter
    normalizeElement<true>(f.norm, b, sScratch, &sum);
    return;
  }
  if (f.acc) {
    // few large elements: thousands of workgroups per element; their counts meet in
    // 256 atomic counters instead of thousands of partial histograms
    if (sum) __hip_atomic_fetch_add(f.acc + (size_t)b * kNumSymbols + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_store(slot, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // ... and performed
  __syncthreads();
  if (tid == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(&f.arrive[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sLast = (prev + 1u == w.parts) ? 1u : 0u;
  }
  __syncthreads();
  if (sLast) {  // uniform
    if (tid == 0) __hip_atomic_store(&f.arrive[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.workMap) normalizeElement<true>(f.norm, b, sScratch, nullptr, hist + (size_t)(w.slot - w.part) * kNumSymbols, w.parts);
    else normalizeElement<true>(f.norm, b, sScratch);
  }
}

template <uint32_t S, bool kNt = true>
__global__ __launch_bounds__(256) void k_histogram(BatchView in, uint32_t* __restrict__ hist, uint32_t partial, HistFuse fuse) {
  __shared__ uint32_t bins[kNumSymbols * S];
  const uint32_t tid = threadIdx.x;
  const HistWork w = histWorkOf(fuse, in, 1u);
  const uint32_t b = w.b;
  histZero<S>(bins, tid);
  __syncthreads();

  uint32_t* myBins = histMine<S>(bins, tid);
  const uint8_t* p = in.ptr(b);
  const uint32_t size = in.size(b);

  // bytes before the first 16-byte boundary
  uint32_t head = (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u);
  head = head < size ? head : size;
  const uint32_t remaining = size - head;
  const uint32_t numVec = remaining / 16u;
  const uint4* pv = (const uint4*)(p + head);

  if (w.part == 0 && tid < head) histAdd<S>(myBins, p[tid]);

  // Every workgroup streams ONE contiguous part of the element (whole 16 KiB steps; the last part takes the rest), four
  // 16-byte loads in flight per lane (the kernel is HBM-latency bound otherwise); see k_float_histogram.
  const uint32_t perPart = roundUp(divUp(numVec, w.parts), 1024u);
  const uint32_t vBegin = w.part * perPart < numVec ? w.part * perPart : numVec;
  const uint32_t vEnd = vBegin + perPart < numVec ? vBegin + perPart : numVec;
  uint32_t i = vBegin + tid;
  for (; i + 768u < vEnd; i += 1024u) {
    const uint4 v0 = streamLoad<kNt>(&pv[i]), v1 = streamLoad<kNt>(&pv[i + 256u]), v2 = streamLoad<kNt>(&pv[i + 512u]), v3 = streamLoad<kNt>(&pv[i + 768u]);
    histAdd4<S>(myBins, v0.x); histAdd4<S>(myBins, v0.y); histAdd4<S>(myBins, v0.z); histAdd4<S>(myBins, v0.w);
    histAdd4<S>(myBins, v1.x); histAdd4<S>(myBins, v1.y); histAdd4<S>(myBins, v1.z); histAdd4<S>(myBins, v1.w);
    histAdd4<S>(myBins, v2.x); histAdd4<S>(myBins, v2.y); histAdd4<S>(myBins, v2.z); histAdd4<S>(myBins, v2.w);
    histAdd4<S>(myBins, v3.x); histAdd4<S>(myBins, v3.y); histAdd4<S>(myBins, v3.z); histAdd4<S>(myBins, v3.w);
  }
  for (; i < vEnd; i += 256u) {
    const uint4 v = streamLoad<kNt>(&pv[i]);
    histAdd4<S>(myBins, v.x);
    histAdd4<S>(myBins, v.y);
    histAdd4<S>(myBins, v.z);
    histAdd4<S>(myBins, v.w);
  }

  if (w.part == 0) {
    uint32_t t = numVec * 16u + tid;
    if (t < remaining) histAdd<S>(myBins, p[head + t]);
  }
  __syncthreads();

  histStore(hist, partial, fuse, w, tid, histFold<S>(bins, tid));
}

// ---------------------------------------------------------------------------
// Checksum: XOR of all bytes, folded to 8 bits (GpuChecksum.cuh:26-93).
// grid = (xBlocks, B), 256 threads; out[] must be zeroed first.
// `floatWords` != 0 reproduces the float-path quirk: the provider's size is in
// float words but is consumed as a byte count, so only the first `size` bytes
// are covered (GpuFloatCompress.cuh:466-468).  `sizesOverride` (nullable) lets
// the decode side checksum exactly the decoded size; `onlyIf` (nullable) skips
// elements whose decode failed (their output buffer may be smaller than the size
// recorded in the archive: nothing of it may be read).
__global__ __launch_bounds__(256) void k_checksum(
    BatchView in, const uint32_t* __restrict__ sizesOverride, const uint8_t* __restrict__ onlyIf,
    uint32_t* __restrict__ out) {
  __shared__ uint32_t partial[4];
  const uint32_t tid = threadIdx.x;
  const uint32_t b = blockIdx.y;
  if (onlyIf && !onlyIf[b]) return;  // uniform
  const uint8_t* p = in.ptr(b);
  uint32_t size = sizesOverride ? sizesOverride[b] : in.size(b);

  uint32_t head = (uint32_t)((16u - ((uintptr_t)p & 15u)) & 15u);
  head = head < size ? head : size;
  const uint32_t remaining = size - head;
  const uint32_t numVec = remaining / 16u;
  const uint4* pv = (const uint4*)(p + head);

  uint32_t c = 0;
  if (blockIdx.x == 0 && tid < head) c ^= p[tid];
  for (uint32_t i = blockIdx.x * 256u + tid; i < numVec; i += gridDim.x * 256u) {
    uint4 v = pv[i];
    c ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (blockIdx.x == 0) {
    uint32_t i = numVec * 16u + tid;
    if (i < remaining) c ^= p[head + i];
  }
  c = (c ^ (c >> 8) ^ (c >> 16) ^ (c >> 24)) & 0xffu;
  c = waveReduceXor(c);
  if ((tid & 63u) == 0) partial[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    c = partial[0] ^ partial[1] ^ partial[2] ^ partial[3];
    if (c) atomicXor(&out[b], c);
  }
}

// ---------------------------------------------------------------------------
// Header info kernels (GpuANSInfo.cuh:17-37, GpuFloatInfo.cuh:18-41).
// One thread per batch element.
__global__ void k_ans_info(
    BatchView in, uint32_t numInBatch, uint32_t* outSizes, uint32_t* outChecksum) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= numInBatch) return;
  const AnsHeader* h = (const AnsHeader*)in.ptr(b);
  bool ok = h->magicAndVersion == ((kAnsMagic << 16) | kAnsVersion);
  if (outSizes) outSizes[b] = ok ? h->totalUncompressedWords : 0u;
  if (outChecksum) outChecksum[b] = ok ? h->checksum : 0u;
}

__global__ void k_float_info(
    BatchView in, uint32_t numInBatch, uint32_t* outSizes, uint32_t* outTypes,
    uint32_t* outChecksum) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= numInBatch) return;
  const FloatHeader* h = (const FloatHeader*)in.ptr(b);
  bool ok = h->magicAndVersion == ((kFloatMagic << 16) | kFloatVersion);
  if (outSizes) outSizes[b] = ok ? h->size : 0u;
  if (outTypes) outTypes[b] = ok ? (h->options & 0xfu) : 0u;
  if (outChecksum) outChecksum[b] = ok ? h->checksum : 0u;
}

}  // namespace dgpu
