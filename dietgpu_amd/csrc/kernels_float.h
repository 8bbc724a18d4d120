// Float split for gfx950: separates each float word into its compressible byte
// (exponent) and the non-compressible rest, writes the rest + the 16-byte
// GpuFloatHeader straight into the output archive, the exponent plane to temp
// memory, and accumulates the exponent histogram in per-wavefront LDS bins.
//
// Behavioural contract: FloatTypeInfo<FT>::split
// (dietgpu/float/GpuFloatUtils.cuh:111-115,141-147,181-185) and splitFloat
// (dietgpu/float/GpuFloatCompress.cuh:280-365).
#pragma once

#include "format.h"
#include "kernels_stats.h"

namespace dgpu {

struct SplitArgs {
  BatchView in;               // float words; size(b) in float words
  BatchView out;              // float archive base pointers
  uint8_t* compOut;           // [B][compStride] exponent plane (temp)
  uint32_t compStride;
  uint32_t useChecksum;
  const uint32_t* checksum;   // [B] nullable
  uint32_t* hist;             // [B][256], zeroed before launch
};

template <uint32_t FT>
__device__ __forceinline__ void splitWord(uint32_t w, uint32_t& comp, uint32_t& nonComp) {
  if (FT == kFloat16) {
    comp = w >> 8;
    nonComp = w & 0xffu;
  } else if (FT == kBFloat16) {
    // rotl32(w * 65537, 1): comp = exponent, nonComp = mantissa7 << 1 | sign
    comp = (w >> 7) & 0xffu;
    nonComp = ((w << 1) & 0xfeu) | (w >> 15);
  } else {
    uint32_t v = (w << 1) | (w >> 31);
    comp = v >> 24;
    nonComp = v & 0xffffffu;
  }
}

// grid = (xBlocks, B), 256 threads.
template <uint32_t FT>
__global__ __launch_bounds__(256) void k_float_split(SplitArgs a) {
  __shared__ uint32_t bins[kHistBlockWords];
  const uint32_t tid = threadIdx.x;
  const uint32_t b = blockIdx.y;
  histZero(bins, tid);
  __syncthreads();
  uint32_t* myBins = histMine(bins, tid);

  const uint32_t n = a.in.size(b);
  const uint8_t* inBytes = a.in.ptr(b);
  uint8_t* archive = a.out.ptr(b);
  uint8_t* comp = a.compOut + (size_t)b * a.compStride;

  if (blockIdx.x == 0 && tid == 0) {
    FloatHeader h;
    h.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
    h.size = n;
    h.options = FT | (a.useChecksum ? 0x10u : 0u);
    h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
    *(FloatHeader*)archive = h;
  }

  const bool aligned = (((uintptr_t)inBytes) & 15u) == 0;

  if (FT == kFloat16 || FT == kBFloat16) {
    const uint16_t* in = (const uint16_t*)inBytes;
    uint8_t* nc = archive + 16u;
    const uint32_t numVec = aligned ? n / 8u : 0u;  // 8 words = 16 bytes per step
    for (uint32_t v = blockIdx.x * 256u + tid; v < numVec; v += gridDim.x * 256u) {
      const uint4 x = ((const uint4*)in)[v];
      const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
      uint32_t c[2] = {0, 0}, r[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t c0, r0, c1, r1;
        splitWord<FT>(xw[j] & 0xffffu, c0, r0);
        splitWord<FT>(xw[j] >> 16, c1, r1);
        atomicAdd(&myBins[c0], 1u);
        atomicAdd(&myBins[c1], 1u);
        c[j >> 1] |= (c0 | (c1 << 8)) << (16 * (j & 1));
        r[j >> 1] |= (r0 | (r1 << 8)) << (16 * (j & 1));
      }
      ((uint2*)comp)[v] = make_uint2(c[0], c[1]);
      ((uint2*)nc)[v] = make_uint2(r[0], r[1]);
    }
    // tail (and the whole element when the input is not 16-byte aligned)
    for (uint32_t i = numVec * 8u + blockIdx.x * 256u + tid; i < n; i += gridDim.x * 256u) {
      uint32_t c0, r0;
      splitWord<FT>(in[i], c0, r0);
      atomicAdd(&myBins[c0], 1u);
      comp[i] = (uint8_t)c0;
      nc[i] = (uint8_t)r0;
    }
    // zero the non-comp padding up to 16 bytes
    if (blockIdx.x == 0) {
      const uint32_t padded = roundUp(n, 16u);
      if (n + tid < padded) nc[n + tid] = 0;
    }
  } else {
    const uint32_t* in = (const uint32_t*)inBytes;
    uint16_t* nc2 = (uint16_t*)(archive + 16u);
    uint8_t* nc1 = archive + 16u + 2u * (size_t)roundUp(n, 8u);
    const uint32_t numVec = aligned ? n / 4u : 0u;  // 4 words = 16 bytes per step
    for (uint32_t v = blockIdx.x * 256u + tid; v < numVec; v += gridDim.x * 256u) {
      const uint4 x = ((const uint4*)in)[v];
      const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
      uint32_t c = 0, hi = 0;
      uint32_t lo[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t c0, r0;
        splitWord<FT>(xw[j], c0, r0);
        atomicAdd(&myBins[c0], 1u);
        c |= c0 << (8 * j);
        hi |= (r0 >> 16) << (8 * j);
        lo[j >> 1] |= (r0 & 0xffffu) << (16 * (j & 1));
      }
      ((uint32_t*)comp)[v] = c;
      ((uint2*)nc2)[v] = make_uint2(lo[0], lo[1]);
      ((uint32_t*)nc1)[v] = hi;
    }
    for (uint32_t i = numVec * 4u + blockIdx.x * 256u + tid; i < n; i += gridDim.x * 256u) {
      uint32_t c0, r0;
      splitWord<FT>(in[i], c0, r0);
      atomicAdd(&myBins[c0], 1u);
      comp[i] = (uint8_t)c0;
      nc2[i] = (uint16_t)(r0 & 0xffffu);
      nc1[i] = (uint8_t)(r0 >> 16);
    }
    if (blockIdx.x == 0) {
      const uint32_t pad2 = roundUp(n, 8u);
      if (n + tid < pad2) nc2[n + tid] = 0;
      const uint32_t pad1 = roundUp(n, 16u);
      if (n + tid < pad1) nc1[n + tid] = 0;
    }
  }

  __syncthreads();
  const uint32_t sum = histFold(bins, tid);
  if (sum) atomicAdd(&a.hist[b * kNumSymbols + tid], sum);
}

}  // namespace dgpu
