// Wire format + batch addressing shared by every kernel.
//
// The archive layout is the reference's, byte for byte
// (dietgpu/ans/GpuANSUtils.cuh:17-229, dietgpu/float/GpuFloatUtils.cuh:19-74):
//
//   ANS archive  = [AnsHeader 32 B][u16 pdf[256]][u32 state[nb][32]]
//                  [uint2 blockWords[roundUp(nb,2)]][block data ...]
//   block i data starts at u16 offset blockWords[i].y, occupies
//   roundUp(compressedWords_i, 8) words; blockWords[i].x =
//   (uncompressedWords_i << 16) | compressedWords_i.
//   Float archive = [FloatHeader 16 B][non-comp plane(s), padded to 16 B]
//                   [ANS archive of the comp (exponent) plane]
//
// Bytes the reference leaves indeterminate are written as zero here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Streaming (non-temporal) access helpers and the cache policy of every stream the kernels touch.  The policies
// were chosen on ROTATING buffers -- inputs, archives and outputs that are not in the 256 MiB memory-side cache when
// their turn comes, as in real use -- not on a loop that re-codes one buffer set (docs/HISTORY.md section 3, "cache
// policy"; profiles/r03_ab_cache_policy_*.txt, profiles/r03_rotating_phases.txt):
//   * decoded float words are written once and not read again by the codec: non-temporal stores (1-byte
//     non-temporal stores are slow, so decoded raw bytes use ordinary ones)
//   * the encoder's input reads are a one-shot stream: non-temporal loads
//   * the histogram pass reads with non-temporal loads too.  Its policy is the one knob that is also a RUN-TIME
//     choice (dgpu_set_histogram_load_policy): with ORDINARY (allocating) loads the read-only histogram pass pushes
//     the dirty lines earlier kernels left in the memory-side cache out while it has write bandwidth to spare, and the
//     encoder then reads its input from that cache.  In a loop that compresses and at once decompresses on rotating
//     buffers this is worth 9 %; in every arrangement that resembles use -- compress only, compress after a producer
//     kernel has written the tensor, one buffer set -- it LOSES 4-20 %, so it is not the default
//   * archive stores and the decoder's archive loads stay cacheable: the consumer (decode, a send) follows soon
namespace dgpu {

constexpr bool kNtDecStores = true;
constexpr bool kNtEncStores = false;
constexpr bool kNtEncLoads = true;
constexpr bool kNtDecLoads = false;
constexpr bool kNtHistLoads = true;  // default; dgpu_set_histogram_load_policy overrides it at run time

typedef uint32_t dgpu_u32x4 __attribute__((ext_vector_type(4)));

template <bool kNt, typename T>
__device__ __forceinline__ void streamStore(T* p, const T& v) {
  if (kNt) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool kNt>
__device__ __forceinline__ void streamStore(uint4* p, const uint4& v) {
  if (kNt) __builtin_nontemporal_store(dgpu_u32x4{v.x, v.y, v.z, v.w}, (dgpu_u32x4*)p);
  else *p = v;
}
template <bool kNt>
__device__ __forceinline__ uint4 streamLoad(const uint4* p) {
  if (kNt) {
    const dgpu_u32x4 v = __builtin_nontemporal_load((const dgpu_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *p;
}


constexpr uint32_t kNumSymbols = 256;
constexpr uint32_t kBlockSize = 4096;      // kDefaultBlockSize, GpuANSUtils.cuh:37
constexpr uint32_t kLanesPerBlock = 32;    // the format's interleave (kWarpSize upstream)
constexpr uint32_t kRowsPerBlock = kBlockSize / kLanesPerBlock;
constexpr uint32_t kStateBits = 31;        // kANSStateBits
constexpr uint32_t kEncodedBits = 16;      // kANSEncodedBits
constexpr uint32_t kStartState = 1u << (kStateBits - kEncodedBits);  // 2^15
constexpr uint32_t kMinState = kStartState;
constexpr uint32_t kAnsMagic = 0xd00du;
constexpr uint32_t kAnsVersion = 0x0001u;
constexpr uint32_t kFloatMagic = 0xf00fu;
constexpr uint32_t kFloatVersion = 0x0001u;
constexpr uint32_t kBlockAlignWords = 8;   // 16 bytes of u16

constexpr uint32_t kFloat16 = 1, kBFloat16 = 2, kFloat32 = 3;

// Blocks handled by one 256-thread workgroup: 4 wave64 x 2 half-waves.
constexpr uint32_t kBlocksPerTile = 8;

struct alignas(32) AnsHeader {
  uint32_t magicAndVersion;
  uint32_t numBlocks;
  uint32_t totalUncompressedWords;
  uint32_t totalCompressedWords;
  uint32_t options;  // probBits | useChecksum << 4
  uint32_t checksum;
  uint32_t unused0;
  uint32_t unused1;
};
static_assert(sizeof(AnsHeader) == 32, "");

struct alignas(16) FloatHeader {
  uint32_t magicAndVersion;
  uint32_t size;     // float words
  uint32_t options;  // floatType | useChecksum << 4
  uint32_t checksum;
};
static_assert(sizeof(FloatHeader) == 16, "");

__host__ __device__ constexpr uint32_t divUp(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
__host__ __device__ constexpr uint32_t roundUp(uint32_t a, uint32_t b) { return divUp(a, b) * b; }

// ANSCoalescedHeader::getCompressedOverhead, GpuANSUtils.cuh:68-82
__host__ __device__ constexpr uint32_t ansOverhead(uint32_t nb) {
  return 32u + 2u * kNumSymbols + 128u * nb + 8u * roundUp(nb, 2u);
}
__host__ __device__ constexpr uint32_t ansStatesOffset() { return 32u + 2u * kNumSymbols; }
__host__ __device__ constexpr uint32_t ansBlockWordsOffset(uint32_t nb) { return ansStatesOffset() + 128u * nb; }

// FloatTypeInfo<FT>::getUncompDataSize, GpuFloatUtils.cuh:123-127,163-167,194-203
__host__ __device__ inline uint32_t floatUncompDataSize(uint32_t ft, uint32_t n) {
  if (ft == kFloat32) return 2u * roundUp(n, 8u) + roundUp(n, 16u);
  return roundUp(n, 16u);
}
// Offset of the embedded ANS archive inside a float archive (0 for raw ANS).
__host__ __device__ inline uint32_t ansOffsetInArchive(uint32_t ft, uint32_t n) {
  return ft ? 16u + floatUncompDataSize(ft, n) : 0u;
}
__host__ __device__ inline uint32_t floatWordBytes(uint32_t ft) { return ft == kFloat32 ? 4u : 2u; }

// One addressing object for every batch flavour of the reference
// (BatchProviderStride / Pointer / SplitSize, BatchProvider.cuh:39-194):
// either an explicit device array of addresses, or base + b * stride; either an
// explicit device array of sizes, or one uniform size.
struct BatchView {
  const uint64_t* ptrs;   // device array [B] of addresses, or nullptr
  uint64_t base;
  uint64_t stride;
  const uint32_t* sizes;  // device array [B], or nullptr
  uint32_t uniformSize;

  // The address is materialised as a GLOBAL (address space 1) pointer before it
  // decays to a generic one: batch addresses arrive as integers, and without
  // this the compiler can only emit FLAT loads/stores, which also tick the LDS
  // counter (lgkmcnt) and so serialise against every LDS wait in the rANS loops.
  __device__ __forceinline__ uint8_t* ptr(uint32_t b) const {
    typedef __attribute__((address_space(1))) uint8_t* GlobalBytes;
    return (uint8_t*)(GlobalBytes)(uintptr_t)(ptrs ? ptrs[b] : base + (uint64_t)b * stride);
  }
  __device__ __forceinline__ uint32_t size(uint32_t b) const {
    return sizes ? sizes[b] : uniformSize;
  }
};

}  // namespace dgpu
