// rANS decode for gfx950, with the float join fused into the write-out.
//
// Behavioural contract: ansDecodeTable / ansDecodeKernel / decodeOneWarp
// (dietgpu/ans/GpuANSDecode.cuh:34-476) and JoinFloatWriter / FloatOutProvider
// (dietgpu/float/GpuFloatDecompress.cuh:320-521).  Organisation for wave64:
//
//   * one wave64 decodes TWO 4 KiB blocks (lanes 0-31 / 32-63), one 64-bit
//     ballot per row; readers take their words from the top of the block down
//     in descending column order (the mirror of the encoder's ascending
//     emission order);
//   * the row loop is a dependent chain -- LUT entry (LDS), state update,
//     ballot, word address, word (LDS), renormalise -- and costs what that
//     chain's latency costs divided by the wavefronts that fit (DESIGN.md
//     section 3: extra instructions beside the chain are nearly free), so:
//       - the decode LUT holds 64-bit entries {pdf | sym << 24, x - cdf}: the
//         state update is one shift + one v_mad_u32_u24, and the symbol is
//         already in the top byte where the write-out wants it;
//       - compressed words sit in a 2 KiB LDS region per block: the whole
//         block when both blocks of the wave have <= 1024 words, else a ring of
//         4 x 512-byte chunks refilled one 8-row group ahead -- not a
//         worst-case 5 KiB stage, so LDS does not cap occupancy harder;
//       - the unread-word positions of the two halves are wave-uniform and live
//         in SGPRs (s_bcnt1 of the ballot halves); a lane's rank comes from
//         v_mbcnt_lo/hi over the ballot;
//       - aligned 16-bit float outputs: a row leaves {sym, 0} in a 512-byte LDS
//         buffer (ds_write_b16_d16_hi, no VALU); every 8 rows a lane joins its
//         8 consecutive words with the 8 non-compressed bytes it fetched with
//         ONE load and stores 16 bytes (RowSink::flushGroup).  Otherwise the
//         per-row sinks join and store one element per lane and row.
#pragma once
#include <type_traits>

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
struct DecodeArgs {
  BatchView in;            // archive pointers
  BatchView out;           // output pointers + capacities (bytes for raw, float words for float)
  uint32_t floatType;      // must equal the template FT
  uint8_t* outSuccess;     // [B] nullable
  uint32_t* outSize;       // [B] nullable
  const uint32_t* inBytes; // [B] nullable: bytes available at in.ptr(b) (the *_bounded entry points); an archive
                           // that claims to be longer is rejected instead of being read past its buffer
  uint32_t numInBatch;     // B
  uint32_t maxTiles;       // k_ans_decode: tiles per element the 1-D grid is laid out for
  uint32_t order;          // k_ans_decode: how workgroup index -> (element, tile), see decodeTileOf
  uint32_t uniformInBytes; // != 0 (and inBytes == nullptr): every archive has this many bytes available (stride batches)
  const uint32_t* workMap; // kDecOrderMap: [grid] element << 16 | tile, the tiles that exist (the host knows the capacities);
                           // k_ans_decode_pair: [numListed] the elements to pair up
  uint32_t numListed;      // k_ans_decode_pair with a workMap: its entries
};
__device__ __forceinline__ uint64_t decodeInBytes(const DecodeArgs& a, uint32_t b) {
  return a.inBytes ? (uint64_t)a.inBytes[b] : (a.uniformInBytes ? (uint64_t)a.uniformInBytes : ~0ull);
}

// Workgroup index -> (element, tile) of k_ans_decode's 1-D grid of B * maxTiles workgroups (kDecOrderXcd: B rounded up to 8).  The hardware
// dispatches workgroups in index order, workgroup w on XCD w mod 8 (docs/HISTORY.md section 3).
//   kDecOrderElementMajor: w = b * T + tile -- the tiles of an element run back to back (rounds 1-4)
//   kDecOrderTileMajor:    w = tile * B + b -- the order the encoder wrote the archives in
//   kDecOrderXcd:          every XCD walks ITS elements (b mod 8 == XCD) element-major: consecutive workgroups touch
//                          eight elements, and an element's header, pdf table and descriptors stay in one XCD's L2
//   kDecOrderMap:          batches whose capacities differ widely: the grid holds only the tiles inside each element's
//                          capacity, listed by the host (a rectangle laid out for the largest element would be mostly
//                          workgroups that start, find nothing and leave)
constexpr uint32_t kDecOrderElementMajor = 0, kDecOrderTileMajor = 1, kDecOrderXcd = 2, kDecOrderMap = 3;
__device__ __forceinline__ bool decodeTileOf(const DecodeArgs& a, uint32_t w, uint32_t* b, uint32_t* tile) {
  const uint32_t T = a.maxTiles, B = a.numInBatch;
  if (a.order == kDecOrderMap) {
    const uint32_t m = a.workMap[w];
    *b = m >> 16;
    *tile = m & 0xffffu;
    return true;
  }
  if (a.order == kDecOrderTileMajor) {
    *tile = w / B;
    *b = w - *tile * B;
    return *tile < T;
  }
  if (a.order == kDecOrderXcd) {
    const uint32_t x = w & 7u, g = w >> 3;
    const uint32_t e = g / T;
    *tile = g - e * T;
    *b = e * 8u + x;
    return *b < B;
  }
  *b = w / T;
  *tile = w - *b * T;
  return *b < B;
}

// ---------------------------------------------------------------------------
// Per-row sinks.  Each lane owns element row * 32 + hl of its block.  `e0` is
// LUT word 0 = pdf | sym << 24.  The float joins are FloatTypeInfo<FT>::join
// (GpuFloatUtils.cuh:117-119,149-159,187-190) arranged so that the result sits
// in the TOP half of a register and is stored with a d16_hi short store; the
// pdf bits that ride along in the low 12 bits never reach the stored half.
// Wide stores: with one 2-byte (1-byte) store per lane and row a half-wave writes 64-byte (32-byte) pieces, which
// the memory system turns into 1.3x the bytes (WRITE_SIZE, profiles/r02_write_calib.txt).  Instead the 8 rows of a
// group are transposed through a 512-byte LDS buffer per block and leave as ONE 16-byte (8-byte) store per lane:
// 512 (256) contiguous bytes (bf16 P=10 decode 100 -> 86 us, WRITE_SIZE back to the algorithmic bytes).  fp32 rows
// are 128-byte pieces already; raw bytes in 16-block tiles keep the narrow stores (decXposeBytes).
__device__ __forceinline__ uint2 decLoad8(const uint8_t* p) {
  typedef uint32_t u32x2n __attribute__((ext_vector_type(2)));
  if (kNtDecLoads) {
    const u32x2n v = __builtin_nontemporal_load((const u32x2n*)p);
    return make_uint2(v.x, v.y);
  }
  return *(const uint2*)p;
}
__host__ __device__ constexpr uint32_t decOutWordBytes(uint32_t ft) { return ft == 0u ? 1u : (ft == kFloat32 ? 4u : 2u); }
__device__ __forceinline__ uint4 decLoad16(const uint8_t* p) { return streamLoad<kNtDecLoads>((const uint4*)p); }
typedef __attribute__((address_space(3))) uint16_t LdsU16w;
typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4w LdsU4w;

// Per-row join (narrow path): {e0.b3 (symbol), r.b0 (non-compressed byte), e0.b1, e0.b0} in one v_perm_b32.
// (The byte loaded a group earlier still costs one v_and per row here: it crosses the loop edge as an i8 and the
// compiler sinks its zero-extension away from the load.  The wide path has no per-row byte at all.)
__device__ __forceinline__ uint32_t joinBytes(uint32_t e0, uint32_t r) { return __builtin_amdgcn_perm(e0, r, 0x07000504u); }

template <uint32_t FT>
struct RowSink;

template <>
struct RowSink<0> {  // raw bytes (BatchWriter, BatchProvider.cuh:16-37)
  uint8_t* out;
  __device__ __forceinline__ void init(uint8_t* outBase, const uint8_t*, uint32_t, size_t first, uint32_t hl) {
    out = outBase + first + hl;
  }
  typedef uint32_t Pre;
  __device__ __forceinline__ uint32_t prefetch(uint32_t) const { return 0; }
  __device__ __forceinline__ void store(uint32_t row, uint32_t e0, uint32_t) const {
    out[row * 32u] = (uint8_t)(e0 >> 24);  // 32-byte row pieces: left to the L2 to merge
  }
  // wide-store variant: every row leaves the TOP half of LUT word 0 ({sym, 0}) in a 512-byte LDS buffer as it is
  // (ds_write_b16_d16_hi: no VALU in the row); a lane then holds 8 consecutive symbols as four {sym, 0, sym, 0}
  // dwords, packs them with two v_perm_b32 and stores 8 bytes
  static constexpr uint32_t kXposeBytes = 512;
  __device__ __forceinline__ uint2 prefetchGroup(uint32_t, uint32_t) const { return make_uint2(0, 0); }
  __device__ __forceinline__ void stageRow(uint32_t xpose, uint32_t j, uint32_t hl, uint32_t e0) const {
    *(LdsU16w*)(uintptr_t)(xpose + (j * 32u + hl) * 2u) = (uint16_t)(e0 >> 16);
  }
  __device__ __forceinline__ void flushGroup(uint32_t xpose, uint32_t g, uint32_t hl, uint2) const {
    typedef uint32_t u32x2l __attribute__((ext_vector_type(2)));
    const u32x4w v = *(const LdsU4w*)(uintptr_t)(xpose + hl * 16u);
    const uint32_t o0 = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);  // {x.b1, x.b3, y.b1, y.b3}
    const uint32_t o1 = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
    *(u32x2l*)(out - hl + g * 256u + hl * 8u) = u32x2l{o0, o1};
  }
};

template <>
struct RowSink<kFloat16> {  // word = comp << 8 | nonComp
  uint16_t* out;
  const uint8_t* nc;
  __device__ __forceinline__ void init(uint8_t* outBase, const uint8_t* archive, uint32_t, size_t first, uint32_t hl) {
    out = (uint16_t*)outBase + first + hl;
    nc = archive + 16u + first + hl;
  }
  typedef uint32_t Pre;
  __device__ __forceinline__ uint32_t prefetch(uint32_t row) const { return nc[row * 32u]; }
  __device__ __forceinline__ void store(uint32_t row, uint32_t e0, uint32_t r) const {
    const uint32_t v = joinBytes(e0, r);  // [sym][nc][0000 pdf]
    streamStore<kNtDecStores>(&out[row * 32u], (uint16_t)(v >> 16));
  }
  // wide variant (see decodeBlock): rows stage {sym, 0}; a lane joins its 8 consecutive words with the 8
  // non-compressed bytes it loaded with ONE 8-byte load: one v_perm_b32 per pair of words
  static constexpr uint32_t kXposeBytes = 512;
  __device__ __forceinline__ uint2 prefetchGroup(uint32_t g, uint32_t hl) const { return decLoad8(nc - hl + g * 256u + hl * 8u); }
  __device__ __forceinline__ void stageRow(uint32_t xpose, uint32_t j, uint32_t hl, uint32_t e0) const {
    *(LdsU16w*)(uintptr_t)(xpose + (j * 32u + hl) * 2u) = (uint16_t)(e0 >> 16);
  }
  __device__ __forceinline__ void flushGroup(uint32_t xpose, uint32_t g, uint32_t hl, uint2 ncb) const {
    const u32x4w v = *(const LdsU4w*)(uintptr_t)(xpose + hl * 16u);
    u32x4w o;
    o.x = __builtin_amdgcn_perm(v.x, ncb.x, 0x07010500u);  // {nc0, sym0, nc1, sym1}
    o.y = __builtin_amdgcn_perm(v.y, ncb.x, 0x07030502u);
    o.z = __builtin_amdgcn_perm(v.z, ncb.y, 0x07010500u);
    o.w = __builtin_amdgcn_perm(v.w, ncb.y, 0x07030502u);
    u32x4w* dst = (u32x4w*)(out - hl + g * 256u + hl * 8u);
    if (kNtDecStores) __builtin_nontemporal_store(o, dst);
    else *dst = o;
  }
};

template <>
struct RowSink<kBFloat16> {  // word = (comp << 8 | nonComp) >> 1 | (nonComp & 1) << 15
  uint16_t* out;
  const uint8_t* nc;
  __device__ __forceinline__ void init(uint8_t* outBase, const uint8_t* archive, uint32_t, size_t first, uint32_t hl) {
    out = (uint16_t*)outBase + first + hl;
    nc = archive + 16u + first + hl;
  }
  typedef uint32_t Pre;
  __device__ __forceinline__ uint32_t prefetch(uint32_t row) const { return nc[row * 32u]; }
  __device__ __forceinline__ void store(uint32_t row, uint32_t e0, uint32_t r) const {
    const uint32_t lo = joinBytes(e0, r);                                 // [sym][nc][0000 pdf]
    const uint32_t v = __builtin_amdgcn_alignbit(r, lo, 1);     // (lo >> 1) | (r << 31)
    streamStore<kNtDecStores>(&out[row * 32u], (uint16_t)(v >> 16));                    // [sign][exp][mant7]
  }
  // wide variant: as fp16, then every 16-bit half {exp, nc} rotated right by one (sign to the top):
  // (h >> 1) + h * 2^15 on packed halves (v_pk_lshrrev_b16 + v_pk_mad_u16)
  static constexpr uint32_t kXposeBytes = 512;
  __device__ __forceinline__ uint2 prefetchGroup(uint32_t g, uint32_t hl) const { return decLoad8(nc - hl + g * 256u + hl * 8u); }
  __device__ __forceinline__ void stageRow(uint32_t xpose, uint32_t j, uint32_t hl, uint32_t e0) const {
    *(LdsU16w*)(uintptr_t)(xpose + (j * 32u + hl) * 2u) = (uint16_t)(e0 >> 16);
  }
  static __device__ __forceinline__ uint32_t rotrHalves(uint32_t x) {
    typedef uint16_t u16x2w __attribute__((ext_vector_type(2)));
    const u16x2w h = __builtin_bit_cast(u16x2w, x);
    const uint32_t shifted = __builtin_bit_cast(uint32_t, (u16x2w)(h >> 1));
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x80008000u), "v"(shifted));
    return r;
  }
  __device__ __forceinline__ void flushGroup(uint32_t xpose, uint32_t g, uint32_t hl, uint2 ncb) const {
    const u32x4w v = *(const LdsU4w*)(uintptr_t)(xpose + hl * 16u);
    u32x4w o;
    o.x = rotrHalves(__builtin_amdgcn_perm(v.x, ncb.x, 0x07010500u));
    o.y = rotrHalves(__builtin_amdgcn_perm(v.y, ncb.x, 0x07030502u));
    o.z = rotrHalves(__builtin_amdgcn_perm(v.z, ncb.y, 0x07010500u));
    o.w = rotrHalves(__builtin_amdgcn_perm(v.w, ncb.y, 0x07030502u));
    u32x4w* dst = (u32x4w*)(out - hl + g * 256u + hl * 8u);
    if (kNtDecStores) __builtin_nontemporal_store(o, dst);
    else *dst = o;
  }
};

template <>
struct RowSink<kFloat32> {  // rotr32(comp << 24 | nonComp24, 1)
  uint32_t* out;
  const uint16_t* nc2;  // u16 plane of roundUp(size, 8) entries
  const uint8_t* nc1;   // then the high-byte plane
  __device__ __forceinline__ void init(uint8_t* outBase, const uint8_t* archive, uint32_t floatSize, size_t first, uint32_t hl) {
    out = (uint32_t*)outBase + first + hl;
    nc2 = (const uint16_t*)(archive + 16u) + first + hl;
    nc1 = archive + 16u + 2u * (size_t)roundUp(floatSize, 8u) + first + hl;
  }
  typedef uint32_t Pre;
  __device__ __forceinline__ uint32_t prefetch(uint32_t row) const {
    return ((uint32_t)nc1[row * 32u] << 16) | nc2[row * 32u];
  }
  __device__ __forceinline__ void store(uint32_t row, uint32_t e0, uint32_t r) const {
    const uint32_t v = (e0 & 0xff000000u) | r;
    streamStore<kNtDecStores>(&out[row * 32u], (uint32_t)__builtin_amdgcn_alignbit(v, v, 1));
  }
  static constexpr uint32_t kXposeBytes = 0;  // 128-byte row pieces already: no transposition
  __device__ __forceinline__ uint2 prefetchGroup(uint32_t, uint32_t) const { return make_uint2(0, 0); }
  __device__ __forceinline__ void stageRow(uint32_t, uint32_t, uint32_t, uint32_t) const {}
  __device__ __forceinline__ void flushGroup(uint32_t, uint32_t, uint32_t, uint2) const {}
};

// ---------------------------------------------------------------------------
// Compressed-word ring: 4 chunks of 256 words (512 bytes) per block.  Chunk k of
// the block's data lives in slot k & 3.  Words are consumed from the top of the
// block downwards, at most 32 per row = 256 per 8-row group = one chunk per
// group, so at every group boundary the ring (a) receives the chunk requested
// one group earlier and (b) requests the next lower chunk once the unread
// position comes within 512 words of the lowest chunk it holds.  Reads of a
// group stay within 256 words below the position at its start, which the
// protocol guarantees to be resident (proof in DESIGN.md).
typedef __attribute__((address_space(3))) uint16_t LdsU16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 LdsU4;
constexpr uint32_t kRingChunkWords = 256;
constexpr uint32_t kRingBytes = 2048;
constexpr uint32_t kGroupRows = 8;

// A decode workgroup is 8 wavefronts = 16 blocks sharing one LUT: 32 KiB of
// rings + 8 KiB LUT (P = 10) lets 4 workgroups = 32 wavefronts (the maximum)
// reside on a CU.
constexpr uint32_t kDecBlocksPerTile = 16;
// Batches of small elements (a few blocks each) use 4-block workgroups instead:
// a 16-block workgroup would leave most of its waves without a block while
// still holding its LDS and wave slots.
constexpr uint32_t kDecBlocksPerSmallTile = 4;
// ... and one wavefront per element for batches of elements of at most 2 blocks
constexpr uint32_t kDecBlocksPerTinyTile = 2;
// ... and batches of single-block elements (every capacity <= 4096 symbols) go to k_ans_decode_pair
// (kernels_pairs.h): two ELEMENTS per wavefront
constexpr uint32_t kDecBlocksPerSingleTile = 1;
__host__ __device__ constexpr uint32_t decThreads(uint32_t tileBlocks) { return tileBlocks * 32u < 64u ? 64u : tileBlocks * 32u; }
// Store (transposition) buffers per block.  Raw bytes in 16-block tiles keep the narrow stores: a fourth
// workgroup per CU (40 KiB instead of 48: 8 waves per SIMD) is worth more to their row loop than the 8-byte
// stores (256 x 1 MiB Zipf bytes decode 152.5 -> 145 us).  16-bit floats are HBM-bound with the wide stores, and
// their narrow 2-byte non-temporal stores inflate the write traffic by 1.35 (docs/HISTORY.md section 4.2).
__host__ __device__ constexpr uint32_t decXposeBytes(int P, uint32_t ft, uint32_t tileBlocks) {
  (void)P;
  if (ft == 0 && tileBlocks >= 16u) return 0u;
  return ft == kFloat32 ? 0u : 512u;
}
// COMPACT LUT entries, 4 bytes {sym:8 | x - cdf:12 | pdf:12} instead of 8 (two more VALU per row to unpack):
//  * tiles of <= 4 blocks (batches of small elements): residency there is set by the LDS a workgroup needs for
//    its own LUT (measured on 32768 x 4 Ki: decode 228 / 323 / 476 us with a 4 / 8 / 16 KiB LUT; 8192 x 16 Ki:
//    139 -> 133 us), and a lone, latency-bound wavefront does not feel the unpacking;
//  * probBits 11 (any tile): the LUT has 2048 slots; compact, it takes the 8 KiB the 8-byte entries take at
//    probBits 10, which leaves room for the store buffers at the same 3 workgroups per CU (fp16 P=11 decode
//    105.5 -> 97.9 us).  At probBits 10 in 16-block tiles the compact entry is SLOWER (+4 .. +9 us on Zipf bytes):
//    its unpacking sits on the dependent chain.
__host__ __device__ constexpr bool decCompactLut(int P, uint32_t tileBlocks) { return tileBlocks <= 4u || P >= 11; }
__host__ __device__ constexpr uint32_t decLutBytes(int P, uint32_t tileBlocks) {
  return decCompactLut(P, tileBlocks) ? (4u << P) : (8u << P);
}
__host__ __device__ constexpr uint32_t decLdsBytes(int P, uint32_t ft, uint32_t tileBlocks) {
  return decLutBytes(P, tileBlocks) + tileBlocks * kRingBytes + tileBlocks * decXposeBytes(P, ft, tileBlocks);
}

// kIdleUpper (full path only): the upper half of the wave has no block (the element's block count is odd, or
// it has a single block): its lanes run the same straight-line code on don't-care data and only their
// stores are suppressed, so the lower half keeps the fast path instead of the predicated one.
//
// Position tracking (kFull): the unread-word counts of the two halves are wave-uniform, so
// they live in two SGPRs (s_bcnt1 of the ballot halves); a reading lane's word index comes from
// v_mbcnt_lo + v_mbcnt_hi over the 64-bit ballot (lower half: its rank; upper half: rank + readers of the lower
// half) plus one v_mad_i32_i24 that moves the upper half onto its own position: 4-5 VALU per row where the
// per-lane bookkeeping (half select by a 64-bit shift, two popcounts, subtract, mask, shift-add) took 7.
//
// kNoRing (kFull): both blocks of the wave have <= 1024 compressed words (every exponent block of N(0,1)
// bf16 has ~650), so the whole block is staged once and the ring maintenance, the wrap mask and the base OR
// disappear from the row loop (the base is folded into the scalar positions).
// Loads a workgroup issues BEFORE it builds its LUT, so that their round trip to memory overlaps with the build instead
// of following it (k_ans_decode: header -> {pdf, block descriptor, lane states} -> {compressed words, first
// non-compressed bytes} was a chain of four dependent round trips at the start of every workgroup; on cold buffers
// that is 10-15 % of a workgroup's life).  Held in registers across the build; valid for whole-block staging only.
struct DecodePre {
  uint4 words[4];  // the block's compressed words, 16 bytes per lane and k (kNoRing)
  uint2 nc[2];     // non-compressed bytes of the last two groups (join at the flush)
};
template <uint32_t FT>
__device__ __forceinline__ void decodePrefetch(DecodePre& pre, const uint8_t* __restrict__ gwords, uint32_t numWords,
                                               const RowSink<FT>& sink, uint32_t hl, bool wide) {
  const uint32_t paddedBytes = roundUp(numWords, kBlockAlignWords) * 2u;
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    const uint32_t off = (k * 32u + hl) * 16u;
    pre.words[k] = make_uint4(0, 0, 0, 0);
    if (off < paddedBytes) pre.words[k] = decLoad16(gwords + off);
  }
  pre.nc[0] = pre.nc[1] = make_uint2(0, 0);
  if (wide) {
    pre.nc[0] = sink.prefetchGroup(kRowsPerBlock / 8u - 1u, hl);
    pre.nc[1] = sink.prefetchGroup(kRowsPerBlock / 8u - 2u, hl);
  }
}

// kTail (kFull): blocks that are NOT full -- the last block of an element, single-block elements of any size.  Only a
// block's LAST row can have lanes without a symbol, so the rows split into `topRows` rows at the top (the partial row,
// what is left of an incomplete 8-row group and -- two elements per wavefront -- the rows one half has and the other
// has not), decoded in groups of eight by the predicated step with one-word stores, and `groups` whole groups below
// them in which every lane of a half that has a block holds a symbol: those run the straight-line step with the wide
// stores.  `n` = symbols of this half's block (0 with kIdleUpper: the upper half has none).
template <int P, uint32_t FT, bool kFull, bool kWide = false, bool kIdleUpper = false, bool kCompact = false, bool kNoRing = false,
          bool kPre = false, bool kTail = false>
__device__ __forceinline__ void decodeBlock(
    uint32_t xpose,                // kWide: LDS address of this half's transposition buffer
    uint32_t state,
    uint32_t n,                    // symbols in this half's block (0 = idle half)
    uint32_t groups,               // wave-uniform number of 8-row groups to run
    const uint8_t* __restrict__ gwords,  // global: this half's compressed words (16-byte aligned)
    uint32_t numWords,
    uint32_t ringBase,             // LDS address of this half's 2 KiB ring (multiple of 2048)
    const void* __restrict__ lutRaw, // LDS: uint2 entries, or uint32 entries (kCompact)
    const RowSink<FT>& sink,
    uint32_t hl,
    bool upper,
    const DecodePre* pre = nullptr,  // kPre: the staging loads were issued by the caller (decodePrefetch)
    uint32_t topRows = 0) {          // kTail: rows above the `groups` whole groups (wave-uniform)
  static_assert(!kPre || (kNoRing && kFull), "");
  static_assert(!kTail || (kFull && !kPre), "");
  constexpr uint32_t kMask = (1u << P) - 1u;
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  // LUT entry of slot x as {pdf | sym << 24, x - cdf} (the compact form is unpacked here)
  auto lutAt = [&](uint32_t x) -> uint2 {
    if (kCompact) {
      const uint32_t e = ((const uint32_t*)lutRaw)[x];
      return make_uint2((e & 0xff000fffu), (e >> 12) & 0xfffu);
    }
    return ((const uint2*)lutRaw)[x];
  };
  const uint32_t paddedBytes = roundUp(numWords, kBlockAlignWords) * 2u;

  // unread words of this half's block
  uint32_t posw = numWords;

  static_assert(!kNoRing || kFull, "whole-block staging is a full-block path");
  // initial fill: every chunk that intersects [numWords - 512, numWords)
  int lowChunk = numWords ? (int)((numWords - 1u) / kRingChunkWords) + 1 : 0;  // lowest chunk requested so far (+1 = none)
  if (kNoRing) {
    // the whole block (<= 1024 words = the 2 KiB ring region), word i at ringBase + 2 i
    uint4 v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const uint32_t off = (k * 32u + hl) * 16u;
      if (kPre) {
        v[k] = pre->words[k];
      } else {
        v[k] = make_uint4(0, 0, 0, 0);
        if (off < paddedBytes) v[k] = decLoad16(gwords + off);
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const uint32_t off = (k * 32u + hl) * 16u;
      if (off < paddedBytes) *(LdsU4*)(uintptr_t)(ringBase + off) = u32x4{v[k].x, v[k].y, v[k].z, v[k].w};
    }
    lowChunk = 0;
  } else {
    const int stop = numWords > 512u ? (int)((numWords - 512u) / kRingChunkWords) : 0;
    while (lowChunk > stop) {
      --lowChunk;
      const uint32_t off = (uint32_t)lowChunk * 512u + hl * 16u;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (off < paddedBytes) v = decLoad16(gwords + off);
      *(LdsU4*)(uintptr_t)(ringBase | (((uint32_t)lowChunk & 3u) * 512u + hl * 16u)) = u32x4{v.x, v.y, v.z, v.w};
    }
  }
  uint4 pending = make_uint4(0, 0, 0, 0);
  int pendingChunk = -1;

  // Generic step (partial blocks): predicated, the word read under a branch.
  auto step = [&](bool valid) -> uint32_t {
    const uint2 e = lutAt(state & kMask);
    if (valid) state = __umul24(e.x, state >> P) + e.y;
    const bool read = valid && (state < kMinState);
    const uint64_t vote = __ballot(read);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    posw -= __popc(vh);
    if (read) {
      // reading lanes take words from the top down in descending lane order:
      // word index = (new position) + number of reading lanes below me
      const uint32_t idx = posw + __popc(vh & laneMaskLt);
      const uint32_t w = *(const LdsU16*)(uintptr_t)(((idx << 1) & (kRingBytes - 1u)) | ringBase);
      state = (state << kEncodedBits) | w;
    }
    return e.x;
  };

  // Full-block step: straight-line code, no exec-mask change and no branch.
  // Every lane reads a ring word (the address always falls inside the ring); only
  // lanes that need to renormalise keep it.
  // wave-uniform positions (SGPRs): unread words of the lower / upper half's block; kNoRing: plus the LDS word
  // address of the block's staging area, so that (position + rank) << 1 IS the LDS address
  typedef typename RowSink<FT>::Pre Pre;
  uint32_t sLo = 0, sHi = 0;
  // an idle upper half follows the lower half's addresses (in bounds; its words are never used)
  int upperSel = (upper && !kIdleUpper) ? 1 : 0;
  asm volatile("" : "+v"(upperSel));  // a VGPR operand of the multiply-add, not a select to be folded into it
  auto stepFull = [&]() -> uint32_t {
    const uint2 e = lutAt(state & kMask);
    state = __umul24(e.x, state >> P) + e.y;
    const bool read = state < kMinState;
    const uint64_t vote = __ballot(read);
    const uint32_t vLo = (uint32_t)vote, vHi = (uint32_t)(vote >> 32);
    const uint32_t sLoOld = sLo;
    sLo -= (uint32_t)__popc(vLo);
    sHi -= (uint32_t)__popc(vHi);
    // readers below me in the wave: lower half = my rank, upper half = rank + readers of the lower half
    uint32_t t = __builtin_amdgcn_mbcnt_hi(vHi, __builtin_amdgcn_mbcnt_lo(vLo, 0u));
    // lower: sLo + rank; upper: sHi + rank = sLo + (rank + readersLo) + (sHi - sLoOld)
    t = (uint32_t)(__mul24(upperSel, (int)(sHi - sLoOld)) + (int)t);
    asm volatile("" : "+v"(t));  // keep the scalar position in the add-shift below (one SGPR operand per VALU op)
    const uint32_t addr = kNoRing ? ((t + sLo) << 1) : ((((t + sLo) << 1) & (kRingBytes - 1u)) | ringBase);
    const uint32_t w = *(const LdsU16*)(uintptr_t)addr;
    state = read ? ((state << kEncodedBits) | w) : state;
    return e.x;
  };

  // Groups gHi .. gLo of the block, top down; F = every lane of a half that has a block holds a symbol in each of them
  // (the straight-line step, wide stores where the sink has them), else the predicated step with one-word stores.
  //
  // Non-compressed bytes are fetched ahead of their use: TWO groups on the wide path (a group takes ~1.5 us with three
  // workgroups per CU -- one group ahead did not cover the latency of HBM under load on cold buffers), one otherwise.
  //  * wide path: ONE 8-byte load per lane and group -- the 8 consecutive bytes that belong to the 8 consecutive
  //    words the lane stores; they meet the decoded symbols only at the flush (RowSink::flushGroup), so a row
  //    costs no VALU for the join and no byte load;
  //  * otherwise one byte load per row, unconditional (row index clamped) so that the compiler can use counted
  //    vmcnt waits instead of draining the memory queue every group.
  auto runGroups = [&](auto fullTag, const int gHi, const int gLo) {
    constexpr bool F = decltype(fullTag)::value;
    constexpr bool kJoinAtFlush = F && kWide;
    Pre preCur[kGroupRows], preNext[kGroupRows];
    uint2 ncCur = make_uint2(0, 0), ncNext = make_uint2(0, 0), ncNext2 = make_uint2(0, 0);
    if (kJoinAtFlush) {
      if (kPre) {
        ncCur = pre->nc[0];
        ncNext = pre->nc[1];
      } else {
        ncCur = sink.prefetchGroup((uint32_t)gHi, hl);
        ncNext = sink.prefetchGroup(gHi > gLo ? (uint32_t)(gHi - 1) : (uint32_t)gLo, hl);
      }
    }
#pragma unroll
    for (int j = 0; j < (int)kGroupRows; ++j) {
      const uint32_t row = (uint32_t)gHi * kGroupRows + j;
      preCur[j] = kJoinAtFlush ? (Pre)0 : ((F || row * 32u + hl < n) ? sink.prefetch(row) : (Pre)0);
    }

#pragma unroll 1
    for (int g = gHi; g >= gLo; --g) {
      const uint32_t gNext = g > gLo ? (uint32_t)(g - 1) : (uint32_t)gLo;
      if (kJoinAtFlush) {
        ncNext2 = sink.prefetchGroup(g > gLo + 1 ? (uint32_t)(g - 2) : (uint32_t)gLo, hl);
      } else {
#pragma unroll
        for (int j = 0; j < (int)kGroupRows; ++j) {
          const uint32_t row = gNext * kGroupRows + j;
          preNext[j] = (F || row * 32u + hl < n) ? sink.prefetch(row) : (Pre)0;
        }
      }
      // ring maintenance
      if (F && !kNoRing) posw = upper ? sHi : sLo;
      if (!kNoRing) {
        if (pendingChunk >= 0) {
          *(LdsU4*)(uintptr_t)(ringBase | (((uint32_t)pendingChunk & 3u) * 512u + hl * 16u)) = u32x4{pending.x, pending.y, pending.z, pending.w};
          pendingChunk = -1;
        }
        if (lowChunk > 0 && (uint32_t)lowChunk * kRingChunkWords + 512u > posw) {
          --lowChunk;
          const uint32_t off = (uint32_t)lowChunk * 512u + hl * 16u;
          pending = (off < paddedBytes) ? decLoad16(gwords + off) : make_uint4(0, 0, 0, 0);
          pendingChunk = lowChunk;
        }
      }
#pragma unroll
      for (int j = (int)kGroupRows - 1; j >= 0; --j) {
        const uint32_t row = (uint32_t)g * kGroupRows + j;
        if (F) {
          const uint32_t e0 = stepFull();
          if (kWide) {
            if (!kIdleUpper || !upper) sink.stageRow(xpose, (uint32_t)j, hl, e0);
          } else if (!kIdleUpper || !upper) {
            sink.store(row, e0, preCur[j]);
          }
        } else {
          const bool valid = row * 32u + hl < n;
          const uint32_t e0 = step(valid);
          if (valid) sink.store(row, e0, preCur[j]);
        }
      }
      if (kJoinAtFlush) {
        if (!kIdleUpper || !upper) sink.flushGroup(xpose, (uint32_t)g, hl, ncCur);
        ncCur = ncNext;
        ncNext = ncNext2;
      } else {
#pragma unroll
        for (int j = 0; j < (int)kGroupRows; ++j) preCur[j] = preNext[j];
      }
    }
  };

  if (kTail) {
    // the groups above the whole ones, predicated (see the comment at the template); the ring, where there is one, is
    // kept by the same per-group maintenance in both phases
    if (topRows != 0u) runGroups(std::false_type{}, (int)(groups + divUp(topRows, kGroupRows)) - 1, (int)groups);
    if (groups == 0u) return;  // uniform: the block had no whole group
  }
  if (kFull) {
    // wave-uniform positions (SGPRs): unread words of the lower / upper half's block (numWords, less what the groups
    // above consumed); kNoRing: plus the LDS word address of the block's staging area, so that (position + rank) << 1
    // IS the LDS address
    sLo = __builtin_amdgcn_readlane(posw, 0);
    sHi = __builtin_amdgcn_readlane(posw, 32);
    if (kNoRing) {
      sLo += __builtin_amdgcn_readlane(ringBase, 0) >> 1;
      sHi += __builtin_amdgcn_readlane(ringBase, 32) >> 1;
    }
  }
  runGroups(std::integral_constant<bool, kFull>{}, (int)groups - 1, 0);
}

// grid = (maxTiles, B), 32 threads per block of the tile (512 or 128), LDS = word rings + 64-bit LUT.
template <int P, uint32_t FT, uint32_t kTileBlocks>
__global__ __launch_bounds__(decThreads(kTileBlocks)) void k_ans_decode(DecodeArgs a) {
  constexpr uint32_t kDecThreads = decThreads(kTileBlocks);
  // the LUT-build scratch (cdf, pdf: 2 KiB, read while the LUT is stored) sits in the ring area
  static_assert(kTileBlocks * kRingBytes >= 2048u, "");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // rings: 16 x 2 KiB at LDS offset 0 (2 KiB aligned), then the LUT, then the transposition buffers
  uint2* sLut = (uint2*)(smem + kTileBlocks * kRingBytes);
  constexpr uint32_t kXpose = decXposeBytes(P, FT, kTileBlocks);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);
  uint32_t b, tile;
  if (!decodeTileOf(a, blockIdx.x, &b, &tile)) return;  // uniform (kDecOrderXcd pads the grid to a multiple of 8 elements)

  const uint8_t* archive = a.in.ptr(b);
  uint32_t floatSize = 0;
  const uint64_t inBytes = decodeInBytes(a, b);
  if (inBytes < (FT ? sizeof(FloatHeader) : sizeof(AnsHeader))) {  // uniform: not even a header
    if (tile == 0 && tid == 0) {
      if (a.outSuccess) a.outSuccess[b] = 0;
      if (a.outSize) a.outSize[b] = 0;
    }
    return;
  }
  if (FT) {
    // the float header locates the ANS archive: check it before following it
    // (GpuFloatHeader::checkMagicAndVersion / getFloatType asserts, GpuFloatUtils.cuh:31-41, GpuFloatDecompress.cuh:332-382)
    const FloatHeader fh = *(const FloatHeader*)archive;
    const bool fhOk = fh.magicAndVersion == ((kFloatMagic << 16) | kFloatVersion) && (fh.options & 0xfu) == FT &&
        fh.size <= a.out.size(b) &&
        (uint64_t)sizeof(FloatHeader) + floatUncompDataSize(FT, fh.size) + sizeof(AnsHeader) <= inBytes;
    if (!fhOk) {  // uniform
      if (tile == 0 && tid == 0) {
        if (a.outSuccess) a.outSuccess[b] = 0;
        if (a.outSize) a.outSize[b] = fh.size;
      }
      return;
    }
  }
  const uint8_t* ans = locateAns(archive, FT, &floatSize);
  // Everything at an offset from the ANS archive that is known NOW is requested in one round trip with the header:
  // the pdf table (wave 0) and -- float archives: a valid one holds ceil(size / 4096) blocks, which the header check
  // below confirms -- this half-wave's block descriptor and lane states.  A part is requested early only where it is
  // known to lie inside the caller's buffer: the caller said how many bytes there are, or (archives of unknown extent)
  // a checked float header says that an ANS archive of that many blocks follows.  Raw ANS archives learn their block
  // count from the header and fetch descriptor and states after it, as before.  A load that is not to be made yet
  // reads the first bytes of the header again.
  const uint32_t block = tile * kTileBlocks + hw;
  const bool extentKnown = inBytes != ~0ull;
  const uint32_t ansOff = ansOffsetInArchive(FT, floatSize);
  const uint32_t nbSpec = FT ? divUp(floatSize, kBlockSize) : 0u;
  const bool pdfEarly = extentKnown ? (uint64_t)ansOff + ansOverhead(0u) <= inBytes : FT != 0u;
  const bool blockEarly = FT != 0u && block < nbSpec && (!extentKnown || (uint64_t)ansOff + ansOverhead(nbSpec) <= inBytes);
  const AnsHeader header = *(const AnsHeader*)ans;
  uint2 pdfRaw = make_uint2(0u, 0u);
  if (wave == 0) pdfRaw = *(const uint2*)(pdfEarly ? ans + sizeof(AnsHeader) + 8u * lane : ans);  // pdf[4 lane .. 4 lane + 3]
  uint2 bwMine = *(const uint2*)(blockEarly ? ans + ansBlockWordsOffset(nbSpec) + 8u * block : ans);
  uint32_t state = *(const uint32_t*)(blockEarly ? ans + ansStatesOffset() + 4u * (block * 32u + hl) : ans);
  const uint32_t nb = header.numBlocks;
  const uint32_t total = header.totalUncompressedWords;
  const uint32_t totalWords = header.totalCompressedWords;

  // Is there room for the output?  Capacity and size are in the API's units
  // (bytes for raw ANS, float words for the float codec); GpuANSDecode.cuh:325-341.
  // The format invariants the reference only asserts (GpuANSDecode.cuh:323,448,
  // GpuANSUtils.cuh:109-112) are CHECKED here: a corrupt or truncated archive is
  // reported through outSuccess instead of being followed out of bounds.  Every
  // quantity used for addressing below is bounded by the caller's capacity:
  // total <= capacity, nb == ceil(total / 4096), block sizes as the format
  // prescribes, block data inside [0, totalCompressedWords).
  bool success = a.out.size(b) >= total;
  success = success && header.magicAndVersion == ((kAnsMagic << 16) | kAnsVersion) &&
      (header.options & 0xfu) == (uint32_t)P;
  if (FT) success = success && floatSize == total;
  success = success && nb == divUp(total, kBlockSize);
  // the archive as a whole fits the bytes the caller has (when told)
  success = success && (uint64_t)ansOffsetInArchive(FT, total) + ansOverhead(nb) + 2ull * totalWords <= inBytes;
  if (!success) {  // uniform: nothing else of the archive is read
    if (tile == 0 && tid == 0) {
      if (a.outSuccess) a.outSuccess[b] = 0;
      if (a.outSize) a.outSize[b] = total;
    }
    return;
  }
  if (tile * kTileBlocks >= nb && tile != 0) return;

  const uint2* blockWords = (const uint2*)(ans + ansBlockWordsOffset(nb));
  // {uncompressed words << 16 | compressed words, start} of block i must be what an
  // encoder can produce: the size the format prescribes, 16-byte aligned data inside the archive
  auto blockOk = [&](uint32_t i, uint2 bw) -> bool {
    const uint32_t want = (i + 1u < nb) ? kBlockSize : total - i * kBlockSize;
    const uint32_t words = bw.x & 0xffffu;
    return (bw.x >> 16) == want && (bw.y & (kBlockAlignWords - 1u)) == 0u &&
        (uint64_t)bw.y + roundUp(words, kBlockAlignWords) <= (uint64_t)totalWords;
  };
  // tile 0 vouches for the whole element (it alone writes outSuccess); its loads are
  // issued here and checked after the LUT build
  bool allBlocksOk = true;
  if (tile == 0) {
    for (uint32_t i = tid; i < nb; i += kDecThreads) allBlocksOk = allBlocksOk && blockOk(i, blockWords[i]);
  }

  // This half-wave's block descriptor and lane states and (wave 0) the pdf table: already on their way (float
  // archives, above), or requested NOW, in the same round trip as tile 0's descriptor checks; as soon as the descriptor
  // is there the block's compressed words and its first non-compressed bytes are requested too (decodePrefetch), and
  // all of it lands while the LUT is being built.
  bool haveBlock = block < nb;
  if (!haveBlock) {
    bwMine = make_uint2(0u, 0u);
    state = 0;
  } else if (!blockEarly) {  // (for a float archive the checked header has confirmed nb == nbSpec)
    bwMine = blockWords[block];
    state = ((const uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl];  // (inside the archive: nb was checked against inBytes)
  }
  if (wave == 0 && !pdfEarly) pdfRaw = ((const uint2*)(ans + sizeof(AnsHeader)))[lane];
  uint32_t n = 0, numWords = 0, start = 0;
  if (haveBlock) {
    if (blockOk(block, bwMine)) {
      n = bwMine.x >> 16;
      numWords = bwMine.x & 0xffffu;
      start = bwMine.y;
    } else {
      haveBlock = false;  // malformed block: neither read nor written (tile 0 reports the element)
      state = 0;
    }
  }
  const uint8_t* gwords = ans + ansOverhead(nb) + 2u * (size_t)start;
  RowSink<FT> sink;
  // (a half without a block points at its wave's first block: its prefetches must stay inside the archive)
  sink.init(a.out.ptr(b), archive, floatSize, (size_t)(haveBlock ? block : (block & ~1u)) * kBlockSize, hl);
  // LDS address of the dynamic segment (0: this kernel has no static LDS); the
  // rings must be 2 KiB aligned for the and-or addressing in decodeBlock
  const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  // uniform per wave: both halves hold full blocks?  small enough to be staged whole?
  const uint32_t nFirst = __shfl(n, 0, 64);
  const uint32_t nSecond = __shfl(n, 32, 64);
  const uint32_t wFirst = __shfl(numWords, 0, 64);
  const uint32_t wSecond = __shfl(numWords, 32, 64);
  const bool noRing = wFirst <= kRingBytes / 2u && wSecond <= kRingBytes / 2u;
  // (the wide stores cover whole 8-row groups of words that exist; they take any word-aligned output element)
  const bool wide = decXposeBytes(P, FT, kTileBlocks) != 0 && (((uintptr_t)a.out.ptr(b)) & (decOutWordBytes(FT) - 1u)) == 0;
  const bool fullPair = nFirst == kBlockSize && nSecond == kBlockSize;
  // one full block in the wave (odd block counts): fast path with an idle upper half
  const bool fullSingle = nFirst == kBlockSize && nSecond == 0u;
  DecodePre pre;
  if ((fullPair || fullSingle) && noRing) decodePrefetch<FT>(pre, gwords, numWords, sink, hl, wide);

  // Decode LUT, built by the workgroup itself from the archive's pdf table (no
  // separate table kernel / LUT round trip through HBM):
  //   lut[x] = { pdf[sym] | sym << 24,  x - cdf[sym] },  sym = symbol whose cdf range holds x
  // (the information of packDecodeLookup, GpuANSDecode.cuh:34-41, re-packed so
  // that v_mad_u32_u24 can take word 0 directly: its low 24 bits are the pdf).
  // Wave 0 scans the 256 pdfs on its own (4 per lane), so one barrier suffices;
  // the cdf/pdf scratch lives in the ring area, which is not in use yet.
  uint32_t* sCdf = (uint32_t*)smem;
  uint32_t* sPdf = sCdf + kNumSymbols;
  uint32_t* sPdfSum = sPdf + kNumSymbols;
  uint32_t* sWaveBad = sPdfSum + 1;  // one flag per wave (<= 8); no static LDS in this kernel (ring alignment)
  // small tiles (one LUT per 4 blocks): the LUT is filled by a max-scan over symbol marks instead of a
  // binary search per slot (see below)
  constexpr bool kScanLut = kTileBlocks <= kDecBlocksPerSmallTile;
  uint32_t* sWaveTop = sWaveBad + 8;                  // kScanLut: running maximum at the end of each wave
  // kScanLut: 2^P mark bytes in the TAIL of the LUT region -- every mark has been read (into registers) before the
  // barrier that precedes the first LUT store
  constexpr bool kCompact = decCompactLut(P, kTileBlocks);
  constexpr uint32_t kLutBytes = decLutBytes(P, kTileBlocks);
  uint8_t* sMark = (uint8_t*)sLut + kLutBytes - (1u << P);
  {
    {
      const bool waveBad = __ballot(!allBlocksOk) != 0ull;
      if (lane == 0u) sWaveBad[wave] = waveBad ? 1u : 0u;
    }
    if (kScanLut) {
      for (uint32_t i = tid; i < (1u << P) / 4u; i += kDecThreads) ((uint32_t*)sMark)[i] = 0u;
    }
    if (wave == 0) {
      const uint2 raw = pdfRaw;
      const uint32_t p0 = raw.x & 0xffffu, p1 = raw.x >> 16, p2 = raw.y & 0xffffu, p3 = raw.y >> 16;
      const uint32_t mine = p0 + p1 + p2 + p3;
      const uint32_t incl = waveInclusiveScan(mine, lane);
      const uint32_t base = incl - mine;
      ((uint4*)sCdf)[lane] = make_uint4(base, base + p0, base + p0 + p1, base + p0 + p1 + p2);
      ((uint4*)sPdf)[lane] = make_uint4(p0, p1, p2, p3);
      if (lane == 63u) *sPdfSum = incl;
    }
    __syncthreads();
    // the reduction of tile 0's block checks
    uint32_t anyBad = 0;
#pragma unroll
    for (uint32_t w = 0; w < kDecThreads / 64u; ++w) anyBad |= sWaveBad[w];
    const bool blocksOk = anyBad == 0u;
    // the probabilities of a non-empty element sum to 2^P (GpuANSStatistics.cuh:256-316)
    const bool pdfOk = nb == 0u || *sPdfSum == (1u << P);
    if (tile == 0 && tid == 0) {
      if (a.outSuccess) a.outSuccess[b] = (blocksOk && pdfOk) ? 1 : 0;
      if (a.outSize) a.outSize[b] = total;
    }
    if (!pdfOk || tile * kTileBlocks >= nb) return;  // uniform
    if constexpr (kScanLut) {
      // mark[cdf[s]] = s for every present symbol; sym(x) = the largest mark at or below x (cdfs ascend with
      // the symbol; slot 0 belongs to the first present symbol, so "no mark" = 0 never surfaces)
      for (uint32_t sidx = tid; sidx < kNumSymbols; sidx += kDecThreads) {
        if (sPdf[sidx]) sMark[sCdf[sidx]] = (uint8_t)sidx;
      }
      __syncthreads();
      // Every wave owns a contiguous part of the table and walks it 256 slots per step: a lane reads ONE dword of
      // marks (four consecutive slots) and writes ONE 16-byte vector of compact entries, both at consecutive
      // addresses across the lanes -- (1 << P) / threads CONSECUTIVE slots per thread meant 8 ... 16 entry stores at an
      // 8 ... 16-dword stride, i.e. that many lanes per LDS bank (the conflict k_ans_decode_pair's build had: section 4.4).
      // All marks are in registers before the barrier that precedes the first entry store (they live in the LUT's tail).
      static_assert(kCompact, "the scan build writes compact entries");
      constexpr uint32_t kWaves = kDecThreads / 64u;
      constexpr uint32_t kPart = (1u << P) / kWaves;  // slots per wave
      static_assert(kPart % 256u == 0u, "");
      constexpr uint32_t kSteps = kPart / 256u;
      auto dmax = [](uint32_t v, uint32_t o) { return v > o ? v : o; };
      auto waveMaxScan = [&](uint32_t v) -> uint32_t {  // inclusive, DPP
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
        v = dmax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
        return v;
      };
      uint32_t m4[kSteps];
      uint32_t top = 0;
#pragma unroll
      for (uint32_t t = 0; t < kSteps; ++t) {
        m4[t] = ((const uint32_t*)sMark)[(wave * kPart + t * 256u) / 4u + lane];
        const uint32_t a01 = dmax(m4[t] & 0xffu, (m4[t] >> 8) & 0xffu);
        const uint32_t a23 = dmax((m4[t] >> 16) & 0xffu, m4[t] >> 24);
        top = dmax(top, dmax(a01, a23));
      }
      top = waveMaxScan(top);
      if (lane == 63u) sWaveTop[wave] = top;  // the largest mark of this wave's part
      __syncthreads();
      uint32_t carry = 0;  // the largest mark before the step (uniform)
      for (uint32_t w = 0; w < wave; ++w) carry = dmax(carry, sWaveTop[w]);
#pragma unroll
      for (uint32_t t = 0; t < kSteps; ++t) {
        uint32_t run[4];
        run[0] = m4[t] & 0xffu;
        run[1] = dmax(run[0], (m4[t] >> 8) & 0xffu);
        run[2] = dmax(run[1], (m4[t] >> 16) & 0xffu);
        run[3] = dmax(run[2], m4[t] >> 24);
        const uint32_t incl = waveMaxScan(run[3]);
        uint32_t excl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, true);  // wave_shr:1 (lane 0: 0)
        excl = dmax(excl, carry);
        uint32_t e[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
          const uint32_t x = wave * kPart + t * 256u + lane * 4u + j;
          const uint32_t sym = dmax(run[j], excl);
          e[j] = (sPdf[sym] & 0xfffu) | (((x - sCdf[sym]) & 0xfffu) << 12) | (sym << 24);
        }
        ((uint4*)sLut)[(wave * kPart + t * 256u) / 4u + lane] = make_uint4(e[0], e[1], e[2], e[3]);
        carry = dmax(carry, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
      }
    } else
    for (uint32_t x = tid; x < (1u << P); x += kDecThreads) {
      // last symbol s with cdf[s] <= x (zero-pdf symbols share the cdf of their
      // successor and are skipped by taking the last one)
      uint32_t lo = 0, hi = kNumSymbols;  // invariant: cdf[lo] <= x, answer in [lo, hi)
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
  }

  __syncthreads();  // LUT visible to every wave, scratch free (each half-wave's ring is private to its wave)

  constexpr uint32_t kLutBytesHere = decLutBytes(P, kTileBlocks);
  const uint32_t xpose = ldsBase + kTileBlocks * kRingBytes + kLutBytesHere + hw * kXpose;
  const uint32_t ringLds = ldsBase + hw * kRingBytes;
#define DGPU_DECODE_FULL(WIDE, IDLE, NORING) \
  decodeBlock<P, FT, true, WIDE, IDLE, kCompact, NORING, NORING>(xpose, state, n, kRowsPerBlock / kGroupRows, gwords, numWords, ringLds, sLut, sink, hl, upper, &pre)
  if (fullPair) {
    if (wide) {
      if (noRing) DGPU_DECODE_FULL(true, false, true); else DGPU_DECODE_FULL(true, false, false);
    } else {
      if (noRing) DGPU_DECODE_FULL(false, false, true); else DGPU_DECODE_FULL(false, false, false);
    }
  } else if (fullSingle) {
    if (wide) {
      if (noRing) DGPU_DECODE_FULL(true, true, true); else DGPU_DECODE_FULL(true, true, false);
    } else {
      if (noRing) DGPU_DECODE_FULL(false, true, true); else DGPU_DECODE_FULL(false, true, false);
    }
#undef DGPU_DECODE_FULL
  } else {
    // A wavefront with a block that is not full -- the element's last: (full, partial) or (partial, none).  The whole
    // 8-row groups in which every lane of a half that has a block holds a symbol run the straight-line step, the rows
    // above them the predicated one (decodeBlock, kTail); in ring mode only if those rows fit the ring's initial fill.
    const uint32_t maxN = nFirst > nSecond ? nFirst : nSecond;
    const uint32_t maxRows = divUp(maxN, 32u);
    const uint32_t fullRows = (nSecond ? nSecond : nFirst) / 32u;  // (the last block of the wave is the partial one)
    const uint32_t groups = fullRows / kGroupRows;
    const uint32_t topRows = maxRows - groups * kGroupRows;
    const bool tailPair = nFirst == kBlockSize && nSecond != 0u, tailSingle = nFirst != 0u && nSecond == 0u;
    if ((tailPair || tailSingle) && groups != 0u) {
#define DGPU_DECODE_TAIL(WIDE, IDLE, NORING) \
  decodeBlock<P, FT, true, WIDE, IDLE, kCompact, NORING, false, true>(xpose, state, n, groups, gwords, numWords, ringLds, sLut, sink, hl, upper, nullptr, topRows)
      if (tailPair) {
        if (wide) {
          if (noRing) DGPU_DECODE_TAIL(true, false, true); else DGPU_DECODE_TAIL(true, false, false);
        } else {
          if (noRing) DGPU_DECODE_TAIL(false, false, true); else DGPU_DECODE_TAIL(false, false, false);
        }
      } else {
        if (wide) {
          if (noRing) DGPU_DECODE_TAIL(true, true, true); else DGPU_DECODE_TAIL(true, true, false);
        } else {
          if (noRing) DGPU_DECODE_TAIL(false, true, true); else DGPU_DECODE_TAIL(false, true, false);
        }
      }
#undef DGPU_DECODE_TAIL
    } else {
      decodeBlock<P, FT, false, false, false, kCompact>(xpose, state, n, divUp(maxRows, kGroupRows), gwords, numWords, ringLds, sLut, sink, hl, upper);
    }
  }
}

}  // namespace dgpu
