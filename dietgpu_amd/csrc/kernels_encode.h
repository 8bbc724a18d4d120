// rANS encode for gfx950, single pass, with the float split fused in.
//
// What it computes is the reference's splitFloat (float path) +
// ansEncodeBatchFull/Partial + batchExclusivePrefixSum + ansEncodeCoalesceBatch
// (dietgpu/float/GpuFloatCompress.cuh:280-365, dietgpu/ans/GpuANSEncode.cuh:
// 49-211, 301-672); how it computes it is different:
//
//   * The wire format interleaves 32 rANS states per 4 KiB block.  A wave64
//     therefore encodes TWO blocks at once: lanes 0-31 own block 2w, lanes
//     32-63 own block 2w+1.  One 64-bit ballot per row serves both halves;
//     each half takes its own 32-bit slice for the prefix popcount.
//   * Symbols reach the lanes through a 512-byte LDS ring per block (16 rows; 8 for fp32):
//     16-byte global loads, one ds_write_b128, then a ds_read_u8 per row whose
//     result is turned into the LDS address of the symbol's table entry right
//     away.  Neither that nor the table lookup depends on the rANS state, so
//     both run ahead of the dependent chain.
//   * For the float codec the SOURCE of a chunk is the float words themselves:
//     the lane splits its 16 words with packed byte tricks (v_perm, packed-u16
//     shift + multiply-add, v_alignbit), stores the 16 non-compressed bytes
//     straight into the archive and hands the 16 exponent bytes to the ring.  The exponent plane
//     never exists in HBM (the reference writes and re-reads it).
//   * The row step of a full block is branch-free straight-line code: the
//     emitting lanes store their word (and, for raw bytes and bfloat16, shift
//     their state) under the row's ballot as execution mask -- two s_mov, no
//     branch, no select (docs/HISTORY.md section 4.1: the vector issue slots are the
//     scarce ones in this loop).
//   * Each half-wave emits its u16 words into an LDS stage (worst case for raw
//     bytes; 1024 words + a spill slot in temp memory for floats, see
//     encStageCap).  A workgroup (4 waves = 8 blocks = one "tile") publishes the
//     padded word count of its tile and obtains its archive offset with a
//     decoupled look-back over the preceding tiles of the same batch element,
//     then copies the stage to its final place with 16-byte stores.  There is no
//     per-block scratch buffer in HBM and no coalesce pass.
//   * Tiles are encoded in TILE-MAJOR ticket order (ticket t -> element t % B, tile t / B), either by persistent
//     workgroups (as many as fit on the chip, a static ticket map: 8-block float tiles) or by one workgroup per
//     ticket that the hardware dispatches in index order (raw bytes, float tiles of 2 / 4 blocks), in both forms
//     protected by per-tile claim words, see k_ans_encode.  A tile only ever waits on tiles that are claimed by a
//     running workgroup: the look-back cannot deadlock whatever order the hardware dispatches workgroups in and
//     however many of them are resident.  With a batch of B elements the predecessor started B tickets earlier,
//     i.e. it has usually finished long before: measured with element-major order the look-back wait was 17 % of
//     a tile's lifetime.
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
__host__ __device__ constexpr uint32_t encStageWords(int P) { return 32u * (8u * (uint32_t)P + 1u); }

// Two stage policies (template parameter kSpill of the kernel):
//   * kSpill = false: the LDS stage of a block holds its worst case
//     (encStageWords).  ~50 KiB of LDS per workgroup => 3 workgroups per CU.
//   * kSpill = true: the stage holds kSpillStageWords.  Every kFlushRows rows
//     the wave checks whether the next kFlushRows rows could overflow it (a row
//     emits at most 32 words per block); if so the stage's whole 16-byte
//     vectors are flushed to a per-workgroup spill slot in temp memory and read
//     back at copy-out.  ~25 KiB of LDS => 6 workgroups per CU, which is what
//     keeps the SIMDs issuing row steps while other tiles sit in their
//     look-back / copy-out phases.  Exponent streams (2-4 bits per symbol) never
//     reach the flush threshold of 3 bits/symbol averaged over a block; the
//     spill path is the safety net for incompressible inputs, not the fast path.
constexpr uint32_t kSpillStageWords = 1024;       // bf16 / fp32: the compressed byte is the 8-bit exponent
// fp16's compressed byte is sign + 5 exponent bits + 2 MANTISSA bits: ~2 bits more entropy.  With
// 1024 words most blocks of BASELINE config 4 flushed (encode 107 us, and the spill traffic pushed
// the archive out of the memory-side cache: decode 122 us); 1280 words (5 workgroups per CU): 91 / 104.
constexpr uint32_t kSpillStageWordsFp16 = 1280;
// ... and bf16 / fp32 batches of elements of few tiles take the same 1280 words (k_ans_encode, kWide; capi.hip,
// encoderWideStage): five workgroups per CU instead of six and no block of N(0,1) exponents -- 717 words on average,
// the flush check fires above 768 -- ever flushes: 256 x 512 Ki bf16 encode 90.2 -> 87.5 us, fp32 step - 1.6 %; elements
// of hundreds of tiles lose (16 x 8 Mi 112 -> 116 us: fewer tiles in flight behind the in-order commit) and keep 1024
// (profiles/r06_ab_encoder_five_per_cu_*.txt).
constexpr uint32_t kSpillStageWordsWide = 1280;
constexpr uint32_t kFlushRows = 8;
// words of spill slot per block: the worst case of a block (whole vectors)
__host__ __device__ constexpr uint32_t encSpillSlotWords(int P) { return roundUp(encStageWords(P), 8u) + 8u; }

// (Raw bytes keep the worst-case stage: a 1664-word stage with spill slots -- 4 workgroups per CU -- measured -2 %
// for 43 MiB more temp memory, and nothing once the row stored under the ballot; docs/HISTORY.md section 5, "Config 2".)
__host__ __device__ constexpr uint32_t encStageCap(int P, bool spill, uint32_t ft, bool wide = false) {
  return spill ? (ft == kFloat16 ? kSpillStageWordsFp16 : (wide ? kSpillStageWordsWide : kSpillStageWords)) : encStageWords(P);
}
// Blocks per tile = per workgroup: 8 (256 threads), or 4 (128 threads) for batches whose elements have
// at most 4 blocks -- an 8-block tile would leave half of its waves without a block there.
constexpr uint32_t kBlocksPerSmallTile = 4;
// ... and a single wavefront (64 threads) for batches of elements of at most 2 blocks: half the LDS per workgroup,
// twice the resident tiles
constexpr uint32_t kBlocksPerTinyTile = 2;
// ... and batches of SINGLE-block elements go to k_ans_encode_pair (kernels_pairs.h): two ELEMENTS per wavefront
constexpr uint32_t kBlocksPerSingleTile = 1;
__host__ __device__ constexpr uint32_t encThreads(uint32_t tileBlocks) { return tileBlocks * 32u; }
__host__ __device__ constexpr uint32_t encLdsBytes(int P, bool spill, uint32_t ft, uint32_t tileBlocks, bool wide = false) {
  return 4096u                                       // packed symbol table
      + 128u                                         // tile bookkeeping
      + tileBlocks * encStageCap(P, spill, ft, wide) * 2u  // bitstream stage per half-wave
      + tileBlocks * 512u;                           // symbol ring, 16 rows per half-wave
}

// descriptors a lane reads per look-back step (one round trip covers 64 x this many predecessor tiles)
constexpr uint32_t kLookbackPerLane = 1;
constexpr uint64_t kDescAggregate = 1ull << 62;
constexpr uint64_t kDescInclusive = 2ull << 62;
constexpr uint64_t kDescValueMask = (1ull << 62) - 1;
constexpr uint64_t kDescFailed = 1ull << 40;  // sticky: a tile overran its stage (see k_ans_encode)
// pause between two polls of an unpublished descriptor: 64 cycles, doubling up to 128 x 64 = 3.4 us (see lookBackExclusive)
constexpr uint32_t kLookbackPollPauseMax = 128;

struct EncodeArgs {
  BatchView in;              // raw bytes (FT == 0) or float words (FT != 0); size(b) = symbols = bytes / words
  BatchView out;             // archive base pointers
  const uint4* encTable;     // [B][256] from the normalisation; null for k_ans_encode_pair (builds its own from the pdf)
  uint32_t maxTiles;         // tiles per element the ticket space is laid out for
  uint32_t numInBatch;       // B
  uint32_t numTickets;       // B * maxTiles, or the entries of workMap
  const uint32_t* workMap;   // nullable: [numTickets] element << 16 | tile: the tiles that exist, element by element
  uint64_t* tileDesc;        // [B][maxTiles] (workMap: [numTickets]), zeroed before launch (by the normalisation step)
  uint32_t* claims;          // [maxTiles][B] (workMap: [numTickets]) tile claim words, zeroed before launch
  // second level of the look-back (lookBackTwoLevel), rectangles with maxTiles > 64 only, else null: per element
  // groupsPerElement = ceil(maxTiles / 64) arrival words (one per 128 bytes) and as many descriptors, zeroed before launch
  uint64_t* groupWords;      // [B] x {[groupsPerElement] x kGroupArriveStride arrival words, [groupsPerElement] descriptors}
  uint32_t groupsPerElement;
  uint32_t absentModulo;     // test hook: workgroups with index % absentModulo == 1 start ~0.5 ms late (0 = off)
  uint32_t pollLong;         // look-back: pause at the maximum from the first poll (many tiles of an element in flight at once: long waits)
  uint16_t* spill;           // kSpill kernels only.  Persistent grids (k_ans_encode, 8-block float tiles): [gridDim.x][blocks
                             // per tile][encSpillSlotWords(P)], a workgroup's slots are its own.  Hardware-dispatched grids
                             // (k_ans_encode with 2- / 4-block float tiles, k_ans_encode_pair): [spillPairs][2]
                             // [encSpillSlotWords(P)], a POOL
  uint32_t* spillFlags;      // hardware-dispatched grids: [spillPairs] 0 = free; library-owned, zero at rest (SpillPool)
  uint32_t spillPairs;       // ... >= the wavefronts of the kernel that can be resident at once
  uint32_t* outSize;         // [B] nullable
  uint32_t outCapacity;      // bytes the caller has at out.ptr(b): block data beyond it is NOT stored (outSize still
                             // reports the full size); 0xffffffff = the reference's contract (room for the maximum).
                             // The host guarantees that header, tables and non-compressed planes fit.
  uint32_t useChecksum;      // float header only
  const uint32_t* checksum;  // [B] nullable (float header only)
};

struct TileShared {
  uint32_t tileLo;            // first tile of the element this workgroup encodes in this round (see k_ans_encode)
  uint32_t tileBase;          // exclusive prefix (u16 words) of this tile in the element
  uint32_t words[kBlocksPerTile];
  uint32_t localOff[kBlocksPerTile];
};

typedef __attribute__((address_space(3))) uint16_t LdsU16e;
typedef uint16_t u16x2e __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4e LdsU4e;

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also
// drains the vector-memory counter (s_waitcnt vmcnt(0)); inside the tile loop no
// wave reads global memory another wave of its workgroup wrote, so the four
// barriers of a tile need not wait for outstanding archive stores / prefetches.
__device__ __forceinline__ void ldsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// table entry at an absolute LDS address
__device__ __forceinline__ uint4 ldsTableEntry(uint32_t addr) {
  const u32x4e v = *(const LdsU4e*)(uintptr_t)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------
// Chunk sources.  A chunk is 16 rows of one block = 512 symbols; lane hl of the
// half-wave owns symbols [16 hl, 16 hl + 16) of the chunk when loading.
//   load(c)    : issue the global loads of chunk c (kept in registers)
//   consume(r) : turn them into the 16 symbol bytes for the ring; the float
//                sources also store the non-compressed bytes into the archive
//   wordAt(i) / splitAt(i, w, valid): scalar path for partial blocks / unaligned inputs
//                (the load is unconditional so that a group of them overlaps)
template <uint32_t FT>
struct ChunkSource;

// The lane's slice of kParts x 16 bytes at `base + begin` of a block that holds `nB` bytes, for the tail forms below
// (encodeRows, kTail): a part wholly inside the block is loaded where it lies, whatever its alignment; a part at or
// beyond nB is not loaded.  The one part that can STRADDLE nB is loaded where it lies when the parts are 16-byte aligned
// windows (the window holds the block's last byte, so it cannot leave the page); otherwise the 16 bytes that END at the
// block's last byte are loaded and shifted down -- no byte beyond the element is touched.  That load begins up to 15
// bytes before the part: the caller sends elements below 16 bytes that are not 16-byte aligned to the scalar path
// (encVectorLoadsOk).  Bytes at or beyond nB come back as zero or as whatever the aligned window held; the callers
// mask them.
template <uint32_t FT>
__device__ __forceinline__ bool encVectorLoadsOk(const uint8_t* in, uint32_t size) {
  constexpr uint32_t kWordBytes = FT == 0u ? 1u : (FT == kFloat32 ? 4u : 2u);
  const uint32_t a = (uint32_t)(uintptr_t)in;
  return (a & 15u) == 0 || ((a & (kWordBytes - 1u)) == 0 && (uint64_t)size * kWordBytes >= 16u);
}
template <uint32_t kParts>
__device__ __forceinline__ void loadSliceBounded(const uint8_t* base, uint32_t begin, uint32_t nB, uint4 (&part)[kParts]) {
  // (laundered: everything here that does not depend on the chunk would otherwise be kept in registers across the row
  // loop, which has none to spare)
  uint32_t lowBits = (uint32_t)(uintptr_t)base;
  asm volatile("" : "+v"(lowBits));
  const bool windows = (lowBits & 15u) == 0;  // (begin is a multiple of 16)
  asm volatile("" : "+v"(nB), "+v"(begin));
#pragma unroll
  for (uint32_t k = 0; k < kParts; ++k) {
    const uint32_t pb = begin + 16u * k;
    uint32_t off = pb, sh = 0u;  // sh: bytes to drop at the bottom of what was loaded
    bool ld = windows ? pb < nB : pb + 16u <= nB;
    if (!windows && pb < nB && nB < pb + 16u) {
      off = nB - 16u;
      sh = pb + 16u - nB;  // 1 .. 15
      ld = true;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ld) v = streamLoad<kNtEncLoads>((const uint4*)(base + (int32_t)off));
    if (sh != 0u) {
      if (sh & 8u) v = make_uint4(v.z, v.w, 0u, 0u);
      if (sh & 4u) v = make_uint4(v.y, v.z, v.w, 0u);
      v.x = __builtin_amdgcn_alignbyte(v.y, v.x, sh & 3u);
      v.y = __builtin_amdgcn_alignbyte(v.z, v.y, sh & 3u);
      v.z = __builtin_amdgcn_alignbyte(v.w, v.z, sh & 3u);
      v.w = __builtin_amdgcn_alignbyte(0u, v.w, sh & 3u);
    }
    part[k] = v;
  }
}

template <>
struct ChunkSource<0> {  // raw bytes: the symbols are the input
  static constexpr uint32_t kRows = 16;  // rows per chunk
  struct Raw { uint4 v; };
  const uint8_t* in;  // this half's block
  __device__ __forceinline__ void init(const uint8_t* elemIn, uint8_t*, uint32_t, uint32_t block) {
    in = elemIn + (size_t)block * kBlockSize;
  }
  __device__ __forceinline__ Raw load(uint32_t c, uint32_t hl) const {
    Raw r;
    r.v = streamLoad<kNtEncLoads>(&((const uint4*)in)[c * 32u + hl]);
    return r;
  }
  __device__ __forceinline__ void consume(const Raw& r, uint32_t, uint32_t hl, uint8_t* ring) const { *(uint4*)(ring + hl * 16u) = r.v; }
  // Tail forms (a block of n < 4096 symbols; see encodeRows, kTail, and loadSliceBounded): a 16-byte part of the
  // lane's slice that begins at or beyond symbol n is not loaded; the symbols of a part that straddles n beyond n are
  // never coded.
  __device__ __forceinline__ Raw loadTail(uint32_t c, uint32_t hl, uint32_t n) const {
    uint4 part[1];
    loadSliceBounded<1>(in, c * 512u + hl * 16u, n, part);
    Raw r;
    r.v = part[0];
    return r;
  }
  __device__ __forceinline__ void consumeTail(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring, uint32_t) const { consume(r, c, hl, ring); }
  // split without the ring write (fused kernel: the symbol bytes stay in registers for a while)
  static constexpr uint32_t kCompRegs = 4;
  __device__ __forceinline__ void splitStore(const Raw& r, uint32_t, uint32_t, uint32_t (&comp)[kCompRegs]) const {
    comp[0] = r.v.x; comp[1] = r.v.y; comp[2] = r.v.z; comp[3] = r.v.w;
  }
  __device__ __forceinline__ uint32_t wordAt(uint32_t i) const { return in[i]; }
  __device__ __forceinline__ uint32_t splitAt(uint32_t, uint32_t w, bool) const { return w; }
};

// v_perm_b32 selectors: {lo.b0, lo.b2, hi.b0, hi.b2} and {lo.b1, lo.b3, hi.b1, hi.b3}
__device__ __forceinline__ uint32_t packBytes02(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }
__device__ __forceinline__ uint32_t packBytes13(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_perm(hi, lo, 0x07050301u); }
// (mask & a) | (~mask & b)
__device__ __forceinline__ uint32_t bitSelect(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

template <uint32_t FT>  // kFloat16 / kBFloat16: 2-byte words, 1 comp byte + 1 non-comp byte
struct ChunkSource16 {
  static constexpr uint32_t kRows = 16;
  struct Raw { uint4 a, b; };
  const uint16_t* in;  // this half's block (4096 words)
  uint8_t* nc;         // this half's block of the non-comp plane
  __device__ __forceinline__ void init(const uint8_t* elemIn, uint8_t* archive, uint32_t, uint32_t block) {
    in = (const uint16_t*)elemIn + (size_t)block * kBlockSize;
    nc = archive + 16u + (size_t)block * kBlockSize;
  }
  __device__ __forceinline__ Raw load(uint32_t c, uint32_t hl) const {
    const uint4* p = (const uint4*)(in + c * 512u + hl * 16u);
    Raw r;
    r.a = streamLoad<kNtEncLoads>(&p[0]);
    r.b = streamLoad<kNtEncLoads>(&p[1]);
    return r;
  }
  static constexpr uint32_t kCompRegs = 4;
  __device__ __forceinline__ void consume(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring) const {
    uint32_t comp[4];
    splitStore(r, c, hl, comp);
    *(uint4*)(ring + hl * 16u) = make_uint4(comp[0], comp[1], comp[2], comp[3]);
  }
  // FloatTypeInfo<FT>::split (GpuFloatUtils.cuh:111-115, 141-147) on packed pairs: the non-compressed
  // bytes go to the archive, the compressed (exponent) bytes of the lane's 16 words are returned
  // Tail forms: see ChunkSource<0>.  Words at or beyond n inside a loaded part are set to zero, so that their
  // non-compressed bytes -- which land in the plane's zero padding up to the next 16-byte boundary -- are zero; a lane
  // whose slice lies wholly beyond n stores nothing.
  __device__ __forceinline__ Raw loadTail(uint32_t c, uint32_t hl, uint32_t n) const {
    const uint32_t first = c * 512u + hl * 16u;
    uint4 part[2];
    loadSliceBounded<2>((const uint8_t*)in, first * 2u, n * 2u, part);
    Raw r;
    r.a = part[0];
    r.b = part[1];
    return r;
  }
  __device__ __forceinline__ void consumeTail(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring, uint32_t n) const {
    const uint32_t first = c * 512u + hl * 16u;
    const uint32_t valid = n > first ? (n - first < 16u ? n - first : 16u) : 0u;  // words of the lane's slice that exist
    auto keep = [&](uint32_t x, uint32_t j) -> uint32_t {  // dword j holds words 2 j, 2 j + 1 of the slice
      return valid >= 2u * j + 2u ? x : (valid == 2u * j + 1u ? (x & 0xffffu) : 0u);
    };
    Raw m;
    m.a = make_uint4(keep(r.a.x, 0), keep(r.a.y, 1), keep(r.a.z, 2), keep(r.a.w, 3));
    m.b = make_uint4(keep(r.b.x, 4), keep(r.b.y, 5), keep(r.b.z, 6), keep(r.b.w, 7));
    uint32_t comp[4];
    splitStore(m, c, hl, comp, valid != 0u);
    *(uint4*)(ring + hl * 16u) = make_uint4(comp[0], comp[1], comp[2], comp[3]);
  }
  __device__ __forceinline__ void splitStore(const Raw& r, uint32_t c, uint32_t hl, uint32_t (&comp)[kCompRegs], bool store = true) const {
    const uint32_t x[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    uint32_t rest[4];
    if (FT == kFloat16) {
      // comp = w >> 8 (bytes 1, 3 of each dword), nonComp = w & 0xff (bytes 0, 2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        comp[j] = packBytes13(x[2 * j + 1], x[2 * j]);
        rest[j] = packBytes02(x[2 * j + 1], x[2 * j]);
      }
    } else {
      // bf16: comp = bits 14..7; nonComp = mantissa7 << 1 | sign, i.e. each
      // 16-bit half rotated left by one, low byte
      uint32_t t[8], q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        t[j] = x[j] >> 7;                                         // comp in bytes 0 and 2
        // each half rotated left by one = w * 2 + (w >> 15) on packed u16 (v_pk_lshrrev_b16 + v_pk_mad_u16)
        // (the compiler expands the multiply into a shift and an add; the asm keeps it to two ops)
        const u16x2e w = __builtin_bit_cast(u16x2e, x[j]);
        const uint32_t signs = __builtin_bit_cast(uint32_t, (u16x2e)(w >> 15));
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(q[j]) : "v"(x[j]), "s"(0x00020002u), "v"(signs));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        comp[j] = packBytes02(t[2 * j + 1], t[2 * j]);
        rest[j] = packBytes02(q[2 * j + 1], q[2 * j]);
      }
    }
    if (store) streamStore<kNtEncStores>(&((uint4*)(nc + c * 512u))[hl], make_uint4(rest[0], rest[1], rest[2], rest[3]));
  }
  __device__ __forceinline__ uint32_t wordAt(uint32_t i) const { return in[i]; }
  __device__ __forceinline__ uint32_t splitAt(uint32_t i, uint32_t w, bool valid) const {
    uint32_t c, r;
    if (FT == kFloat16) {
      c = w >> 8;
      r = w & 0xffu;
    } else {
      c = (w >> 7) & 0xffu;
      r = ((w << 1) & 0xfeu) | (w >> 15);
    }
    if (valid) nc[i] = (uint8_t)r;
    return c;
  }
};
template <>
struct ChunkSource<kFloat16> : ChunkSource16<kFloat16> {};
template <>
struct ChunkSource<kBFloat16> : ChunkSource16<kBFloat16> {};

template <>
struct ChunkSource<kFloat32> {  // 4-byte words: comp byte + 24 non-comp bits (u16 plane, then u8 plane)
  // 8-row chunks (256 symbols, 8 words = 32 bytes per lane): a 16-row chunk in
  // flight is 16 registers, which the row loop cannot afford (it spilled)
  static constexpr uint32_t kRows = 8;
  struct Raw { uint4 v[2]; };
  const uint32_t* in;
  uint16_t* nc2;
  uint8_t* nc1;
  __device__ __forceinline__ void init(const uint8_t* elemIn, uint8_t* archive, uint32_t size, uint32_t block) {
    in = (const uint32_t*)elemIn + (size_t)block * kBlockSize;
    nc2 = (uint16_t*)(archive + 16u) + (size_t)block * kBlockSize;
    nc1 = archive + 16u + 2u * (size_t)roundUp(size, 8u) + (size_t)block * kBlockSize;
  }
  __device__ __forceinline__ Raw load(uint32_t c, uint32_t hl) const {
    const uint4* p = (const uint4*)(in + c * 256u + hl * 8u);
    Raw r;
    r.v[0] = streamLoad<kNtEncLoads>(&p[0]);
    r.v[1] = streamLoad<kNtEncLoads>(&p[1]);
    return r;
  }
  static constexpr uint32_t kCompRegs = 2;
  __device__ __forceinline__ void consume(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring) const {
    uint32_t comp[2];
    splitStore(r, c, hl, comp);
    *(uint2*)(ring + hl * 8u) = make_uint2(comp[0], comp[1]);
  }
  // FloatTypeInfo<kFloat32>::split (GpuFloatUtils.cuh:181-185): v = rotl(w, 1)
  // Tail forms: see ChunkSource16 (8 words per lane and chunk here, one per dword).
  __device__ __forceinline__ Raw loadTail(uint32_t c, uint32_t hl, uint32_t n) const {
    const uint32_t first = c * 256u + hl * 8u;
    Raw r;
    loadSliceBounded<2>((const uint8_t*)in, first * 4u, n * 4u, r.v);
    return r;
  }
  __device__ __forceinline__ void consumeTail(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring, uint32_t n) const {
    const uint32_t first = c * 256u + hl * 8u;
    const uint32_t valid = n > first ? (n - first < 8u ? n - first : 8u) : 0u;
    auto keep = [&](uint32_t x, uint32_t j) -> uint32_t { return valid > j ? x : 0u; };
    Raw m;
    m.v[0] = make_uint4(keep(r.v[0].x, 0), keep(r.v[0].y, 1), keep(r.v[0].z, 2), keep(r.v[0].w, 3));
    m.v[1] = make_uint4(keep(r.v[1].x, 4), keep(r.v[1].y, 5), keep(r.v[1].z, 6), keep(r.v[1].w, 7));
    uint32_t comp[2];
    splitStore(m, c, hl, comp, valid != 0u);
    *(uint2*)(ring + hl * 8u) = make_uint2(comp[0], comp[1]);
  }
  __device__ __forceinline__ void splitStore(const Raw& r, uint32_t c, uint32_t hl, uint32_t (&comp)[kCompRegs], bool store = true) const {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      v[4 * j + 0] = __builtin_amdgcn_alignbit(r.v[j].x, r.v[j].x, 31);
      v[4 * j + 1] = __builtin_amdgcn_alignbit(r.v[j].y, r.v[j].y, 31);
      v[4 * j + 2] = __builtin_amdgcn_alignbit(r.v[j].z, r.v[j].z, 31);
      v[4 * j + 3] = __builtin_amdgcn_alignbit(r.v[j].w, r.v[j].w, 31);
    }
    uint32_t hi[2], lo[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // bytes 2 (high non-comp byte) and 3 (comp) of four words -> one dword each
      const uint32_t a = __builtin_amdgcn_perm(v[4 * j + 1], v[4 * j + 0], 0x07030602u);  // {v0.b2, v1.b2, v0.b3, v1.b3}
      const uint32_t b = __builtin_amdgcn_perm(v[4 * j + 3], v[4 * j + 2], 0x07030602u);
      hi[j] = __builtin_amdgcn_perm(b, a, 0x05040100u);    // {a.b0, a.b1, b.b0, b.b1}
      comp[j] = __builtin_amdgcn_perm(b, a, 0x07060302u);  // {a.b2, a.b3, b.b2, b.b3}
      lo[2 * j + 0] = __builtin_amdgcn_perm(v[4 * j + 1], v[4 * j + 0], 0x05040100u);  // low 16 bits of two words
      lo[2 * j + 1] = __builtin_amdgcn_perm(v[4 * j + 3], v[4 * j + 2], 0x05040100u);
    }
    if (store) {
      streamStore<kNtEncStores>((uint4*)(nc2 + c * 256u + hl * 8u), make_uint4(lo[0], lo[1], lo[2], lo[3]));
      *(uint2*)(nc1 + c * 256u + hl * 8u) = make_uint2(hi[0], hi[1]);
    }
  }
  __device__ __forceinline__ uint32_t wordAt(uint32_t i) const { return in[i]; }
  __device__ __forceinline__ uint32_t splitAt(uint32_t i, uint32_t w, bool valid) const {
    const uint32_t v = (w << 1) | (w >> 31);
    if (valid) {
      nc2[i] = (uint16_t)(v & 0xffffu);
      nc1[i] = (uint8_t)((v >> 16) & 0xffu);
    }
    return v >> 24;
  }
};

// ---------------------------------------------------------------------------
// The emitting lanes of a full-block row store their word under the row's ballot as execution mask -- two scalar
// instructions instead of the v_cndmask that would park the idle lanes' store on a scratch slot -- and, for raw
// bytes and bfloat16, shift their state down in the same window (two v_cndmask fewer per row).  Measured on MI355X
// (profiles/r03_ab_encoder_exec_write.txt): raw bytes 163.2 -> 152.7 us on 256 x 1 MiB, bfloat16 87.2 / 82.8 ->
// 81.2 us with the shift inside; float16 83.2 -> 82.2 us with the store alone (86.5 us with the shift), float32
// indifferent.  Variants that measured SLOWER and are not in this file (tools/experiments/README.md): emit
// positions in SGPRs (-2 VALU, +7 SALU: +8 %), a packed 8-byte table entry (half the LDS bytes, +3 VALU: +8 %),
// table entries 4 rows ahead instead of 2 (+8 %), a hand-scheduled SDWA select (+5 %).
template <uint32_t FT>
constexpr bool encShiftUnderBallot() { return FT == 0u || FT == kBFloat16; }
// The incoming execution mask is saved and restored (s_and_saveexec), so a caller with inactive lanes keeps them
// inactive; every caller in this library runs the full-block step with all 64 lanes active.
__device__ __forceinline__ void stageWriteUnder(uint64_t vote, uint32_t addr, uint32_t state) {
  uint64_t saved;
  asm volatile("s_and_saveexec_b64 %[sv], %[v]\n\tds_write_b16 %[a], %[s]\n\ts_mov_b64 exec, %[sv]"
               : [sv] "=&s"(saved) : [a] "v"(addr), [s] "v"(state), [v] "s"(vote) : "memory", "scc");
}
__device__ __forceinline__ void stageWriteShiftUnder(uint64_t vote, uint32_t addr, uint32_t& state) {
  uint64_t saved;
  asm volatile("s_and_saveexec_b64 %[sv], %[v]\n\tds_write_b16 %[a], %[s]\n\tv_lshrrev_b32 %[s], 16, %[s]\n\ts_mov_b64 exec, %[sv]"
               : [s] "+v"(state), [sv] "=&s"(saved) : [a] "v"(addr), [v] "s"(vote) : "memory", "scc");
}

// Spill slots of a hardware-dispatched grid (k_ans_encode_pair and the small float tiles of k_ans_encode: one workgroup
// per pair of elements / per tile, so blockIdx.x is no bound on what is resident).  A wavefront that has to flush takes a PAIR of slots (one per half) out of a pool with
// one flag word per pair and gives it back after its copy-out.  The pool has at least as many pairs as wavefronts of
// the kernel can be resident and a wavefront holds at most one, so the probe terminates; the flags are zero at rest.
// A slot changes hands between wavefronts on DIFFERENT XCDs, whose L2s are not coherent with each other: everything
// written to or read from a pooled slot goes through agent-scope accesses (write-through stores, L2-bypassing loads --
// the hand-off discipline of the look-back descriptors).  Rare path: incompressible data only.
struct SpillPool {
  uint16_t* base;
  uint32_t* flags;
  uint32_t pairs;
  uint32_t pair;  // kNoSpillPair until this wavefront has taken one (for the tile / pair of elements it is encoding)
};
constexpr uint32_t kNoSpillPair = 0xffffffffu;
__device__ __forceinline__ uint32_t spillAcquire(const SpillPool& sp, uint32_t seed) {  // whole wavefront
  uint32_t idx = 0;
  if ((threadIdx.x & 63u) == 0u) {
    uint32_t i = seed % sp.pairs;
    for (;;) {
      uint32_t expected = 0;
      if (__hip_atomic_compare_exchange_strong(sp.flags + i, &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      i = i + 1u == sp.pairs ? 0u : i + 1u;
    }
    idx = i;
  }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
}
__device__ __forceinline__ void spillRelease(SpillPool& sp) {  // whole wavefront, after its last read of the slots
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(sp.flags + sp.pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sp.pair = kNoSpillPair;
}
__device__ __forceinline__ void coherentStore16(uint4* p, const uint4& v) {
  __hip_atomic_store((uint64_t*)p, (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((uint64_t*)p + 1, (uint64_t)v.z | ((uint64_t)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 coherentLoad16(const uint4* p) {
  const uint64_t lo = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t hi = __hip_atomic_load((const uint64_t*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// Encodes the rows of one block per half-wave.  Returns the words left in the LDS stage; `spilledOut` = words
// flushed to the spill slot (kSpill), `stateOut` = the lane's final state, `overrunOut` = the block emitted more
// words than stage (+ spill slot) can hold.  That cannot happen with a table made from this data's histogram (the
// stage of the non-spilling variant holds the proven worst case, encStageWords); with a CALLER-SUPPLIED histogram
// that does not cover the data it can, and the element is then reported as failed (k_ans_encode): stores beyond
// the stage land in the neighbouring stages / rings of the same workgroup -- the same, already failed, element --
// or beyond the workgroup's LDS allocation, where the hardware drops them.  kGuard (k_ans_encode_pair, whose two
// stages belong to DIFFERENT elements): every kFlushRows rows a half whose word count exceeds what ANY covering table
// can have produced by then (encGuardLimit) starts over at the stage's first word; with kEncGuardSlackWords of room
// behind the stage nothing is ever stored outside a half's own stage (three VALU per eight rows).
// After r rows a lane has emitted at most (r P + 16 + 0.09 r) / 16 words (information conservation: <= P bits per
// symbol, 16 bits of state head-room, < 0.09 bit of rounding slop per step), a block at most 2 r P + 55.
__host__ __device__ constexpr uint32_t encGuardLimit(int P, uint32_t row) { return 2u * row * (uint32_t)P + 96u; }
// ... so between two checks (<= 32 words per row) a guarded stage holds at most encGuardLimit(P, 120) + 256 words
constexpr uint32_t kEncGuardSlackWords = 192;
static_assert(encGuardLimit(9, 120) + 256u <= encStageWords(9) + kEncGuardSlackWords &&
              encGuardLimit(10, 120) + 256u <= encStageWords(10) + kEncGuardSlackWords &&
              encGuardLimit(11, 120) + 256u <= encStageWords(11) + kEncGuardSlackWords, "");
// kPool (with kSpill): the slot comes from `pool` at the first flush instead of being `spill` (see SpillPool).
// kTail (with kFull): the chunked path for blocks that are NOT full -- the last block of an element, single-block
// elements of any size -- on word-aligned inputs (encVectorLoadsOk): `n` symbols per half (0: idle half), `maxRows` rows, chunk loads
// and non-compressed stores bounded by n (ChunkSource::loadTail / consumeTail), every row step predicated by
// `symbol index < n`.  Three VALU more per row than the full-block step, against the scalar path's one memory round
// trip per eight rows (bf16, 256 x 530 000: encode 120 -> 112 us, 32768 x 4000: 237 -> 138 us;
// profiles/r05_ab_partial_blocks.txt).
// kWide (with kSpill): the stage holds encStageCap(P, true, FT, true) words.
template <int P, uint32_t FT, bool kFull, bool kSpill, bool kGuard = false, bool kPool = false, bool kTail = false, bool kWide = false>
__device__ __forceinline__ uint32_t encodeRows(
    const ChunkSource<FT>& src,
    uint32_t n,                           // symbols in this half's block (0 = idle half)
    uint32_t maxRows,                     // wave-uniform row count
    uint32_t tableLds,                    // LDS address of the element's table (256 x 16 bytes)
    uint32_t stageBase,                   // LDS address of this half's word stage
    uint8_t* __restrict__ ring,           // LDS, this half's 512-byte symbol ring
    uint32_t hl,
    bool upper,
    uint16_t* spill,                      // this half's spill slot (kSpill only; kPool: located at the first flush)
    uint32_t& spilledOut,                 // words flushed to it (multiple of 8)
    uint32_t& stateOut,
    bool& overrunOut,
    SpillPool* pool = nullptr) {
  static_assert(!kPool || kSpill, "");
  static_assert(!kTail || kFull, "kTail is a mode of the chunked path");
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  uint32_t state = kStartState;
  uint32_t outOff = 0;
  uint32_t spilled = 0;
  bool overrun = false;

  // Called every kFlushRows rows: make room for the next kFlushRows rows.
  auto makeRoom = [&](uint32_t row) {
    if (!kSpill && kGuard && outOff > encGuardLimit(P, row)) {
      outOff = 0;
      overrun = true;
    }
    if (!kSpill) return;
    const uint32_t o0 = __builtin_amdgcn_readlane(outOff, 0);
    const uint32_t o1 = __builtin_amdgcn_readlane(outOff, 32);
    if ((o0 > o1 ? o0 : o1) + kFlushRows * 32u <= encStageCap(P, true, FT, kWide)) return;  // wave-uniform
    // whole 16-byte vectors go to the spill slot, the (< 8 word) rest moves to the front
    uint32_t nvec = outOff >> 3;
    if (spilled + nvec * 8u > encSpillSlotWords(P)) {  // (only with a table that does not cover the data)
      nvec = 0;
      overrun = true;
    }
    if constexpr (kPool) {
      if (pool->pair == kNoSpillPair) pool->pair = spillAcquire(*pool, blockIdx.x);  // wave-uniform
      spill = pool->base + ((size_t)pool->pair * 2u + (upper ? 1u : 0u)) * encSpillSlotWords(P);
    }
    uint4* dst = (uint4*)(spill + spilled);
    uint32_t hlf = hl;  // (kPool: laundered, so that the 64-bit slot offsets of this rare path are computed here)
    if constexpr (kPool) asm volatile("" : "+v"(hlf));
    for (uint32_t i = hlf; i < nvec; i += 32u) {
      const u32x4e v = *(const LdsU4e*)(uintptr_t)(stageBase + 16u * i);
      if constexpr (kPool) coherentStore16(&dst[i], make_uint4(v.x, v.y, v.z, v.w));
      else dst[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
    const uint32_t rem = outOff & 7u;
    uint16_t t = 0;
    if (hl < rem) t = *(const LdsU16e*)(uintptr_t)(stageBase + 2u * (nvec * 8u + hl));
    if (hl < rem) *(LdsU16e*)(uintptr_t)(stageBase + 2u * hl) = t;
    spilled += nvec * 8u;
    outOff = rem;
  };

  // Generic step (partial blocks): predicated, emission under a branch.
  auto step = [&](const uint4 e, bool valid) {
    const bool write = valid && (state >= e.x);
    const uint64_t vote = __ballot(write);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    if (write) {
      // emitters of a row write in ascending lane order (ds_write_b16 keeps the low half)
      const uint32_t idx = outOff + __popc(vh & laneMaskLt);
      *(LdsU16e*)(uintptr_t)(stageBase + 2u * idx) = (uint16_t)state;
    }
    state = write ? (state >> kEncodedBits) : state;
    // state = ((state / pdf) << P) + state % pdf + cdf
    //       = state + cdf + (state / pdf) * (2^P - pdf)      (table: normalizeElement)
    const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
    const uint32_t next = __umul24(div, e.w) + state + e.z;
    state = valid ? next : state;
    outOff += __popc(vh);
  };

  // Full-block step: straight-line code, no branch.
  auto stepFull = [&](const uint4 e) {
    const bool write = state >= e.x;
    const uint64_t vote = __ballot(write);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    const uint32_t idx = outOff + __popc(vh & laneMaskLt);
    if (encShiftUnderBallot<FT>()) {
      stageWriteShiftUnder(vote, stageBase + 2u * idx, state);
    } else {
      stageWriteUnder(vote, stageBase + 2u * idx, state);
      state = write ? (state >> kEncodedBits) : state;
    }
    const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
    state = __umul24(div, e.w) + state + e.z;
    outOff += __popc(vh);
  };

  if (kFull) {
    // chunks of 16 rows (8 for fp32); chunk c+1 is in flight in registers while chunk c is
    // consumed from the LDS ring (same wave writes and reads it: LDS ops of one
    // wave execute in order, no barrier needed).
    // Table entries are fetched kAhead rows ahead of the row being encoded, symbol bytes kSymAhead rows ahead
    // (measured: 2 beats 4 entries -- eight registers fewer matter more than the extra LDS latency cover at 6
    // waves per SIMD; reading all 16 symbols of the chunk up front costs 16 live registers).  The window holds the
    // raw symbol VALUES; the table address is formed where the entry is fetched, two rows after the symbol was
    // requested, so no row waits for the LDS round trip of the byte it has just asked for.
    constexpr int kAhead = 2;
    constexpr int kSymAhead = 4;
    constexpr uint32_t kChunkRows = ChunkSource<FT>::kRows;
    static_assert(kSymAhead > kAhead && kSymAhead <= (int)kChunkRows, "a symbol slot is reused only after its table load was issued");
    const uint32_t numChunks = kTail ? divUp(maxRows, kChunkRows) : kRowsPerBlock / kChunkRows;  // (uniform)
    typename ChunkSource<FT>::Raw cur = kTail ? src.loadTail(0, hl, n) : src.load(0, hl);
#pragma unroll 1
    for (uint32_t c = 0; c < numChunks; ++c) {
      if (kTail) src.consumeTail(cur, c, hl, ring, n);
      else src.consume(cur, c, hl, ring);
      if (c + 1 < numChunks) cur = kTail ? src.loadTail(c + 1, hl, n) : src.load(c + 1, hl);
      auto symAt = [&](int r) -> uint32_t { return (uint32_t)ring[r * 32 + hl]; };
      auto fetchEntry = [&](uint32_t sym) -> uint4 { return ldsTableEntry(tableLds + (sym << 4)); };
      uint32_t sym[kSymAhead];
#pragma unroll
      for (int r = 0; r < kSymAhead; ++r) sym[r] = symAt(r);
      uint4 e[kAhead];
#pragma unroll
      for (int r = 0; r < kAhead; ++r) e[r] = fetchEntry(sym[r]);
#pragma unroll
      for (int r = 0; r < (int)kChunkRows; ++r) {
        const uint32_t row = c * kChunkRows + (uint32_t)r;
        if (kTail && row >= maxRows) break;  // uniform (rows fetched ahead of the end read ring bytes nobody uses)
        if (r % kFlushRows == 0) makeRoom(row);
        const uint4 cur_e = e[r % kAhead];
        if (r + kAhead < (int)kChunkRows) e[r % kAhead] = fetchEntry(sym[(r + kAhead) % kSymAhead]);
        if (r + kSymAhead < (int)kChunkRows) sym[r % kSymAhead] = symAt(r + kSymAhead);
        if (kTail) step(cur_e, row * 32u + hl < n);
        else stepFull(cur_e);
      }
    }
  } else {
    // Partial blocks, unaligned inputs, a wave with a single block: rows in groups
    // of kFlushRows, the symbol loads (and, for floats, the non-compressed stores)
    // of a whole group issued before its first step, so a group costs one memory
    // round trip instead of one per row.
#pragma unroll 1
    for (uint32_t row0 = 0; row0 < maxRows; row0 += kFlushRows) {
      makeRoom(row0);
      uint32_t word[kFlushRows];
#pragma unroll
      for (uint32_t j = 0; j < kFlushRows; ++j) {
        const uint32_t i = (row0 + j) * 32u + hl;
        word[j] = src.wordAt(i < n ? i : 0u);  // unconditional (index 0 is always readable, see the caller)
      }
      // table entries two rows ahead of the dependent chain, as in the full path
      uint32_t taddr[kFlushRows];
#pragma unroll
      for (uint32_t j = 0; j < kFlushRows; ++j) {
        const uint32_t i = (row0 + j) * 32u + hl;
        taddr[j] = tableLds + ((src.splitAt(i, word[j], i < n) & 0xffu) << 4);
      }
      uint4 ent[2] = {ldsTableEntry(taddr[0]), ldsTableEntry(taddr[1])};
#pragma unroll
      for (uint32_t j = 0; j < kFlushRows; ++j) {
        const uint32_t i = (row0 + j) * 32u + hl;
        const uint4 cur_e = ent[j % 2u];
        if (j + 2u < kFlushRows) ent[j % 2u] = ldsTableEntry(taddr[j + 2u]);
        if (row0 + j < maxRows) step(cur_e, i < n);  // uniform condition
      }
    }
  }
  // The copy-out reads the slot back through other lanes of this wave, after two
  // workgroup barriers: workgroup-scope ordering is all that is needed (the
  // waves of a workgroup share their CU's L1), no L2 write-back.
  if (!kSpill && outOff > encStageCap(P, false, FT)) {
    outOff = encStageCap(P, false, FT);
    overrun = true;
  }
  spilledOut = spilled;
  stateOut = state;
  overrunOut = overrun;
  return outOff;
}

// Decoupled look-back of a tile's first wavefront over the descriptors of the preceding tiles of its element: sums their
// aggregates down to the nearest inclusive prefix and returns the tile's exclusive prefix (u16 words); `failed` picks up
// the sticky failure flag of every descriptor counted.  64 * kLookbackPerLane predecessors per round trip (lane l holds
// the kLookbackPerLane nearest ones beyond those of lanes < l).  A descriptor that has not been published is polled with
// LONG pauses between the polls (doubling from 30 ns to 3.4 us: a short wait stays short): the descriptors of the ~1500
// tiles in flight share a few dozen cache lines, every poll is an L2-bypassing load of such a line and every publication a
// write-through store to one, and with short pauses the polls of the waiting tiles are what the publications queue
// behind (measured on MI355X, bf16, fixed pauses,
// profiles/r06_ab_lookback_backoff_*.txt: 1 x 128 Mi encode 115.1 us with s_sleep 1, 113.3 / 110.3 / 106.0 with 8 / 32 /
// 127; 16 x 8 Mi 122.0 -> 113.4; and reading 256 or 512 descriptors per round trip instead of 64 -- 4 or 8 times the
// lines per poll -- costs 25-40 us, profiles/r06_ab_lookback_window_*.txt).
// `longChains` (EncodeArgs::pollLong: more than a dozen tiles of an element in flight at once -- few large elements,
// long waits): the pause starts at its maximum (1 x 128 Mi encode 112.5 -> 105.8 us against the doubling pause, 16 x 8 Mi
// 119.9 -> 113.1, 64 x 2 Mi 93.2 -> 91.6).  Batches of many elements -- a handful of an element's tiles in flight, short
// waits -- keep the doubling pause: 256 x 512 Ki bf16 is 1 % faster with it, and 256 x 1 MiB Zipf bytes lose 4 us of 144 to
// the long one (profiles/r06_ab_poll_pause_*.txt, r06_ab_round6_vs_round5_u8.txt).
// `endedAtStart` (nullable): set when the walk went all the way down to index 0 without meeting a real inclusive
// prefix -- what the caller then holds is the sum of `desc[0 .. tile)` (the second level of lookBackTwoLevel needs to know).
// (kTrackStart = false compiles to the plain walk: the raw-byte encoder, whose row loop the compiler schedules 3 % worse
// with the tracking in the same kernel, never asks -- profiles/r06_ab_raw_encoder_regression_u8.txt)
template <bool kTrackStart = false>
__device__ __forceinline__ uint32_t lookBackExclusive(const uint64_t* desc, uint32_t tile, uint32_t lane, bool& failed, bool longChains,
                                                      bool* endedAtStart = nullptr) {
  uint32_t exclusive = 0;
  int base = (int)tile - 1;
  bool virtualEnd = true;  // (tile 0: nothing to walk)
  while (base >= 0) {
    uint32_t sum = 0;  // aggregates of this lane's descriptors down to (and including) its first inclusive one
    bool sawIncl = false, sawFailed = false, inclIsVirtual = false;
    uint64_t d[kLookbackPerLane];
#pragma unroll
    for (int j = 0; j < (int)kLookbackPerLane; ++j) {  // all of the lane's loads in flight at once
      const int idx = base - (int)(lane * kLookbackPerLane) - j;
      d[j] = kDescInclusive;  // virtual tile -1: inclusive prefix 0
      if (idx >= 0) d[j] = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int j = 0; j < (int)kLookbackPerLane; ++j) {
      if (sawIncl) continue;  // (beyond the lane's first inclusive prefix nothing counts)
      const int idx = base - (int)(lane * kLookbackPerLane) - j;
      uint32_t pause = longChains ? kLookbackPollPauseMax : 1u;  // in units of s_sleep(1) = 64 cycles: 1, 2, 4, ... kLookbackPollPauseMax
      while ((d[j] >> 62) == 0) {
        for (uint32_t q = 0; q < pause; ++q) __builtin_amdgcn_s_sleep(1);
        pause = pause * 2u < kLookbackPollPauseMax ? pause * 2u : kLookbackPollPauseMax;
        d[j] = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sum += (uint32_t)d[j];
      sawFailed = sawFailed || (d[j] & kDescFailed) != 0ull;
      sawIncl = (d[j] >> 62) == 2;
      if (kTrackStart) inclIsVirtual = idx < 0;
    }
    const uint64_t inclMask = __ballot(sawIncl);
    const int firstIncl = inclMask ? (__ffsll((unsigned long long)inclMask) - 1) : 64;
    const bool counted = (int)lane <= firstIncl;
    exclusive += waveReduceSum(counted ? sum : 0u);
    failed = failed || __ballot(counted && sawFailed) != 0ull;
    if (firstIncl < 64) {
      if (kTrackStart) virtualEnd = ((__ballot(sawIncl && inclIsVirtual) >> firstIncl) & 1ull) != 0ull;
      break;
    }
    base -= 64 * (int)kLookbackPerLane;
  }
  if (kTrackStart && endedAtStart) *endedAtStart = virtualEnd;
  return exclusive;
}

// Two levels for elements of more than 64 tiles (EncodeArgs::groupWords).  The tiles of an element come in GROUPS of 64.
// A tile adds its aggregate to its group's arrival word (one 64-bit atomic: sum in the low half, arrivals and failures
// above it); the tile that completes the group publishes the group's aggregate descriptor, and the group's last tile
// later upgrades it to the inclusive prefix.  A tile then sums at most 63 tile descriptors of its OWN group and -- if no
// inclusive prefix lay among them -- the group descriptors below its group down to the nearest inclusive one: two
// round trips instead of up to tiles / 64 (a 16 Mi-word tensor's 512 tiles finish together in the one round of its
// call and the last of them walked eight steps back; the reference's own scan is two-level for the same reason,
// BatchPrefixSum.cuh:69-110).
constexpr uint32_t kLookbackGroup = 64;
constexpr uint32_t kGroupArriveStride = 16;  // u64 words between two groups' arrival words (a 128-byte line each)
__device__ __forceinline__ uint32_t lookBackTwoLevel(const uint64_t* desc, uint64_t* groupArrive, uint64_t* groupDesc, uint32_t tile,
                                                     uint32_t numTiles, uint32_t aggregate, uint32_t lane, bool& failed, bool longChains) {
  const uint32_t g = tile / kLookbackGroup, l = tile % kLookbackGroup;
  const uint32_t groupTiles = (g + 1u) * kLookbackGroup <= numTiles ? kLookbackGroup : numTiles - g * kLookbackGroup;
  if (lane == 0) {
    const uint64_t mine = (1ull << 32) | (failed ? (1ull << 48) : 0ull) | (uint64_t)aggregate;
    const uint64_t prev = __hip_atomic_fetch_add(groupArrive + (size_t)g * kGroupArriveStride, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t now = prev + mine;
    if (((now >> 32) & 0xffffull) == groupTiles) {
      // the group is complete.  Its last tile may have published the inclusive prefix already (it only needs the tiles
      // BEFORE it): never put an aggregate over it
      uint64_t expected = 0;
      (void)__hip_atomic_compare_exchange_strong(groupDesc + g, &expected, kDescAggregate | ((now >> 48) ? kDescFailed : 0ull) | (now & 0xffffffffull),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  bool atGroupStart = false;
  uint32_t exclusive = lookBackExclusive<true>(desc + (size_t)g * kLookbackGroup, l, lane, failed, longChains, &atGroupStart);
  if (atGroupStart && g > 0u) exclusive += lookBackExclusive(groupDesc, g, lane, failed, longChains);
  return exclusive;
}


// Persistent workgroups, STATIC tile map with claim words.  Workgroup w owns the
// tickets w, w + G, w + 2G, ... (G = gridDim.x; ticket t -> element t % B, tile
// t / B) and walks them in order: no ticket atomic on the critical path and a
// perfectly regular round structure (a dynamic ticket counter measured 111 us, eight
// counters 105 us, this 98 us at the time; docs/HISTORY.md section 4.1).  A plain
// static map would hang whenever part of the grid is not resident (another
// kernel holding CUs): running workgroups would spin in the look-back on tiles
// whose owner never starts.  Hence one claim word per tile:
//   * at its start a workgroup claims ALL its tiles (CAS 0 -> w + 1, fire and
//     forget), so every tile of a running workgroup is claimed from then on;
//   * before encoding a tile it makes sure the element's previous tile is
//     claimed by somebody; an unclaimed one (its owner is not running) it
//     claims itself and encodes FIRST, recursively down the element;
//   * a workgroup that finds one of its tiles claimed by someone else skips it.
// Every tile is encoded exactly once (the CAS decides), and a workgroup only
// ever waits (look-back) on tiles that are claimed, i.e. whose encoder is
// running or done and was itself started only after ITS predecessor was
// claimed: the tile with the smallest (element-wise) index among the
// unfinished ones never waits, so there is no deadlock at any residency.
// With the whole grid resident nobody steals and this IS the static schedule.
//
// kPersistent = false (capi.hip encoderHardwareDispatch): the same kernel launched with one workgroup per ticket -- the
// hardware dispatches workgroups in index order as slots free up, so a slow CU simply takes fewer tiles.  Raw bytes
// (256 x 1 MiB Zipf bytes: 152.7 -> 141.5 us) and float tiles of 2 or 4 blocks (16384 x 8 Ki bf16: 118 -> 96 us,
// 8192 x 16 Ki: 98 -> 88 us) run this way; 8-block float tiles gain nothing and stay persistent
// (profiles/r05_ab_encoder_hw_dispatch.txt, r05_ab_small_tiles_hw_dispatch.txt).  The claim words stay: a workgroup
// still makes sure its element's previous tile is claimed and encodes it first if it is not, so nothing depends on the
// dispatch order.  Float kernels spill to a slot: a persistent workgroup indexes its own by blockIdx.x, a
// hardware-dispatched one takes a pair from a pool (SpillPool).
//
// Tile descriptors carry {status:2, ..., failed:1 (bit 40), words:32}: the padded word count of a tile (aggregate) or
// of all tiles up to it (inclusive), and a sticky flag set by a tile that overran its stage (caller-supplied
// histogram that does not cover the data, see encodeRows).  The element's last tile finds the flag in its inclusive
// prefix and reports the element as FAILED -- outSize[b] = 0, archive magic cleared -- instead of as a success with
// a corrupt archive.  (Upstream has no such outcome: its per-block scratch is simply overrun, GpuANSEncode.cuh:355-358.)
template <int P, uint32_t FT, bool kSpill, uint32_t kTB, bool kPersistent, bool kWide = false>
__global__ __launch_bounds__(encThreads(kTB)) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_ans_encode(EncodeArgs a) {
  static_assert(kTB >= 2u, "single-block elements are k_ans_encode_pair's");
  static_assert(!kWide || (kSpill && FT != kFloat16 && kTB == kBlocksPerTile && kPersistent), "the wide stage exists for persistent 8-block bf16 / fp32 tiles");
  constexpr uint32_t kThreads = encThreads(kTB);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kCap = encStageCap(P, kSpill, FT, kWide);
  // bookkeeping sits BELOW the stages so that a stage overrun (see encodeRows) can never reach it
  uint4* sTable = (uint4*)smem;
  TileShared* sh = (TileShared*)(smem + 4096);
  uint16_t* sStage = (uint16_t*)(smem + 4096 + 128);
  uint8_t* sRing = smem + 4096 + 128 + kTB * kCap * 2u;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const bool upper = lane >= 32u;
  const uint32_t hl = lane & 31u;
  const uint32_t hw = wave * 2u + (upper ? 1u : 0u);

  uint16_t* stage = sStage + hw * kCap;
  const uint32_t stageLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)stage;
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)sTable;
  // Spill slots (kSpill): a persistent workgroup owns kTB of them; under hardware dispatch (tiles of 2 or 4 blocks)
  // blockIdx.x does not bound what is resident and a wavefront takes a pair from the pool when it has to (SpillPool)
  constexpr bool kPool = kSpill && !kPersistent;
  static_assert(!kPool || kTB < kBlocksPerTile, "8-block float tiles run persistent (hardware dispatch measured no gain there)");
  uint16_t* spillSlot = (kSpill && !kPool) ? a.spill + ((size_t)blockIdx.x * kTB + hw) * encSpillSlotWords(P) : nullptr;
  SpillPool pool;
  pool.base = a.spill;
  pool.flags = a.spillFlags;
  pool.pairs = a.spillPairs;
  pool.pair = kNoSpillPair;

  if (a.absentModulo && blockIdx.x % a.absentModulo == 1u) {
    // test hook: a workgroup that becomes resident late (~0.5 ms after the others)
    for (int i = 0; i < 150; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const uint32_t me = blockIdx.x + 1u;
  const uint32_t B = a.numInBatch;
  auto claimLoad = [&](uint32_t idx) -> uint32_t {
    return __hip_atomic_load(a.claims + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto claimTry = [&](uint32_t idx) -> bool {  // true if this workgroup now owns idx
    uint32_t expected = 0;
    return __hip_atomic_compare_exchange_strong(a.claims + idx, &expected, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
  };
  // Ticket -> (element, tile): tile-major over the B x maxTiles rectangle, or -- batches whose elements differ widely
  // in size -- the host's list of the tiles that exist (a.workMap: element << 16 | tile), element by element.  Either
  // way a tile's predecessor has a smaller ticket and the claim word of a ticket is indexed by the ticket; under a
  // list the descriptors are too (an element's are consecutive), so only tiles that exist need either.
  // (One path for both forms -- a list entry is turned into the rectangle's index and divided like a ticket: two
  // assignments under a branch instead cost the float encoders a register they do not have.)
  auto rectIndexOf = [&](uint32_t t) -> uint32_t {
    if (!a.workMap) return t;
    const uint32_t m = a.workMap[t];
    return (m & 0xffffu) * B + (m >> 16);
  };
  bool firstOwned = false;
  if (tid == 0 && blockIdx.x < a.numTickets) firstOwned = claimTry(blockIdx.x);
  if (kPersistent) {
    for (uint32_t t = blockIdx.x + (tid + 1u) * gridDim.x; t < a.numTickets; t += kThreads * gridDim.x) {
      (void)claimTry(t);  // later tickets: result looked up when the ticket comes up
    }
  }
  // (hardware-dispatched grid: gridDim.x == numTickets, the one ticket of this workgroup is its index)
  for (uint32_t ticket0 = blockIdx.x; ticket0 < a.numTickets; ticket0 += kPersistent ? gridDim.x : a.numTickets) {
    const uint32_t rect0 = rectIndexOf(ticket0);
    const uint32_t tile0 = rect0 / B;
    const uint32_t b = rect0 - tile0 * B;
    const uint32_t size = a.in.size(b);
    const uint32_t nb = divUp(size, kBlockSize);
    const uint32_t numTiles = divUp(nb, kTB);
    if (tile0 >= numTiles) continue;  // uniform (ragged batch)
    if (tid == 0) {
      bool mine = (ticket0 == blockIdx.x) ? firstOwned : (claimLoad(ticket0) == me);
      uint32_t lo = tile0 + 1u;  // empty range: the tile was taken over by somebody else
      if (mine) {
        lo = tile0;
        while (lo > 0u) {
          const uint32_t idx = a.workMap ? ticket0 - tile0 + (lo - 1u) : (lo - 1u) * B + b;
          uint32_t p = claimLoad(idx);
          for (int spin = 0; p == 0u && spin < 4; ++spin) {  // give a running owner's claim time to land
            __builtin_amdgcn_s_sleep(32);
            p = claimLoad(idx);
          }
          if (p != 0u) break;
          if (!claimTry(idx)) break;
          --lo;
        }
      }
      sh->tileLo = lo;
    }
    ldsBarrier();
    const uint32_t tileLo = sh->tileLo;
    if (tileLo > tile0) {
      ldsBarrier();  // everybody has read sh->tileLo
      continue;
    }
    for (uint32_t tile = tileLo; tile <= tile0; ++tile) {
      for (uint32_t i = tid; i < kNumSymbols; i += kThreads) sTable[i] = a.encTable[b * kNumSymbols + i];
      ldsBarrier();

      const uint8_t* in = a.in.ptr(b);
      uint8_t* archive = a.out.ptr(b);
      uint8_t* ans = archive + ansOffsetInArchive(FT, size);

      if (FT != 0 && tile == 0) {
        // GpuFloatHeader (GpuFloatCompress.cuh:325-337) and the zero padding of the
        // non-comp plane(s) up to 16 bytes
        if (tid == 0) {
          FloatHeader h;
          h.magicAndVersion = (kFloatMagic << 16) | kFloatVersion;
          h.size = size;
          h.options = FT | (a.useChecksum ? 0x10u : 0u);
          h.checksum = (a.useChecksum && a.checksum) ? a.checksum[b] : 0u;
          *(FloatHeader*)archive = h;
        }
        if (FT == kFloat32) {
          uint16_t* nc2 = (uint16_t*)(archive + 16u);
          uint8_t* nc1 = archive + 16u + 2u * (size_t)roundUp(size, 8u);
          if (size + tid < roundUp(size, 8u)) nc2[size + tid] = 0;
          if (size + tid < roundUp(size, 16u)) nc1[size + tid] = 0;
        } else {
          uint8_t* nc = archive + 16u;
          if (size + tid < roundUp(size, 16u)) nc[size + tid] = 0;
        }
      }

      const uint32_t block = tile * kTB + hw;
      const bool haveBlock = block < nb;
      uint32_t n = 0;
      if (haveBlock) {
        const uint32_t begin = block * kBlockSize;
        n = size - begin < kBlockSize ? size - begin : kBlockSize;
      }
      // wave-uniform: are both halves full blocks (and the input vector-aligned)?
      const uint32_t firstBlockOfWave = tile * kTB + wave * 2u;
      // (vector loads take any word-aligned address -- elements of a split tensor, rows of a matrix; an element below 16
      // bytes must be a 16-byte aligned window for the tail form's last load, see loadSliceBounded)
      const bool aligned = encVectorLoadsOk<FT>(in, size);
      const bool waveFull = (uint64_t)(firstBlockOfWave + 2u) * kBlockSize <= (uint64_t)size && aligned;
      // ONE full block in the wave and no second one (odd block counts): the
      // lower half keeps the straight-line path; the upper half shadows it -- same block, same table, the same
      // non-compressed bytes stored a second time to the same addresses -- and its results are dropped below
      const bool waveHalf = !waveFull && aligned && firstBlockOfWave + 1u == nb &&
          (uint64_t)(firstBlockOfWave + 1u) * kBlockSize == (uint64_t)size;

      ChunkSource<FT> src;
      // (otherwise an idle half reads, and discards, word 0 of block 0)
      src.init(in, archive, size, haveBlock ? block : (waveHalf ? firstBlockOfWave : 0u));

      uint32_t state;
      uint32_t words;        // words left in the LDS stage
      uint32_t spilled = 0;  // words already in the spill slot
      bool overrun = false;
      if (waveFull || waveHalf) {
        words = encodeRows<P, FT, true, kSpill, false, kPool, false, kWide>(src, n, kRowsPerBlock, tableLds, stageLds, sRing + hw * 512u, hl, upper,
                                                              spillSlot, spilled, state, overrun, &pool);
      } else {
        // rows needed by the larger of the two halves (uniform)
        uint32_t nA = 0;
        if (firstBlockOfWave < nb) {
          uint32_t beginA = firstBlockOfWave * kBlockSize;
          nA = size - beginA < kBlockSize ? size - beginA : kBlockSize;  // first half is never smaller than the second
        }
        if (aligned) {  // uniform: the chunked path bounded by n (kTail); else the scalar path
          words = encodeRows<P, FT, true, kSpill, false, kPool, true, kWide>(src, n, divUp(nA, 32u), tableLds, stageLds, sRing + hw * 512u, hl,
                                                                      upper, spillSlot, spilled, state, overrun, &pool);
        } else {
          words = encodeRows<P, FT, false, kSpill, false, kPool, false, kWide>(src, n, divUp(nA, 32u), tableLds, stageLds, nullptr, hl, upper, spillSlot,
                                                                 spilled, state, overrun, &pool);
        }
      }

      if (haveBlock) {
        // final lane states, 128 contiguous bytes per block (GpuANSEncode.cuh:207, :584-590)
        ((uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl] = state;
        // zero the pad up to the 16-byte boundary (the spilled part is whole vectors)
        const uint32_t padded = roundUp(words, kBlockAlignWords);
        if (words + hl < padded) stage[words + hl] = 0;
      }
      // (block word counts in the low half, the overrun flag above them)
      if (hl == 0) sh->words[hw] = haveBlock ? (spilled + words) | (overrun ? 0x80000000u : 0u) : 0u;
      ldsBarrier();

      if (wave == 0) {
        // local exclusive scan of the padded sizes of the tile's blocks
        const uint32_t mine = (lane < kTB) ? sh->words[lane] : 0u;
        const bool tileFailed = __ballot((mine & 0x80000000u) != 0u) != 0ull;
        const uint32_t myWords = mine & 0x7fffffffu;
        uint32_t myPadded = roundUp(myWords, kBlockAlignWords);
        uint32_t incl = waveInclusiveScan(myPadded, lane);
        const uint32_t aggregate = __shfl(incl, kTB - 1, 64);
        if (lane < kTB) sh->localOff[lane] = incl - myPadded;

        uint64_t* desc = a.tileDesc + (a.workMap ? (size_t)(ticket0 - tile0) : (size_t)b * a.maxTiles);
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], kDescAggregate | (tileFailed ? kDescFailed : 0ull) | (uint64_t)aggregate,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        bool failed = tileFailed;
        // Few tiles of an element in flight (a.pollLong == 0: short waits): the walk written out in place with short
        // pauses, as rounds 1-5 had it -- as a function with its pause bookkeeping it cost the raw-byte encoder 3 us of
        // 144 (profiles/r06_ab_raw_encoder_regression_u8.txt).  Many in flight: long pauses, and two levels for float
        // elements of more than 64 tiles (lookBackExclusive, lookBackTwoLevel).
        const bool twoLevel = FT != 0u && a.groupWords != nullptr && numTiles > kLookbackGroup;  // (uniform)
        uint64_t* groupDesc = nullptr;
        uint32_t exclusive = 0;
        if (a.pollLong == 0u && !twoLevel) {
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
            const bool counted = (int)lane <= firstIncl;
            exclusive += waveReduceSum(counted ? (uint32_t)d : 0u);
            failed = failed || __ballot(counted && (d & kDescFailed) != 0ull) != 0ull;
            if (firstIncl < 64) break;
            base -= 64;
          }
        } else if constexpr (FT != 0u) {
          uint64_t* groupArrive = twoLevel ? a.groupWords + (size_t)b * a.groupsPerElement * (kGroupArriveStride + 1u) : nullptr;
          groupDesc = twoLevel ? groupArrive + (size_t)a.groupsPerElement * kGroupArriveStride : nullptr;
          exclusive = twoLevel ? lookBackTwoLevel(desc, groupArrive, groupDesc, tile, numTiles, aggregate, lane, failed, a.pollLong != 0u)
                               : lookBackExclusive(desc, tile, lane, failed, true);
        } else {
          exclusive = lookBackExclusive(desc, tile, lane, failed, true);
        }

        const uint32_t inclusive = exclusive + aggregate;
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], kDescInclusive | (failed ? kDescFailed : 0ull) | (uint64_t)inclusive,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (twoLevel && (tile % kLookbackGroup == kLookbackGroup - 1u || tile == numTiles - 1u)) {
            // the group's last tile: its inclusive prefix is the group's
            __hip_atomic_store(&groupDesc[tile / kLookbackGroup], kDescInclusive | (failed ? kDescFailed : 0ull) | (uint64_t)inclusive,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          sh->tileBase = exclusive;
          if (tile == numTiles - 1) {
            // complete the header (GpuANSEncode.cuh:533-566)
            ((AnsHeader*)ans)->totalCompressedWords = failed ? 0u : inclusive;
            if (failed) ((AnsHeader*)ans)->magicAndVersion = 0u;  // no decoder will follow this archive
            if (a.outSize) a.outSize[b] = failed ? 0u : ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2u * inclusive;
          }
        }
        // per-block word counts and start offsets (GpuANSEncode.cuh:595-608)
        uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(nb));
        const uint32_t blk = tile * kTB + lane;
        if (lane < kTB && blk < nb) {
          const uint32_t begin = blk * kBlockSize;
          const uint32_t bn = size - begin < kBlockSize ? size - begin : kBlockSize;
          blockWords[blk] = make_uint2((bn << 16) | myWords, exclusive + (incl - myPadded));
        }
        if (tile == numTiles - 1 && (nb & 1u) && lane == kTB) {
          uint32_t zero = 0;  // (made here: as a hoisted loop invariant this constant pair has been seen spilled to scratch)
          asm volatile("" : "+v"(zero));
          blockWords[nb] = make_uint2(zero, zero);  // alignment pad entry
        }
      }
      ldsBarrier();

      if (haveBlock) {
        const uint64_t dataOff = (uint64_t)ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2ull * (sh->tileBase + sh->localOff[hw]);
        uint4* dst = (uint4*)(archive + dataOff);
        // 16-byte vectors of this block that still fit the caller's capacity (all of them under the reference's contract)
        const uint64_t room = (uint64_t)a.outCapacity > dataOff ? ((uint64_t)a.outCapacity - dataOff) / 16u : 0u;
        uint32_t fit = room > 0xffffffffull ? 0xffffffffu : (uint32_t)room;
        if (kSpill && spilled) {
          // spilled vectors first (written by this wave; its stores must have been performed)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const uint4* sp = (const uint4*)(kPool ? pool.base + ((size_t)pool.pair * 2u + (upper ? 1u : 0u)) * encSpillSlotWords(P) : spillSlot);
          uint32_t sv = spilled / kBlockAlignWords;
          const uint32_t svFit = sv < fit ? sv : fit;
          // (the lane index is laundered: what this rare block derives from it -- 64-bit slot offsets -- is then computed
          // here instead of being hoisted to the kernel's prologue, where it costs a register the row loops do not have)
          uint32_t hls = hl;
          asm volatile("" : "+v"(hls));
          for (uint32_t i = hls; i < svFit; i += 32u) streamStore<kNtEncStores>(&dst[i], kPool ? coherentLoad16(&sp[i]) : sp[i]);
          dst += sv;
          fit -= svFit;
        }
        uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
        vecs = vecs < fit ? vecs : fit;
        const uint4* s4 = (const uint4*)stage;
        for (uint32_t i = hl; i < vecs; i += 32u) streamStore<kNtEncStores>(&dst[i], s4[i]);
      }
      if (kPool && pool.pair != kNoSpillPair) spillRelease(pool);  // wave-uniform
    }  // tiles [tileLo, tile0] of element b
  }
}

// ---------------------------------------------------------------------------
// Exponent histogram of float inputs: a read-only pass over the float words
// (the histogram the reference fuses into splitFloat,
// GpuFloatCompress.cuh:144, 352-364).  grid = (xBlocks, B), 256 threads;
// hist must be zeroed first.
template <uint32_t FT, uint32_t S, bool kNt = true>
__global__ __launch_bounds__(256) void k_float_histogram(BatchView in, uint32_t* __restrict__ hist, uint32_t partial, HistFuse fuse) {
  __shared__ uint32_t bins[kNumSymbols * S];
  const uint32_t tid = threadIdx.x;
  const HistWork w = histWorkOf(fuse, in, FT == kFloat32 ? 4u : 2u);
  const uint32_t b = w.b;
  histZero<S>(bins, tid);
  __syncthreads();
  uint32_t* myBins = histMine<S>(bins, tid);

  const uint32_t n = in.size(b);
  const uint8_t* inBytes = in.ptr(b);
  // words before the first 16-byte boundary (elements of a split tensor and rows of a matrix start anywhere; a count
  // does not care about order, so the vectors simply start at the boundary)
  constexpr uint32_t kWordBytes = FT == kFloat32 ? 4u : 2u;
  constexpr uint32_t kWordsPerVec = 16u / kWordBytes;
  const bool wordAligned = (((uintptr_t)inBytes) & (kWordBytes - 1u)) == 0;
  uint32_t head = wordAligned ? (uint32_t)((16u - ((uintptr_t)inBytes & 15u)) & 15u) / kWordBytes : n;
  head = head < n ? head : n;
  const uint32_t numVec = (n - head) / kWordsPerVec;
  const uint32_t stride = w.parts * 256u;
  const uint4* pv = (const uint4*)(inBytes + (size_t)head * kWordBytes);

  auto addVec = [&](const uint4& x) {
    if (FT == kFloat32) {
      histAdd<S>(myBins, (x.x >> 23) & 0xffu);
      histAdd<S>(myBins, (x.y >> 23) & 0xffu);
      histAdd<S>(myBins, (x.z >> 23) & 0xffu);
      histAdd<S>(myBins, (x.w >> 23) & 0xffu);
    } else {
      constexpr uint32_t kShift = FT == kFloat16 ? 8u : 7u;
      const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        histAdd<S>(myBins, (xw[j] >> kShift) & 0xffu);
        histAdd<S>(myBins, (xw[j] >> (16u + kShift)) & 0xffu);
      }
    }
  };

  // Every workgroup streams ONE contiguous part of the element (whole 16 KiB steps of 4 x 256 vectors; the last part
  // takes the rest), four 16-byte loads in flight per lane (the kernel is HBM-latency bound otherwise).  Parts
  // interleaved at 4 KiB -- workgroup p reading vectors p * 256 + k * parts * 256 -- cost few-large-element batches
  // 20 % (16 x 8 Mi bf16: 32 workgroups striding through one 16 MiB element, 57 us against 46).
  const uint32_t perPart = roundUp(divUp(numVec, w.parts), 1024u);
  const uint32_t vBegin = w.part * perPart < numVec ? w.part * perPart : numVec;
  const uint32_t vEnd = vBegin + perPart < numVec ? vBegin + perPart : numVec;
  uint32_t v = vBegin + tid;
  for (; v + 768u < vEnd; v += 1024u) {
    const uint4 x0 = streamLoad<kNt>(&pv[v]), x1 = streamLoad<kNt>(&pv[v + 256u]), x2 = streamLoad<kNt>(&pv[v + 512u]), x3 = streamLoad<kNt>(&pv[v + 768u]);
    addVec(x0);
    addVec(x1);
    addVec(x2);
    addVec(x3);
  }
  for (; v < vEnd; v += 256u) addVec(streamLoad<kNt>(&pv[v]));

  // head and tail, word by word (the whole element when the input is not even word-aligned)
  const uint32_t loose = n - numVec * kWordsPerVec;
  for (uint32_t j = w.part * 256u + tid; j < loose; j += stride) {
    const uint32_t i = j < head ? j : numVec * kWordsPerVec + j;
    uint32_t c;
    if (FT == kFloat32) c = (((const uint32_t*)inBytes)[i] >> 23) & 0xffu;
    else c = ((uint32_t)((const uint16_t*)inBytes)[i] >> (FT == kFloat16 ? 8u : 7u)) & 0xffu;
    histAdd<S>(myBins, c);
  }
  __syncthreads();
  histStore(hist, partial, fuse, w, tid, histFold<S>(bins, tid));
}

}  // namespace dgpu
