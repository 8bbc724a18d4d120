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
//   * The row step of a full block is branch-free straight-line code: lanes
//     that do not emit store to a private scratch slot, so no exec-mask
//     juggling or branch issue slots are spent (integer VALU ops issue at ~4
//     cycles per wave-instruction per SIMD on this chip, LDS/VMEM instructions
//     at 10-15: every instruction in the row counts).
//   * Each half-wave emits its u16 words into an LDS stage (worst case for raw
//     bytes; 1024 words + a spill slot in temp memory for floats, see
//     encStageCap).  A workgroup (4 waves = 8 blocks = one "tile") publishes the
//     padded word count of its tile and obtains its archive offset with a
//     decoupled look-back over the preceding tiles of the same batch element,
//     then copies the stage to its final place with 16-byte stores.  There is no
//     per-block scratch buffer in HBM and no coalesce pass.
//   * Workgroups are persistent (as many as fit on the chip) and encode tiles in
//     TILE-MAJOR ticket order (ticket t -> element t % B, tile t / B) under a
//     static map protected by per-tile claim words, see k_ans_encode.  A tile
//     only ever waits on tiles that are claimed by a running workgroup: the
//     look-back cannot deadlock whatever order the hardware dispatches
//     workgroups in and however many of them are resident.  With a batch of B
//     elements the predecessor started B tickets earlier, i.e. it has usually
//     finished long before: measured with element-major order the look-back
//     wait was 17 % of a tile's lifetime.
//   * Hand-off words are single 8-byte {status, value} granules written and
//     polled with relaxed agent-scope atomics (write-through sc1 stores /
//     L1-bypassing loads), the placement-independent form for gfx950's
//     non-coherent per-XCD L2s.
#pragma once

#include "format.h"
#include "kernels_stats.h"

// Tile schedule of the encoder: 0 = dynamic tickets (eight counters), 1 = static map WITHOUT
// residency protection (analysis only, can deadlock), 2 = static map with claim words (default).
#ifndef DGPU_SCHEDULE
#define DGPU_SCHEDULE 2
#endif
#define DGPU_STATIC_SCHEDULE (DGPU_SCHEDULE == 1)

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
constexpr uint32_t kFlushRows = 8;
// words of spill slot per block: the worst case of a block (whole vectors)
__host__ __device__ constexpr uint32_t encSpillSlotWords(int P) { return roundUp(encStageWords(P), 8u) + 8u; }

// raw bytes (spilling variant, DGPU_RAW_SPILLS): byte streams carry more bits per symbol than exponents
#ifndef DGPU_RAW_STAGE_WORDS
#define DGPU_RAW_STAGE_WORDS 1664
#endif
constexpr uint32_t kSpillStageWordsRaw = DGPU_RAW_STAGE_WORDS;
__host__ __device__ constexpr uint32_t encStageCap(int P, bool spill, uint32_t ft) {
  return spill ? (ft == kFloat16 ? kSpillStageWordsFp16 : ft == 0 ? kSpillStageWordsRaw : kSpillStageWords) : encStageWords(P);
}
// Blocks per tile = per workgroup: 8 (256 threads), or 4 (128 threads) for batches whose elements have
// at most 4 blocks -- an 8-block tile would leave half of its waves without a block there.
constexpr uint32_t kBlocksPerSmallTile = 4;
// ... and a single wavefront (64 threads) for batches of elements of at most 2 blocks: half the LDS per workgroup,
// twice the resident tiles
constexpr uint32_t kBlocksPerTinyTile = 2;
// ... and, for batches of SINGLE-block elements, a one-wavefront tile with one stage and one ring: the upper
// half of the wave (which shadows the lower one, see waveHalf) shares them -- it writes the same values to the
// same addresses -- so a workgroup needs 7 KiB instead of 10 and more of them are resident
constexpr uint32_t kBlocksPerSingleTile = 1;
__host__ __device__ constexpr uint32_t encThreads(uint32_t tileBlocks) { return tileBlocks * 32u < 64u ? 64u : tileBlocks * 32u; }
__host__ __device__ constexpr uint32_t encLdsBytes(int P, bool spill, uint32_t ft, uint32_t tileBlocks) {
  return 4096u                                       // packed symbol table
      + 128u                                         // tile bookkeeping
      + tileBlocks * encStageCap(P, spill, ft) * 2u  // bitstream stage per half-wave
      + tileBlocks * 512u                            // symbol ring, 16 rows per half-wave
      + 512u;                                        // scratch slots of non-emitting lanes
}

constexpr uint64_t kDescAggregate = 1ull << 62;
constexpr uint64_t kDescInclusive = 2ull << 62;
constexpr uint64_t kDescValueMask = (1ull << 62) - 1;

struct EncodeArgs {
  BatchView in;              // raw bytes (FT == 0) or float words (FT != 0); size(b) = symbols = bytes / words
  BatchView out;             // archive base pointers
  const uint4* encTable;     // [B][256] from k_normalize
  uint32_t maxTiles;         // tiles per element the ticket space is laid out for
  uint32_t numInBatch;       // B
  uint32_t numTickets;       // B * maxTiles
  uint64_t* tileDesc;        // [B][maxTiles], zeroed before launch
  uint32_t* ticket;          // kTicketCounters counters, kTicketStride words apart, zeroed before launch
  uint32_t* claims;          // [maxTiles][B] tile claim words, zeroed before launch (DGPU_SCHEDULE 2)
  uint32_t absentModulo;     // test hook: workgroups with index % absentModulo == 1 start ~0.5 ms late (0 = off)
  uint16_t* spill;           // [gridDim.x][blocks per tile][encSpillSlotWords(P)] (kSpill kernels only)
  uint32_t* outSize;         // [B] nullable
  uint32_t outCapacity;      // bytes the caller has at out.ptr(b): block data beyond it is NOT stored (outSize still
                             // reports the full size); 0xffffffff = the reference's contract (room for the maximum).
                             // The host guarantees that header, tables and non-compressed planes fit.
  uint32_t useChecksum;      // float header only
  const uint32_t* checksum;  // [B] nullable (float header only)
};

// Ticket counters: up to kTicketCounters of them, 128 bytes apart.
#ifndef DGPU_TICKET_COUNTERS
#define DGPU_TICKET_COUNTERS 8
#endif
constexpr uint32_t kTicketCounters = DGPU_TICKET_COUNTERS;
constexpr uint32_t kTicketStride = 32;    // u32 words between counters

struct TileShared {
  uint32_t ticket;
  uint32_t tileBase;          // exclusive prefix (u16 words) of this tile in the element
  uint32_t words[kBlocksPerTile];
  uint32_t localOff[kBlocksPerTile];
};

// Optional in-kernel phase timing (debug builds only: -DDGPU_PHASE_TIMING).
// Thread 0 of every tile records s_memtime at phase boundaries into
// g_phaseBuf[ticket * 8 + k]; tools/phase_timing.py reduces them.
#ifdef DGPU_PHASE_TIMING
__device__ uint64_t* g_phaseBuf = nullptr;
#define DGPU_PHASE(k)                                                             \
  do {                                                                            \
    if (threadIdx.x == 0 && g_phaseBuf) g_phaseBuf[(size_t)phaseSlot * 8 + (k)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define DGPU_PHASE(k) do {} while (0)
#endif

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
    r.v = streamLoad<DGPU_NT_ENC_LOADS != 0>(&((const uint4*)in)[c * 32u + hl]);
    return r;
  }
  __device__ __forceinline__ void consume(const Raw& r, uint32_t, uint32_t hl, uint8_t* ring) const { *(uint4*)(ring + hl * 16u) = r.v; }
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
    r.a = streamLoad<DGPU_NT_ENC_LOADS != 0>(&p[0]);
    r.b = streamLoad<DGPU_NT_ENC_LOADS != 0>(&p[1]);
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
  __device__ __forceinline__ void splitStore(const Raw& r, uint32_t c, uint32_t hl, uint32_t (&comp)[kCompRegs]) const {
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
    streamStore<DGPU_NT_ENC_STORES != 0>(&((uint4*)(nc + c * 512u))[hl], make_uint4(rest[0], rest[1], rest[2], rest[3]));
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
    r.v[0] = streamLoad<DGPU_NT_ENC_LOADS != 0>(&p[0]);
    r.v[1] = streamLoad<DGPU_NT_ENC_LOADS != 0>(&p[1]);
    return r;
  }
  static constexpr uint32_t kCompRegs = 2;
  __device__ __forceinline__ void consume(const Raw& r, uint32_t c, uint32_t hl, uint8_t* ring) const {
    uint32_t comp[2];
    splitStore(r, c, hl, comp);
    *(uint2*)(ring + hl * 8u) = make_uint2(comp[0], comp[1]);
  }
  // FloatTypeInfo<kFloat32>::split (GpuFloatUtils.cuh:181-185): v = rotl(w, 1)
  __device__ __forceinline__ void splitStore(const Raw& r, uint32_t c, uint32_t hl, uint32_t (&comp)[kCompRegs]) const {
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
    streamStore<DGPU_NT_ENC_STORES != 0>((uint4*)(nc2 + c * 256u + hl * 8u), make_uint4(lo[0], lo[1], lo[2], lo[3]));
    *(uint2*)(nc1 + c * 256u + hl * 8u) = make_uint2(hi[0], hi[1]);
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
// kEntry8: the table in LDS holds PACKED 8-byte entries {magic, (2^P - pdf) | cdf' << 12 | shift << 24} (one
// ds_read_b64 per symbol: half the LDS bytes of the 16-byte entry and a quarter of its bank-conflict surface);
// the row then spends three more VALU instructions: the threshold test becomes the sign of
// state + (2^P - pdf) << (31 - P)  (= state - (pdf << (31 - P)) + 2^31), and (2^P - pdf), cdf' are extracted.
#ifndef DGPU_ENC_ENTRY8
#define DGPU_ENC_ENTRY8 0
#endif
__host__ __device__ constexpr bool encEntry8(uint32_t ft, bool spill, uint32_t tileBlocks) {
  return DGPU_ENC_ENTRY8 && ft == 0 && !spill && tileBlocks == kBlocksPerTile;
}
__device__ __forceinline__ uint2 packEntry8(const uint4 e) { return make_uint2(e.y, e.w | (e.z << 12)); }
template <int P>
__device__ __forceinline__ uint4 unpackEntry8(const uint2 p) {
  const uint32_t q = p.y & 0xfffu;
  return make_uint4(((1u << P) - q) << (kStateBits - P), p.x, (p.y >> 12) & 0xfffu, p.y & 0xff000fffu);
}
__device__ __forceinline__ uint2 ldsTableEntry8(uint32_t addr) {
  typedef uint32_t u32x2e __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) u32x2e LdsU2e;
  const u32x2e v = *(const LdsU2e*)(uintptr_t)addr;
  return make_uint2(v.x, v.y);
}

// DGPU_ENC_EXEC_WRITE: the emitting lanes of a full-block row store under the row's ballot as execution mask (1: two
// s_mov instead of the v_cndmask that parks the idle lanes' store on a scratch slot; 2: the renormalisation shift of the
// emitting lanes happens in the same exec window as well -- two v_cndmask fewer per row).  -1 (default) = by element
// type, as measured on MI355X with tools/gpu_r3m.sh (profiles/r03_ab_encoder_exec_write.txt): 2 for raw bytes (encode
// 163.2 -> 152.7 us on 256 x 1 MiB) and bfloat16 (87.2 / 82.8 -> 81.2 us), 1 for float16 (83.2 -> 82.2 us; 86.5 us
// with 2) and float32 (no difference between the three).  Archives are byte-identical either way.
#ifndef DGPU_ENC_EXEC_WRITE
#define DGPU_ENC_EXEC_WRITE -1
#endif
template <uint32_t FT>
constexpr int encExecWrite() {
  return DGPU_ENC_EXEC_WRITE >= 0 ? DGPU_ENC_EXEC_WRITE : ((FT == 0u || FT == kBFloat16) ? 2 : 1);
}
__device__ __forceinline__ void stageWriteUnder(uint64_t vote, uint32_t addr, uint32_t state) {
  asm volatile("s_mov_b64 exec, %2\n\tds_write_b16 %0, %1\n\ts_mov_b64 exec, -1" : : "v"(addr), "v"(state), "s"(vote) : "memory");
}
// ... and the renormalisation shift of the emitting lanes in the same exec window (DGPU_ENC_EXEC_WRITE=2: two
// v_cndmask fewer per row for two s_mov)
__device__ __forceinline__ void stageWriteShiftUnder(uint64_t vote, uint32_t addr, uint32_t& state) {
  asm volatile("s_mov_b64 exec, %[v]\n\tds_write_b16 %[a], %[s]\n\tv_lshrrev_b32 %[s], 16, %[s]\n\ts_mov_b64 exec, -1"
               : [s] "+v"(state) : [a] "v"(addr), [v] "s"(vote) : "memory");
}

template <int P, uint32_t FT, bool kFull, bool kSpill, bool kEntry8 = false>
__device__ __forceinline__ uint32_t encodeRows(
    const ChunkSource<FT>& src,
    uint32_t n,                           // symbols in this half's block (0 = idle half)
    uint32_t maxRows,                     // wave-uniform row count
    const uint4* __restrict__ table,      // LDS
    uint32_t tableLds,                    // LDS address of `table`
    uint32_t stageBase,                   // LDS address of this half's word stage
    uint32_t dummyAddr,                   // LDS address of this lane's scratch slot
    uint8_t* __restrict__ ring,           // LDS, this half's 512-byte symbol ring
    uint32_t hl,
    bool upper,
    uint16_t* __restrict__ spill,         // this half's spill slot (kSpill only)
    uint32_t& spilledOut,                 // words flushed to it (multiple of 8)
    uint32_t& stateOut) {
  constexpr int kExecWrite = encExecWrite<FT>();
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  uint32_t state = kStartState;
  uint32_t outOff = 0;
  uint32_t spilled = 0;
  // Full blocks (DGPU_ENC_SCALAR_POS): the write positions of the two halves are wave-uniform, so they live in two
  // SGPRs as LDS WORD addresses (stage base folded in, advanced by s_bcnt1 of the ballot halves); an emitting
  // lane's slot is v_mbcnt_lo + v_mbcnt_hi over the 64-bit ballot (lower half: its rank among the emitters;
  // upper half: rank + emitters of the lower half) plus one v_mad_i32_i24 that moves the upper half onto its own
  // position -- no 64-bit shift to pick the half, no per-lane popcounts (as in the decoder, kernels_decode.h).
  // MEASURED SLOWER and therefore off: 256 x 1 MiB Zipf bytes encode 169.8 -> 177.3 us (bf16 unchanged, HBM-bound).
  // The row loop is bound by its dependent chain, and the hop VALU -> SALU -> VALU (v_cmp, s_bcnt1, s_sub,
  // v_mad_i32_i24) is longer than the all-VALU chain it replaces, although it is two VALU instructions shorter.
#ifndef DGPU_ENC_SCALAR_POS
#define DGPU_ENC_SCALAR_POS 0
#endif
  constexpr bool kScalarPos = kFull && DGPU_ENC_SCALAR_POS;
  const uint32_t baseLo = __builtin_amdgcn_readlane(stageBase, 0) >> 1;
  const uint32_t baseHi = __builtin_amdgcn_readlane(stageBase, 32) >> 1;
  uint32_t fLo = baseLo, fHi = baseHi;
  int upperSel = upper ? 1 : 0;
  asm volatile("" : "+v"(upperSel));  // a VGPR operand of the multiply-add, not a select to be folded into it

  // Called every kFlushRows rows: make room for the next kFlushRows rows.
  auto makeRoom = [&]() {
    if (!kSpill) return;
    const uint32_t o0 = kScalarPos ? fLo - baseLo : __builtin_amdgcn_readlane(outOff, 0);
    const uint32_t o1 = kScalarPos ? fHi - baseHi : __builtin_amdgcn_readlane(outOff, 32);
    if ((o0 > o1 ? o0 : o1) + kFlushRows * 32u <= encStageCap(P, true, FT)) return;  // wave-uniform
    if (kScalarPos) outOff = upper ? o1 : o0;
    // whole 16-byte vectors go to the spill slot, the (< 8 word) rest moves to the front
    uint32_t nvec = outOff >> 3;
    // cannot happen with a table made from this data's histogram; keeps a
    // mismatching caller-supplied histogram from writing past the slot
    if (spilled + nvec * 8u > encSpillSlotWords(P)) nvec = 0;
    uint4* dst = (uint4*)(spill + spilled);
    for (uint32_t i = hl; i < nvec; i += 32u) {
      const u32x4e v = *(const LdsU4e*)(uintptr_t)(stageBase + 16u * i);
      dst[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
    const uint32_t rem = outOff & 7u;
    uint16_t t = 0;
    if (hl < rem) t = *(const LdsU16e*)(uintptr_t)(stageBase + 2u * (nvec * 8u + hl));
    if (hl < rem) *(LdsU16e*)(uintptr_t)(stageBase + 2u * hl) = t;
    spilled += nvec * 8u;
    outOff = rem;
    if (kScalarPos) {
      fLo = baseLo + (o0 & 7u);
      fHi = baseHi + (o1 & 7u);
    }
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
    //       = state + cdf + (state / pdf) * (2^P - pdf)      (table: k_normalize)
    const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
    const uint32_t next = __umul24(div, e.w) + state + e.z;
    state = valid ? next : state;
    outOff += __popc(vh);
  };

  // Full-block step: straight-line code, no exec-mask change and no branch.
  // (A hand-scheduled variant of this step with an SDWA select -- two instructions
  // shorter -- measured 5 % slower: see DESIGN.md section 4.1.)
  // DGPU_ENC_ABLATE (timing experiments only, archives are WRONG): 1 = constant table entry (no table read), 2 = no
  // stage write, 3 = constant symbol (every lane reads entry 0: no bank conflicts), 4 = no division (shift instead of
  // mul_hi), 5 = constant emit address (no position arithmetic), 6 = never emit (no ballot-dependent work at all)
#ifndef DGPU_ENC_ABLATE
#define DGPU_ENC_ABLATE 0
#endif
  auto stepFullC = [&](const uint4 e) {
#if DGPU_ENC_ABLATE == 6
    {
      const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
      state = (__umul24(div, e.w) + state + e.z) & 0x7fffffffu;
      return;
    }
#endif
#if DGPU_ENC_ABLATE == 5
    {
      const bool write5 = state >= e.x;
      *(LdsU16e*)(uintptr_t)dummyAddr = (uint16_t)state;
      state = write5 ? (state >> kEncodedBits) : state;
      const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
      state = __umul24(div, e.w) + state + e.z;
      return;
    }
#endif
    const bool write = state >= e.x;
    const uint64_t vote = __ballot(write);
    if (kScalarPos) {
      const uint32_t vLo = (uint32_t)vote, vHi = (uint32_t)(vote >> 32);
      const uint32_t fLoOld = fLo, fHiOld = fHi;
      fLo += (uint32_t)__popc(vLo);
      fHi += (uint32_t)__popc(vHi);
      // emitters below me in the wave: lower half = my rank, upper half = rank + emitters of the lower half
      uint32_t t = __builtin_amdgcn_mbcnt_hi(vHi, __builtin_amdgcn_mbcnt_lo(vLo, 0u));
      // lower: fLoOld + rank; upper: fHiOld + rank = fLoOld + (rank + emittersLo) + (fHiOld - fLo)
      t = (uint32_t)(__mul24(upperSel, (int)(fHiOld - fLo)) + (int)t);
      asm volatile("" : "+v"(t));  // keep the scalar position in the add-shift below (one SGPR operand per VALU op)
      const uint32_t addr = write ? ((t + fLoOld) << 1) : dummyAddr;
      *(LdsU16e*)(uintptr_t)addr = (uint16_t)state;
      state = write ? (state >> kEncodedBits) : state;
      const uint32_t div = __umulhi(state, e.y) >> (e.w >> 24);
      state = __umul24(div, e.w) + state + e.z;
      return;
    }
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    const uint32_t idx = outOff + __popc(vh & laneMaskLt);
    if (kExecWrite == 2) {
      stageWriteShiftUnder(vote, stageBase + 2u * idx, state);
    } else if (kExecWrite) {
      stageWriteUnder(vote, stageBase + 2u * idx, state);
    } else if (DGPU_ENC_ABLATE != 2) {
      const uint32_t addr = write ? stageBase + 2u * idx : dummyAddr;
      *(LdsU16e*)(uintptr_t)addr = (uint16_t)state;
    }
    if (kExecWrite != 2) state = write ? (state >> kEncodedBits) : state;
    const uint32_t div = DGPU_ENC_ABLATE == 4 ? (state >> 9) : (__umulhi(state, e.y) >> (e.w >> 24));
    state = __umul24(div, e.w) + state + e.z;
    if (DGPU_ENC_ABLATE == 4) state &= 0x7fffffffu;
    outOff += __popc(vh);
    if (DGPU_ENC_ABLATE) outOff &= 1023u;  // (wrong tables may emit more than a stage holds)
  };
  // the same step on a packed 8-byte entry
  auto stepFullC8 = [&](const uint2 p) {
    const uint32_t q = p.y & 0xfffu;
    const uint32_t u = __umul24(q, 1u << (kStateBits - P)) + state;  // >= 2^31  <=>  state >= pdf << (31 - P)
    const bool write = (int)u < 0;
    const uint64_t vote = __ballot(write);
    const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
    const uint32_t idx = outOff + __popc(vh & laneMaskLt);
    if (kExecWrite) {
      stageWriteUnder(vote, stageBase + 2u * idx, state);
    } else {
      const uint32_t addr = write ? stageBase + 2u * idx : dummyAddr;
      *(LdsU16e*)(uintptr_t)addr = (uint16_t)state;
    }
    state = write ? (state >> kEncodedBits) : state;
    const uint32_t div = __umulhi(state, p.x) >> (p.y >> 24);
    state = __umul24(div, q) + state + ((p.y >> 12) & 0xfffu);
    outOff += __popc(vh);
  };
  static_assert(!kEntry8 || !DGPU_ENC_SCALAR_POS, "");
  if (kFull) {
    // chunks of 16 rows (8 for fp32); chunk c+1 is in flight in registers while chunk c is
    // consumed from the LDS ring (same wave writes and reads it: LDS ops of one
    // wave execute in order, no barrier needed).
    // table entries in flight ahead of the row being encoded (measured: 2 beats 4 --
    // eight registers fewer matter more than the extra LDS latency cover at 6 waves per SIMD)
#ifndef DGPU_ENC_AHEAD
#define DGPU_ENC_AHEAD 2
#endif
    constexpr int kAhead = DGPU_ENC_AHEAD;
    constexpr uint32_t kChunkRows = ChunkSource<FT>::kRows;
    typename ChunkSource<FT>::Raw cur = src.load(0, hl);
#pragma unroll 1
    for (uint32_t c = 0; c < kRowsPerBlock / kChunkRows; ++c) {
      src.consume(cur, c, hl, ring);
      if (c + 1 < kRowsPerBlock / kChunkRows) cur = src.load(c + 1, hl);
      // LDS addresses of the table entries (table + sym * 16), formed right at the
      // symbol load; kSymAhead symbols and kAhead table entries are in flight.
      // (Reading all 16 symbols of the chunk up front costs 16 live registers.)
#ifndef DGPU_ENC_SYM_AHEAD
#define DGPU_ENC_SYM_AHEAD 4
#endif
      constexpr int kSymAhead = DGPU_ENC_SYM_AHEAD;
      static_assert(kSymAhead > kAhead && kSymAhead <= (int)kChunkRows, "a symbol slot is reused only after its table load was issued");
      // DGPU_ENC_LATE_ADDR: the window holds the raw symbol values and the table address is formed where the table
      // entry is fetched (two rows after the symbol was requested), not where the symbol is requested: with the
      // address pinned at the request (0) every row waits for the LDS round trip of the byte it has just asked for.
#ifndef DGPU_ENC_LATE_ADDR
#define DGPU_ENC_LATE_ADDR 1
#endif
      auto symAddr = [&](int r) -> uint32_t {
        if (DGPU_ENC_LATE_ADDR) return (uint32_t)ring[r * 32 + hl];
        uint32_t t = tableLds + ((uint32_t)ring[r * 32 + hl] << (kEntry8 ? 3 : 4));
        asm volatile("" : "+v"(t));  // keep the scaled address; do not re-derive it (with a mask) at the use
        return t;
      };
      auto entryAddr = [&](uint32_t v) -> uint32_t {
        return DGPU_ENC_LATE_ADDR ? tableLds + (v << (kEntry8 ? 3 : 4)) : v;
      };
      uint32_t toff[kSymAhead];
#pragma unroll
      for (int r = 0; r < kSymAhead; ++r) toff[r] = symAddr(r);
      if constexpr (kEntry8) {
        uint2 e[kAhead];
#pragma unroll
        for (int r = 0; r < kAhead; ++r) e[r] = ldsTableEntry8(entryAddr(toff[r]));
#pragma unroll
        for (int r = 0; r < (int)kChunkRows; ++r) {
          if (r % kFlushRows == 0) makeRoom();
          const uint2 cur_e = e[r % kAhead];
          if (r + kAhead < (int)kChunkRows) e[r % kAhead] = ldsTableEntry8(entryAddr(toff[(r + kAhead) % kSymAhead]));
          if (r + kSymAhead < (int)kChunkRows) toff[r % kSymAhead] = symAddr(r + kSymAhead);
          stepFullC8(cur_e);
        }
      } else {
      uint4 e[kAhead];
      auto fetchEntry = [&](uint32_t v) -> uint4 {
#if DGPU_ENC_ABLATE == 1
        return make_uint4(0x00800000u + (v & 0xff0u), 0x80000001u, 3u, 1020u | (1u << 24));  // no LDS read
#elif DGPU_ENC_ABLATE == 3
        return ldsTableEntry(tableLds + (v & 0u));  // every lane the same entry
#else
        return ldsTableEntry(entryAddr(v));
#endif
      };
#pragma unroll
      for (int r = 0; r < kAhead; ++r) e[r] = fetchEntry(toff[r]);
#pragma unroll
      for (int r = 0; r < (int)kChunkRows; ++r) {
        if (r % kFlushRows == 0) makeRoom();
        const uint4 cur_e = e[r % kAhead];
        if (r + kAhead < (int)kChunkRows) e[r % kAhead] = fetchEntry(toff[(r + kAhead) % kSymAhead]);
        if (r + kSymAhead < (int)kChunkRows) toff[r % kSymAhead] = symAddr(r + kSymAhead);
        stepFullC(cur_e);
      }
      }
    }
  } else {
    // Partial blocks, unaligned inputs, a wave with a single block: rows in groups
    // of kFlushRows, the symbol loads (and, for floats, the non-compressed stores)
    // of a whole group issued before its first step, so a group costs one memory
    // round trip instead of one per row.
#pragma unroll 1
    for (uint32_t row0 = 0; row0 < maxRows; row0 += kFlushRows) {
      makeRoom();
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
        taddr[j] = tableLds + ((src.splitAt(i, word[j], i < n) & 0xffu) << (kEntry8 ? 3 : 4));
      }
      auto entryAt = [&](uint32_t addr) -> uint4 {
        if (kEntry8) return unpackEntry8<P>(ldsTableEntry8(addr));
        return ldsTableEntry(addr);
      };
      uint4 ent[2] = {entryAt(taddr[0]), entryAt(taddr[1])};
#pragma unroll
      for (uint32_t j = 0; j < kFlushRows; ++j) {
        const uint32_t i = (row0 + j) * 32u + hl;
        const uint4 cur_e = ent[j % 2u];
        if (j + 2u < kFlushRows) ent[j % 2u] = entryAt(taddr[j + 2u]);
        if (row0 + j < maxRows) step(cur_e, i < n);  // uniform condition
      }
    }
  }
  // The copy-out reads the slot back through other lanes of this wave, after two
  // workgroup barriers: workgroup-scope ordering is all that is needed (the
  // waves of a workgroup share their CU's L1), no L2 write-back.
  spilledOut = spilled;
  stateOut = state;
  if (kScalarPos) return upper ? fHi - baseHi : fLo - baseLo;
  return outOff;
}

template <int P, uint32_t FT, bool kSpill, uint32_t kTB>
__global__ __launch_bounds__(encThreads(kTB)) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_ans_encode(EncodeArgs a) {
  constexpr uint32_t kThreads = encThreads(kTB);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t kCap = encStageCap(P, kSpill, FT);
  constexpr bool kEntry8 = encEntry8(FT, kSpill, kTB);
  // bookkeeping sits BELOW the stages so that a stage overrun (only possible
  // without spilling, with a caller-supplied histogram that does not match the
  // data) can never reach it
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

  const uint32_t slot = hw < kTB ? hw : kTB - 1u;  // (kTB == 1: both halves of the wave use slot 0)
  uint16_t* stage = sStage + slot * kCap;
  const uint32_t stageLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)stage;
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)sTable;
  // per-lane scratch slot for non-emitting lanes (512 bytes after the rings)
  const uint32_t dummyLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(sRing + kTB * 512u) + tid * 2u;
  uint16_t* spillSlot = kSpill ? a.spill + ((size_t)blockIdx.x * kTB + slot) * encSpillSlotWords(P) : nullptr;

#if DGPU_SCHEDULE == 2
  // Persistent workgroups, STATIC tile map with claim words.  Workgroup w owns the
  // tickets w, w + G, w + 2G, ... (G = gridDim.x; ticket t -> element t % B, tile
  // t / B) and walks them in order: no ticket atomic on the critical path and a
  // perfectly regular round structure (106 -> 93 us for the kernel).  A plain
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
  bool firstOwned = false;
  if (tid == 0 && blockIdx.x < a.numTickets) firstOwned = claimTry(blockIdx.x);
  for (uint32_t t = blockIdx.x + (tid + 1u) * gridDim.x; t < a.numTickets; t += kThreads * gridDim.x) {
    (void)claimTry(t);  // later tickets: result looked up when the ticket comes up
  }
#ifdef DGPU_ENC_STAGGER_UNITS
  // Experiment (off): the workgroups that share a CU (blockIdx i, i + 256, ...) start a fraction of a tile time
  // apart (after their claims are out, so nobody takes their tiles over).  The first tile of a workgroup takes
  // 44 K cycles, the later ones 29-32 K (tools/phase_timing.py), which looked like "everybody loads, then
  // everybody computes"; staggering by 2 / 4 / 8 K cycles per slot changes nothing / +1 / +4 us: the first
  // round is slower because all six workgroups of a CU share the memory system then -- the kernel moves its
  // bytes at the fabric rate either way (profiles/r02_ab_encoder_stagger.txt).
  for (uint32_t i = 0, n = (blockIdx.x >> 8) * DGPU_ENC_STAGGER_UNITS; i < n; ++i) __builtin_amdgcn_s_sleep(32);
#endif
  for (uint32_t ticket0 = blockIdx.x; ticket0 < a.numTickets; ticket0 += gridDim.x) {
    const uint32_t tile0 = ticket0 / B;
    const uint32_t b = ticket0 - tile0 * B;
    {
      const uint32_t tilesOfB = divUp(divUp(a.in.size(b), kBlockSize), kTB);
      if (tile0 >= tilesOfB) continue;  // uniform (ragged batch)
    }
    if (tid == 0) {
      bool mine = (ticket0 == blockIdx.x) ? firstOwned : (claimLoad(ticket0) == me);
      uint32_t lo = tile0 + 1u;  // empty range: the tile was taken over by somebody else
      if (mine) {
        lo = tile0;
        while (lo > 0u) {
          const uint32_t idx = (lo - 1u) * B + b;
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
      sh->ticket = lo;
    }
    ldsBarrier();
    const uint32_t tileLo = sh->ticket;
    if (tileLo > tile0) {
      ldsBarrier();  // everybody has read sh->ticket
      continue;
    }
   for (uint32_t tile = tileLo; tile <= tile0; ++tile) {
    const uint32_t ticket = tile * B + b;
    (void)ticket;  // only the phase-timing build uses it
#elif DGPU_SCHEDULE == 1  // analysis only: UNSAFE unless every workgroup of the grid is resident
  for (uint32_t ticket = blockIdx.x; ticket < a.numTickets; ticket += gridDim.x) {
#else
  // Persistent workgroup: tiles are drawn from ticket counters until they run out.
  // Several ticket counters instead of one (1536 workgroups queueing on a single
  // address cost ~12 us at kernel start): with C counters, C the largest power
  // of two <= kTicketCounters that divides B, counter x hands out the tickets
  // x, x + C, x + 2C, ...  A ticket's predecessors t - B, t - 2B, ... are in its
  // own residue class and smaller tickets of a class are always drawn first, so
  // a tile only ever waits on tiles drawn by running workgroups.
  uint32_t numCounters = kTicketCounters;
  while (a.numInBatch % numCounters) numCounters >>= 1;
  uint32_t myCounter = 0;
  if (tid == 0) myCounter = blockIdx.x % numCounters;
  uint32_t exhausted = 0;  // thread 0: counters seen to be exhausted
  for (;;) {
    if (tid == 0) {
      uint32_t t = 0xffffffffu;
      while (exhausted < numCounters) {
        const uint32_t k = __hip_atomic_fetch_add(a.ticket + myCounter * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t cand = myCounter + k * numCounters;
        if (cand < a.numTickets) {
          t = cand;
          break;
        }
        ++exhausted;  // every counter is visited at most once more after it ran dry
        myCounter = (myCounter + 1u) % numCounters;
      }
      sh->ticket = t;
    }
    ldsBarrier();
    const uint32_t ticket = sh->ticket;
    if (ticket >= a.numTickets) break;  // uniform for the workgroup
#endif
#ifdef DGPU_PHASE_TIMING
    const uint32_t phaseSlot = ticket;
#endif
    DGPU_PHASE(0);
#ifdef DGPU_PHASE_TIMING
    if (threadIdx.x == 0 && g_phaseBuf) g_phaseBuf[(size_t)phaseSlot * 8 + 7] = blockIdx.x;
#endif
#if DGPU_SCHEDULE != 2
    const uint32_t tile = ticket / a.numInBatch;
    const uint32_t b = ticket - tile * a.numInBatch;
#endif

    const uint32_t size = a.in.size(b);
    const uint32_t nb = divUp(size, kBlockSize);
    const uint32_t numTiles = divUp(nb, kTB);
#if DGPU_SCHEDULE != 2
    if (tile >= numTiles) {  // uniform; nobody reads sh->ticket after the barrier below
#if DGPU_SCHEDULE == 0
      ldsBarrier();
#endif
      continue;
    }
#endif

    if constexpr (kEntry8) {
      for (uint32_t i = tid; i < kNumSymbols; i += kThreads) ((uint2*)sTable)[i] = packEntry8(a.encTable[b * kNumSymbols + i]);
    } else {
      for (uint32_t i = tid; i < kNumSymbols; i += kThreads) sTable[i] = a.encTable[b * kNumSymbols + i];
    }
    ldsBarrier();
    DGPU_PHASE(1);

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
    const bool aligned = (((uintptr_t)in & 15u) == 0);
    const bool waveFull = (uint64_t)(firstBlockOfWave + 2u) * kBlockSize <= (uint64_t)size && aligned;
    // ONE full block in the wave and no second one (batches of single-block elements, odd block counts): the
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
    if (waveFull || waveHalf) {
      words = encodeRows<P, FT, true, kSpill, kEntry8>(src, n, kRowsPerBlock, sTable, tableLds, stageLds, dummyLds,
                                              sRing + slot * 512u, hl, upper, spillSlot, spilled, state);
    } else {
      // rows needed by the larger of the two halves (uniform)
      uint32_t nA = 0;
      if (firstBlockOfWave < nb) {
        uint32_t beginA = firstBlockOfWave * kBlockSize;
        nA = size - beginA < kBlockSize ? size - beginA : kBlockSize;  // first half is never smaller than the second
      }
      words = encodeRows<P, FT, false, kSpill, kEntry8>(src, n, divUp(nA, 32u), sTable, tableLds, stageLds, dummyLds,
                                               nullptr, hl, upper, spillSlot, spilled, state);
    }

    DGPU_PHASE(2);
    if (!kSpill) words = words < kCap ? words : kCap;
    if (haveBlock) {
      // final lane states, 128 contiguous bytes per block (GpuANSEncode.cuh:207, :584-590)
      ((uint32_t*)(ans + ansStatesOffset()))[block * 32u + hl] = state;
      // zero the pad up to the 16-byte boundary (the spilled part is whole vectors)
      const uint32_t padded = roundUp(words, kBlockAlignWords);
      if (words + hl < padded) stage[words + hl] = 0;
    }
    if (hl == 0) sh->words[hw] = haveBlock ? spilled + words : 0u;
    ldsBarrier();
    DGPU_PHASE(3);

    if (wave == 0) {
      // local exclusive scan of the padded sizes of the tile's 8 blocks
      uint32_t myPadded = (lane < kTB) ? roundUp(sh->words[lane], kBlockAlignWords) : 0u;
      uint32_t incl = waveInclusiveScan(myPadded, lane);
      const uint32_t aggregate = __shfl(incl, kTB - 1, 64);
      if (lane < kTB) sh->localOff[lane] = incl - myPadded;

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
            a.outSize[b] = ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2u * inclusive;
          }
        }
      }
      // per-block word counts and start offsets (GpuANSEncode.cuh:595-608)
      uint2* blockWords = (uint2*)(ans + ansBlockWordsOffset(nb));
      const uint32_t blk = tile * kTB + lane;
      if (lane < kTB && blk < nb) {
        const uint32_t begin = blk * kBlockSize;
        const uint32_t bn = size - begin < kBlockSize ? size - begin : kBlockSize;
        blockWords[blk] = make_uint2((bn << 16) | sh->words[lane], exclusive + (incl - myPadded));
      }
      if (tile == numTiles - 1 && (nb & 1u) && lane == kTB) {
        blockWords[nb] = make_uint2(0u, 0u);  // alignment pad entry
      }
    }
    ldsBarrier();
    DGPU_PHASE(4);

    if (haveBlock) {
      const uint64_t dataOff = (uint64_t)ansOffsetInArchive(FT, size) + ansOverhead(nb) + 2ull * (sh->tileBase + sh->localOff[hw]);
      uint4* dst = (uint4*)(archive + dataOff);
      // 16-byte vectors of this block that still fit the caller's capacity (all of them under the reference's contract)
      const uint64_t room = (uint64_t)a.outCapacity > dataOff ? ((uint64_t)a.outCapacity - dataOff) / 16u : 0u;
      uint32_t fit = room > 0xffffffffull ? 0xffffffffu : (uint32_t)room;
      if (kSpill && spilled) {
        // spilled vectors first (written by this wave; its stores must have been performed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4* sp = (const uint4*)spillSlot;
        uint32_t sv = spilled / kBlockAlignWords;
        const uint32_t svFit = sv < fit ? sv : fit;
        for (uint32_t i = hl; i < svFit; i += 32u) streamStore<DGPU_NT_ENC_STORES != 0>(&dst[i], sp[i]);
        dst += sv;
        fit -= svFit;
      }
      uint32_t vecs = roundUp(words, kBlockAlignWords) / kBlockAlignWords;
      vecs = vecs < fit ? vecs : fit;
      const uint4* s4 = (const uint4*)stage;
      for (uint32_t i = hl; i < vecs; i += 32u) streamStore<DGPU_NT_ENC_STORES != 0>(&dst[i], s4[i]);
    }
    DGPU_PHASE(5);
#if DGPU_SCHEDULE == 2
   }  // tiles [tileLo, tile0] of element b
#endif
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
  const uint32_t b = blockIdx.y;
  histZero<S>(bins, tid);
  __syncthreads();
  uint32_t* myBins = histMine<S>(bins, tid);

  const uint32_t n = in.size(b);
  const uint8_t* inBytes = in.ptr(b);
  const bool aligned = (((uintptr_t)inBytes) & 15u) == 0;

  constexpr uint32_t kWordsPerVec = FT == kFloat32 ? 4u : 8u;
  const uint32_t numVec = aligned ? n / kWordsPerVec : 0u;
  const uint32_t stride = gridDim.x * 256u;
  const uint4* pv = (const uint4*)inBytes;

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

  // four 16-byte loads in flight per lane (the kernel is HBM-latency bound otherwise)
  uint32_t v = blockIdx.x * 256u + tid;
  for (; v + 3u * stride < numVec; v += 4u * stride) {
    const uint4 x0 = streamLoad<kNt>(&pv[v]), x1 = streamLoad<kNt>(&pv[v + stride]), x2 = streamLoad<kNt>(&pv[v + 2u * stride]), x3 = streamLoad<kNt>(&pv[v + 3u * stride]);
    addVec(x0);
    addVec(x1);
    addVec(x2);
    addVec(x3);
  }
  for (; v < numVec; v += stride) addVec(streamLoad<kNt>(&pv[v]));

  // tail (and the whole element when the input is not 16-byte aligned)
  for (uint32_t i = numVec * kWordsPerVec + blockIdx.x * 256u + tid; i < n; i += stride) {
    uint32_t c;
    if (FT == kFloat32) c = (((const uint32_t*)inBytes)[i] >> 23) & 0xffu;
    else c = ((uint32_t)((const uint16_t*)inBytes)[i] >> (FT == kFloat16 ? 8u : 7u)) & 0xffu;
    histAdd<S>(myBins, c);
  }
  __syncthreads();
  histStore(hist, partial, fuse, b, tid, histFold<S>(bins, tid));
}

}  // namespace dgpu
