/*
 * dietgpu_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see dietgpu_oracle.h).
 *
 * A sequential restatement of the reference's CUDA kernels.  One reference
 * "warp" (32 lanes) is simulated with a 32-entry state array; within a row,
 * encoders emit in ascending lane order (ballot + popc(lanemask_lt)) and
 * decoders consume from the end in descending lane order (lanemask_ge).
 *
 * Compile with -ffp-contract=off: the one floating point expression
 * (GpuANSStatistics.cuh:215) must be a correctly rounded fp32 divide followed
 * by a correctly rounded fp32 multiply, then truncation.
 */
#define _GNU_SOURCE /* pthread barriers / affinity, mallopt (the bench leg at the end of this file) */
#include "dietgpu_oracle.h"

#include <malloc.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>

/* ---- constants: dietgpu/ans/GpuANSUtils.cuh:33-60 ---- */
#define K_NUM_SYMBOLS 256u
#define K_BLOCK 4096u
#define K_WARP 32u
#define K_STATE_BITS 31
#define K_ENC_BITS 16
#define K_START_STATE (1u << (K_STATE_BITS - K_ENC_BITS)) /* 2^15 */
#define K_MIN_STATE K_START_STATE
#define K_ANS_MAGIC 0xd00du
#define K_ANS_VERSION 0x0001u
#define K_BLOCK_ALIGN 16u
#define K_FLOAT_MAGIC 0xf00fu
#define K_FLOAT_VERSION 0x0001u

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint32_t round_up(uint32_t a, uint32_t b) { return div_up(a, b) * b; }

static inline void put32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
static inline uint32_t get32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void put16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }
static inline uint16_t get16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

/* ------------------------------------------------------------------ sizes */

/* ANSCoalescedHeader::getCompressedOverhead, GpuANSUtils.cuh:68-82 */
uint32_t dgo_ans_compressed_overhead(uint32_t numBlocks) {
  return 32u                      /* sizeof(ANSCoalescedHeader) */
      + 2u * K_NUM_SYMBOLS        /* u16 probs[256] */
      + 128u * numBlocks          /* ANSWarpState[numBlocks] */
      + 8u * round_up(numBlocks, 2u); /* uint2 blockWords, count rounded to 2 */
}

/* getRawCompBlockMaxSize, GpuANSEncode.cuh:31-36 */
static uint32_t raw_comp_block_max_size(uint32_t blockBytes) {
  return round_up(blockBytes + blockBytes / 4u, K_BLOCK_ALIGN);
}

/* getMaxCompressedSize, GpuANSEncode.cu:13-25.  NB the reference passes the
 * block SIZE (4096) as the block COUNT to getCompressedOverhead [sic]. */
uint32_t dgo_ans_max_compressed_size(uint32_t uncompressedBytes) {
  uint32_t blocks = div_up(uncompressedBytes, K_BLOCK);
  size_t raw = dgo_ans_compressed_overhead(K_BLOCK);
  raw += (size_t)raw_comp_block_max_size(K_BLOCK) * blocks;
  raw = (raw + 15u) / 16u * 16u;
  if (raw > (size_t)INT32_MAX) return 0u; /* CHECK_LE(rawSize, INT32_MAX), GpuANSEncode.cu:22 (upstream aborts) */
  return (uint32_t)raw;
}

/* getUncompDataSize, GpuFloatUtils.cuh:123-127,163-167,194-203 */
uint32_t dgo_float_uncomp_data_size(uint32_t ft, uint32_t n) {
  switch (ft) {
    case DGO_FLOAT16:
    case DGO_BFLOAT16:
      return round_up(n, 16u);
    case DGO_FLOAT32:
      return 2u * round_up(n, 8u) + round_up(n, 16u);
    default:
      return 0;
  }
}

/* getMaxFloatCompressedSize, GpuFloatCompress.cu:23-45 */
uint32_t dgo_float_max_compressed_size(uint32_t ft, uint32_t n) {
  if (dgo_ans_max_compressed_size(n) == 0u) return 0u;
  return 16u + dgo_ans_max_compressed_size(n) + dgo_float_uncomp_data_size(ft, n);
}

/* ------------------------------------------------------------- statistics */

void dgo_histogram(const uint8_t* in, uint32_t size, uint32_t counts[256]) {
  memset(counts, 0, 256 * sizeof(uint32_t));
  for (uint32_t i = 0; i < size; ++i) counts[in[i]]++;
}

static int cmp_desc_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x < y) - (x > y);
}

/* normalizeProbabilitiesFromHistogram, GpuANSStatistics.cuh:178-367 */
void dgo_normalize(
    const uint32_t counts[256],
    uint32_t total,
    int probBits,
    uint32_t table[256 * 4]) {
  memset(table, 0, 256 * 4 * sizeof(uint32_t));
  if (total == 0) return; /* :193-195 */

  const uint32_t kProbWeight = 1u << probBits;
  uint32_t key[256];
  int qSum = 0;

  for (uint32_t s = 0; s < 256; ++s) {
    uint32_t count = counts[s];
    /* :215  qProb = kProbWeight * ((float)count / (float)totalNum); */
    volatile float ratio = (float)count / (float)total;
    volatile float scaled = (float)kProbWeight * ratio;
    uint32_t q = (uint32_t)scaled;
    /* :218 */
    q = (count > 0 && q == 0) ? 1u : q;
    qSum += (int)q;
    key[s] = (q << 16) | s; /* :234 */
  }

  /* :239-241 BlockRadixSort::SortDescending -- keys are unique */
  qsort(key, 256, sizeof(uint32_t), cmp_desc_u32);

  uint32_t rankSym[256], rankQ[256];
  for (uint32_t r = 0; r < 256; ++r) {
    rankSym[r] = key[r] & 0xffffu;
    rankQ[r] = key[r] >> 16;
  }

  int diff = (int)kProbWeight - qSum; /* :256 */
  if (diff > 0) {
    /* :258-274  NB: compares the entry's SYMBOL index, not its rank */
    while (diff > 0) {
      int iter = diff < 256 ? diff : 256;
      for (uint32_t r = 0; r < 256; ++r) {
        if ((int)rankSym[r] < iter) rankQ[r] += 1;
      }
      diff -= iter;
    }
  } else if (diff < 0) {
    /* :275-315 subtract 1 from the smallest entries that are still > 1 */
    diff = -diff;
    while (diff > 0) {
      int numGt1 = 0;
      for (uint32_t r = 0; r < 256; ++r) numGt1 += (rankQ[r] > 1);
      int iter = diff < numGt1 ? diff : numGt1;
      if (iter <= 0) break; /* reference asserts iter > 0 */
      int start = numGt1 - iter;
      for (int r = start; r < numGt1; ++r) rankQ[r] -= 1;
      diff -= iter;
    }
  }

  /* :318-334 un-sort */
  uint32_t pdf[256];
  for (uint32_t r = 0; r < 256; ++r) pdf[rankSym[r]] = rankQ[r];

  /* :336-341 exclusive scan; :349-358 division magic */
  uint32_t cdf = 0;
  for (uint32_t s = 0; s < 256; ++s) {
    uint32_t p = pdf[s];
    uint32_t shift = 0, magic = 0;
    if (p > 0) {
      /* shift = 32 - clz(p - 1); clz(0) == 32 on the GPU */
      uint32_t pm1 = p - 1;
      shift = pm1 == 0 ? 0u : (32u - (uint32_t)__builtin_clz(pm1));
      uint64_t one = 1;
      uint64_t magic64 = ((one << 32) * ((one << shift) - p)) / p + 1;
      magic = (uint32_t)magic64;
    }
    table[s * 4 + 0] = p;
    table[s * 4 + 1] = cdf;
    table[s * 4 + 2] = magic;
    table[s * 4 + 3] = shift;
    cdf += p;
  }
}

/* checksumSingle, GpuChecksum.cuh:26-93 */
uint32_t dgo_checksum(const uint8_t* in, uint32_t size) {
  uint32_t c = 0;
  for (uint32_t i = 0; i < size; ++i) c ^= in[i];
  return c & 0xffu;
}

/* -------------------------------------------------------------- ANS block */

/* encodeOneWarp / encodeOnePartialWarp / ansEncodeWarpBlock,
 * GpuANSEncode.cuh:49-211 */
uint32_t dgo_ans_encode_block(
    const uint8_t* in,
    uint32_t n,
    int probBits,
    const uint32_t table[256 * 4],
    uint16_t* outWords,
    uint32_t outState[32]) {
  uint32_t state[K_WARP];
  for (uint32_t l = 0; l < K_WARP; ++l) state[l] = K_START_STATE; /* :157 */

  const uint32_t kStateCheckMul = 1u << (K_STATE_BITS - probBits); /* :63 */
  uint32_t outOffset = 0;
  uint32_t rows = div_up(n, K_WARP);

  for (uint32_t row = 0; row < rows; ++row) {
    for (uint32_t lane = 0; lane < K_WARP; ++lane) { /* ascending = popc(vote & lanemask_lt) */
      uint32_t i = row * K_WARP + lane;
      if (i >= n) continue; /* partial row: invalid lanes do nothing :113,:132 */
      uint32_t sym = in[i];
      uint32_t pdf = table[sym * 4 + 0];
      uint32_t cdf = table[sym * 4 + 1];
      uint32_t magic = table[sym * 4 + 2];
      uint32_t shift = table[sym * 4 + 3];

      uint32_t s = state[lane];
      if (s >= pdf * kStateCheckMul) { /* :65-66 */
        outWords[outOffset++] = (uint16_t)(s & 0xffffu); /* :73 */
        s >>= K_ENC_BITS;
      }
      /* :79-86 */
      uint32_t t = (uint32_t)(((uint64_t)s * (uint64_t)magic) >> 32);
      uint32_t div = (t + s) >> shift;
      uint32_t mod = s - div * pdf;
      state[lane] = div * (1u << probBits) + mod + cdf;
    }
  }
  memcpy(outState, state, sizeof(state)); /* :207 */
  return outOffset;
}

/* packDecodeLookup + ansDecodeTable, GpuANSDecode.cuh:34-41, 405-476 */
void dgo_ans_decode_table(const uint16_t pdf[256], int probBits, uint32_t* lut) {
  uint32_t total = 1u << probBits;
  memset(lut, 0, total * sizeof(uint32_t));
  uint32_t cdf = 0;
  for (uint32_t s = 0; s < 256; ++s) {
    uint32_t p = pdf[s];
    for (uint32_t j = 0; j < p && cdf + j < total; ++j) {
      lut[cdf + j] = (j << 20) | (p << 8) | s;
    }
    cdf += p;
  }
}

/* decodeOneWarp / decodeOnePartialWarp / ansDecodeWarpBlock,
 * GpuANSDecode.cuh:55-217, 274-297 */
int dgo_ans_decode_block(
    const uint16_t* words,
    uint32_t numWords,
    const uint32_t stateIn[32],
    uint32_t n,
    int probBits,
    const uint32_t* lut,
    uint8_t* out) {
  uint32_t state[K_WARP];
  memcpy(state, stateIn, sizeof(state));
  const uint32_t mask = (1u << probBits) - 1u;
  int64_t pos = numWords;
  uint32_t rows = div_up(n, K_WARP);

  for (int64_t row = (int64_t)rows - 1; row >= 0; --row) { /* partial row first :185-200 */
    for (int lane = (int)K_WARP - 1; lane >= 0; --lane) { /* descending = lanemask_ge */
      uint32_t i = (uint32_t)row * K_WARP + (uint32_t)lane;
      if (i >= n) continue;
      uint32_t s = state[lane];
      uint32_t e = lut[s & mask];
      uint32_t sym = e & 0xffu;
      uint32_t pdf = (e >> 8) & 0xfffu;
      uint32_t sMinusCdf = e >> 20;
      out[i] = (uint8_t)sym;
      s = pdf * (s >> probBits) + sMinusCdf; /* :85 */
      if (s < K_MIN_STATE) { /* :88 */
        if (pos <= 0) return -2;
        pos -= 1;
        s = (s << K_ENC_BITS) + (uint32_t)words[pos]; /* :99-100 */
      }
      state[lane] = s;
    }
  }
  if (pos != 0) return -3;
  for (uint32_t l = 0; l < K_WARP; ++l)
    if (state[l] != K_START_STATE) return -4;
  return 0;
}

/* ------------------------------------------------------------ ANS archive */

/* ansEncodeBatchDevice + ansEncodeCoalesce, GpuANSEncode.cuh:515-628,674-849 */
uint32_t dgo_ans_encode(
    const uint8_t* in,
    uint32_t size,
    int probBits,
    int useChecksum,
    const uint32_t* countsIn,
    uint8_t* out) {
  uint32_t counts[256];
  if (countsIn) {
    memcpy(counts, countsIn, sizeof(counts));
  } else {
    dgo_histogram(in, size, counts);
  }
  uint32_t table[256 * 4];
  dgo_normalize(counts, size, probBits, table);

  uint32_t nb = div_up(size, K_BLOCK);
  uint32_t overhead = dgo_ans_compressed_overhead(nb);

  uint8_t* pdfOut = out + 32;
  uint8_t* statesOut = pdfOut + 2 * K_NUM_SYMBOLS;
  uint8_t* blockWordsOut = statesOut + 128u * nb;
  uint8_t* dataOut = out + overhead;

  /* zero the fixed-size part (incl. the odd blockWords pad entry) */
  memset(out, 0, overhead);

  for (uint32_t s = 0; s < 256; ++s) put16(pdfOut + 2 * s, (uint16_t)table[s * 4]); /* :571-573 */

  uint16_t* words = (uint16_t*)malloc(sizeof(uint16_t) * (K_BLOCK + 8));
  uint32_t startWord = 0;
  for (uint32_t b = 0; b < nb; ++b) {
    uint32_t begin = b * K_BLOCK;
    uint32_t n = size - begin < K_BLOCK ? size - begin : K_BLOCK;
    uint32_t st[32];
    uint32_t w = dgo_ans_encode_block(in + begin, n, probBits, table, words, st);
    memcpy(statesOut + 128u * b, st, 128); /* :584-590 */
    put32(blockWordsOut + 8u * b, (n << 16) | w); /* :606-607 */
    put32(blockWordsOut + 8u * b + 4, startWord);
    uint32_t wPad = round_up(w, K_BLOCK_ALIGN / 2); /* Align<u16,16>: multiple of 8 words */
    uint8_t* dst = dataOut + 2u * (size_t)startWord;
    memcpy(dst, words, 2u * w);
    memset(dst + 2u * w, 0, 2u * (wPad - w));
    startWord += wPad;
  }
  free(words);

  uint32_t totalWords = startWord; /* :533-544 */
  put32(out + 0, (K_ANS_MAGIC << 16) | K_ANS_VERSION);
  put32(out + 4, nb);
  put32(out + 8, size);
  put32(out + 12, totalWords);
  put32(out + 16, (uint32_t)probBits | ((useChecksum ? 1u : 0u) << 4));
  put32(out + 20, useChecksum ? dgo_checksum(in, size) : 0u);
  put32(out + 24, 0);
  put32(out + 28, 0);
  return overhead + 2u * totalWords; /* getTotalCompressedSize :83-86 */
}

int dgo_ans_info(
    const uint8_t* in,
    uint32_t* uncompressedSize,
    uint32_t* compressedSize,
    uint32_t* checksum,
    uint32_t* probBits) {
  uint32_t mv = get32(in);
  if ((mv >> 16) != K_ANS_MAGIC || (mv & 0xffffu) != K_ANS_VERSION) return -1;
  uint32_t nb = get32(in + 4);
  if (uncompressedSize) *uncompressedSize = get32(in + 8);
  if (compressedSize) *compressedSize = dgo_ans_compressed_overhead(nb) + 2u * get32(in + 12);
  if (probBits) *probBits = get32(in + 16) & 0xfu;
  if (checksum) *checksum = get32(in + 20);
  return 0;
}

/* ansDecodeKernel + ansDecodeTable, GpuANSDecode.cuh:299-476 */
int dgo_ans_decode(
    const uint8_t* in,
    int probBits,
    uint8_t* out,
    uint32_t outCapacity,
    uint32_t* outSize) {
  uint32_t mv = get32(in);
  if ((mv >> 16) != K_ANS_MAGIC || (mv & 0xffffu) != K_ANS_VERSION) return -1;
  uint32_t nb = get32(in + 4);
  uint32_t size = get32(in + 8);
  uint32_t opts = get32(in + 16);
  if ((int)(opts & 0xfu) != probBits) return -5; /* assert :323 */
  if (outSize) *outSize = size; /* :335-337 */
  if (outCapacity < size) return 1; /* :325-341 */
  if (size == 0) return 0;

  const uint8_t* pdfIn = in + 32;
  const uint8_t* statesIn = pdfIn + 2 * K_NUM_SYMBOLS;
  const uint8_t* blockWordsIn = statesIn + 128u * nb;
  const uint8_t* dataIn = in + dgo_ans_compressed_overhead(nb);

  uint16_t pdf[256];
  uint32_t pdfSum = 0;
  for (uint32_t s = 0; s < 256; ++s) { pdf[s] = get16(pdfIn + 2 * s); pdfSum += pdf[s]; }
  if (pdfSum != (1u << probBits)) return -6; /* assert :448 */

  uint32_t* lut = (uint32_t*)malloc(sizeof(uint32_t) << probBits);
  dgo_ans_decode_table(pdf, probBits, lut);

  int rc = 0;
  for (uint32_t b = 0; b < nb && rc == 0; ++b) {
    uint32_t st[32];
    memcpy(st, statesIn + 128u * b, 128);
    uint32_t bw = get32(blockWordsIn + 8u * b);
    uint32_t start = get32(blockWordsIn + 8u * b + 4);
    uint32_t n = bw >> 16, w = bw & 0xffffu;
    if ((size_t)b * K_BLOCK + n > size) { rc = -7; break; }
    rc = dgo_ans_decode_block(
        (const uint16_t*)(dataIn + 2u * (size_t)start), w, st, n, probBits, lut,
        out + (size_t)b * K_BLOCK);
  }
  free(lut);
  return rc;
}

/* ------------------------------------------------------------ float codec */

static inline uint32_t rotl32(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
static inline uint32_t rotr32(uint32_t v, int s) { return (v >> s) | (v << (32 - s)); }

/* FloatTypeInfo<FT>::split, GpuFloatUtils.cuh:111-115,141-147,181-185 */
void dgo_float_split(
    uint32_t ft, const void* inV, uint32_t n, uint8_t* comp, uint8_t* nonComp) {
  memset(nonComp, 0, dgo_float_uncomp_data_size(ft, n));
  if (ft == DGO_FLOAT16) {
    const uint16_t* in = (const uint16_t*)inV;
    for (uint32_t i = 0; i < n; ++i) {
      comp[i] = (uint8_t)(in[i] >> 8);
      nonComp[i] = (uint8_t)(in[i] & 0xff);
    }
  } else if (ft == DGO_BFLOAT16) {
    const uint16_t* in = (const uint16_t*)inV;
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t v = (uint32_t)in[i] * 65536u + (uint32_t)in[i];
      v = rotl32(v, 1);
      comp[i] = (uint8_t)(v >> 24);
      nonComp[i] = (uint8_t)(v & 0xff);
    }
  } else if (ft == DGO_FLOAT32) {
    const uint32_t* in = (const uint32_t*)inV;
    /* low 2 bytes as a u16 plane of roundUp(n,8) entries, then the high byte
     * plane (GpuFloatCompress.cuh:59-63) */
    uint8_t* nc2 = nonComp;
    uint8_t* nc1 = nonComp + 2u * round_up(n, 8u);
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t v = rotl32(in[i], 1);
      comp[i] = (uint8_t)(v >> 24);
      uint32_t nc = v & 0xffffffu;
      put16(nc2 + 2u * i, (uint16_t)(nc & 0xffffu));
      nc1[i] = (uint8_t)(nc >> 16);
    }
  }
}

/* FloatTypeInfo<FT>::join, GpuFloatUtils.cuh:117-119,149-159,187-190 */
void dgo_float_join(
    uint32_t ft, const uint8_t* comp, const uint8_t* nonComp, uint32_t n, void* outV) {
  if (ft == DGO_FLOAT16) {
    uint16_t* out = (uint16_t*)outV;
    for (uint32_t i = 0; i < n; ++i)
      out[i] = (uint16_t)((uint16_t)comp[i] * 256u + (uint16_t)nonComp[i]);
  } else if (ft == DGO_BFLOAT16) {
    uint16_t* out = (uint16_t*)outV;
    for (uint32_t i = 0; i < n; ++i) {
      /* shf.r.clamp.b32 out, lo, hi, 1  ==  (lo >> 1) | (hi << 31) */
      uint32_t lo = ((uint32_t)comp[i] * 256u + (uint32_t)nonComp[i]) << 16;
      uint32_t hi = nonComp[i];
      uint32_t o = (lo >> 1) | (hi << 31);
      out[i] = (uint16_t)(o >> 16);
    }
  } else if (ft == DGO_FLOAT32) {
    uint32_t* out = (uint32_t*)outV;
    const uint8_t* nc2 = nonComp;
    const uint8_t* nc1 = nonComp + 2u * round_up(n, 8u);
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t nc = (uint32_t)nc1[i] * 65536u + (uint32_t)get16(nc2 + 2u * i);
      uint32_t v = (uint32_t)comp[i] * 16777216u + nc;
      out[i] = rotr32(v, 1);
    }
  }
}

static uint32_t float_word_size(uint32_t ft) { return ft == DGO_FLOAT32 ? 4u : 2u; }

/* floatCompressDevice, GpuFloatCompress.cuh:446-579 */
uint32_t dgo_float_compress(
    uint32_t ft, const void* in, uint32_t n, int probBits, int useChecksum, uint8_t* out) {
  uint32_t ncBytes = dgo_float_uncomp_data_size(ft, n);
  uint8_t* comp = (uint8_t*)malloc(n ? n : 1);
  dgo_float_split(ft, in, n, comp, out + 16);

  /* GpuFloatHeader, GpuFloatUtils.cuh:26-74; written at GpuFloatCompress.cuh:325-337 */
  put32(out + 0, (K_FLOAT_MAGIC << 16) | K_FLOAT_VERSION);
  put32(out + 4, n);
  put32(out + 8, ft | ((useChecksum ? 1u : 0u) << 4));
  /* Quirk (GpuFloatCompress.cuh:466-468 + GpuChecksum.cuh:100-101): the size
   * handed to the checksum is in float WORDS but is used as a BYTE count, so
   * only the first n bytes of the input are covered. */
  put32(out + 12, useChecksum ? dgo_checksum((const uint8_t*)in, n) : 0u);

  /* ANS on the comp plane; float-level checksum only (GpuFloatCodec.h:50) */
  uint32_t ansBytes = dgo_ans_encode(comp, n, probBits, 0, NULL, out + 16 + ncBytes);
  free(comp);
  return 16u + ncBytes + ansBytes; /* incOutputSizes :369-377 */
}

int dgo_float_info(
    const uint8_t* in, uint32_t* numFloats, uint32_t* floatType, uint32_t* checksum,
    uint32_t* compressedSize) {
  uint32_t mv = get32(in);
  if ((mv >> 16) != K_FLOAT_MAGIC || (mv & 0xffffu) != K_FLOAT_VERSION) return -1;
  uint32_t n = get32(in + 4);
  uint32_t ft = get32(in + 8) & 0xfu;
  if (numFloats) *numFloats = n;
  if (floatType) *floatType = ft;
  if (checksum) *checksum = get32(in + 12);
  if (compressedSize) {
    uint32_t ncBytes = dgo_float_uncomp_data_size(ft, n);
    uint32_t ansBytes = 0;
    if (dgo_ans_info(in + 16 + ncBytes, NULL, &ansBytes, NULL, NULL) != 0) return -2;
    *compressedSize = 16u + ncBytes + ansBytes;
  }
  return 0;
}

/* floatDecompressDevice, GpuFloatDecompress.cuh:565-738 */
int dgo_float_decompress(
    uint32_t ft, const uint8_t* in, int probBits, void* out, uint32_t capFloats,
    uint32_t* outSize) {
  uint32_t n = 0, ftIn = 0;
  if (dgo_float_info(in, &n, &ftIn, NULL, NULL) != 0) return -1;
  if (ftIn != ft) return -8; /* assert(FT == h.getFloatType()) :334 */
  uint32_t ncBytes = dgo_float_uncomp_data_size(ft, n);
  uint32_t ansSize = 0;
  if (dgo_ans_info(in + 16 + ncBytes, &ansSize, NULL, NULL, NULL) != 0) return -2;
  if (outSize) *outSize = ansSize;
  /* capacity is compared in float words == comp bytes (GpuANSDecode.cuh:325-327) */
  if (capFloats < ansSize) return 1;
  if (ansSize != n) return -9; /* joinFloat size check, GpuFloatDecompress.cuh:306-310 */
  uint8_t* comp = (uint8_t*)malloc(n ? n : 1);
  int rc = dgo_ans_decode(in + 16 + ncBytes, probBits, comp, n, &ansSize);
  if (rc == 0) dgo_float_join(ft, comp, in + 16, n, out);
  free(comp);
  (void)float_word_size;
  return rc;
}

/* ---------------------------------------------- batch helpers (cpu_baseline) */

typedef struct {
  int kind; /* 0 ans enc, 1 ans dec, 2 float enc, 3 float dec */
  const uint8_t* in; size_t inStride; uint32_t size; uint32_t batch; int probBits;
  uint8_t* out; size_t outStride; uint32_t* outSizes; uint32_t cap; uint32_t ft;
  int tid, nthreads;
} job_t;

static void* worker(void* p) {
  job_t* j = (job_t*)p;
  for (uint32_t b = (uint32_t)j->tid; b < j->batch; b += (uint32_t)j->nthreads) {
    const uint8_t* in = j->in + (size_t)b * j->inStride;
    uint8_t* out = j->out + (size_t)b * j->outStride;
    uint32_t sz = 0;
    switch (j->kind) {
      case 0: sz = dgo_ans_encode(in, j->size, j->probBits, 0, NULL, out); break;
      case 1: dgo_ans_decode(in, j->probBits, out, j->cap, &sz); break;
      case 2: sz = dgo_float_compress(j->ft, in, j->size, j->probBits, 0, out); break;
      case 3: dgo_float_decompress(j->ft, in, j->probBits, out, j->cap, &sz); break;
    }
    if (j->outSizes) j->outSizes[b] = sz;
  }
  return NULL;
}

static void run_jobs(job_t base, int threads) {
  if (threads < 1) threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base; jobs[t].tid = t; jobs[t].nthreads = threads;
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}

void dgo_ans_encode_batch(
    const uint8_t* in, uint32_t size, size_t inStride, uint32_t batch, int probBits,
    uint8_t* out, size_t outStride, uint32_t* outSizes, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.kind = 0; j.in = in; j.inStride = inStride; j.size = size; j.batch = batch;
  j.probBits = probBits; j.out = out; j.outStride = outStride; j.outSizes = outSizes;
  run_jobs(j, threads);
}

void dgo_ans_decode_batch(
    const uint8_t* in, size_t inStride, uint32_t batch, int probBits, uint8_t* out,
    size_t outStride, uint32_t outCapacity, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.kind = 1; j.in = in; j.inStride = inStride; j.batch = batch; j.probBits = probBits;
  j.out = out; j.outStride = outStride; j.cap = outCapacity;
  run_jobs(j, threads);
}

void dgo_float_compress_batch(
    uint32_t ft, const void* in, uint32_t numFloats, size_t inStride, uint32_t batch,
    int probBits, uint8_t* out, size_t outStride, uint32_t* outSizes, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.kind = 2; j.ft = ft; j.in = (const uint8_t*)in; j.inStride = inStride; j.size = numFloats;
  j.batch = batch; j.probBits = probBits; j.out = out; j.outStride = outStride;
  j.outSizes = outSizes;
  run_jobs(j, threads);
}

void dgo_float_decompress_batch(
    uint32_t ft, const uint8_t* in, size_t inStride, uint32_t batch, int probBits,
    void* out, size_t outStride, uint32_t outCapacityFloats, int threads) {
  job_t j; memset(&j, 0, sizeof(j));
  j.kind = 3; j.ft = ft; j.in = in; j.inStride = inStride; j.batch = batch;
  j.probBits = probBits; j.out = (uint8_t*)out; j.outStride = outStride;
  j.cap = outCapacityFloats;
  run_jobs(j, threads);
}


/* ------------------------------------------------- timed round trip (bench.py's cpu_baseline leg)
 * What the one-call-per-phase helpers above cost on a many-core host is mostly NOT the codec: every call creates and
 * joins its threads, the caller allocates (and first-touches) fresh output arrays, and the per-element temporaries
 * (the 512 KiB exponent plane of a float row) are above glibc's mmap threshold, so 256 threads queue on the process's
 * address-space lock.  Here: persistent threads (one per core, pinned when asked), every buffer allocated and touched
 * before the clock starts, malloc kept off mmap, phases separated by barriers and timed inside.  Row b belongs to
 * thread b mod T.  Returns 0, or -1 on allocation / thread failure; *mismatches = rows that did not round-trip. */
typedef struct {
  uint32_t ft; const uint8_t* in; size_t inStride; uint32_t size; uint32_t batch; int probBits;
  uint8_t* comp; size_t compStride; uint8_t* out; size_t rowBytes;
  int tid, nthreads, reps, pin;
  pthread_barrier_t* bar; double* encSeconds; double* decSeconds; int* bad;
  struct bench_gate* gate;
} bench_t;

/* start gate: the workers touch the barrier only once EVERY thread exists (a barrier sized for `threads` never fills
 * when pthread_create fails midway; the threads already started are then told to leave instead) */
struct bench_gate {
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int state; /* 0: wait, 1: go, 2: abort */
};

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* bench_worker(void* p) {
  bench_t* j = (bench_t*)p;
  pthread_mutex_lock(&j->gate->mu);
  while (j->gate->state == 0) pthread_cond_wait(&j->gate->cv, &j->gate->mu);
  const int go = j->gate->state == 1;
  pthread_mutex_unlock(&j->gate->mu);
  if (!go) return NULL;
  if (j->pin) {
    /* the tid-th CPU of the set this process may run on */
    cpu_set_t allowed, one;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
      int want = j->tid % CPU_COUNT(&allowed), seen = 0;
      for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &allowed)) continue;
        if (seen++ == want) {
          CPU_ZERO(&one);
          CPU_SET(c, &one);
          (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
          break;
        }
      }
    }
  }
  double enc = 0.0, dec = 0.0;
  for (int rep = -1; rep < j->reps; ++rep) { /* rep -1: untimed (first touch of every page, warm caches) */
    pthread_barrier_wait(j->bar);
    const double t0 = now_s();
    for (uint32_t b = (uint32_t)j->tid; b < j->batch; b += (uint32_t)j->nthreads) {
      const uint8_t* in = j->in + (size_t)b * j->inStride;
      uint8_t* c = j->comp + (size_t)b * j->compStride;
      if (j->ft) (void)dgo_float_compress(j->ft, in, j->size, j->probBits, 0, c);
      else (void)dgo_ans_encode(in, j->size, j->probBits, 0, NULL, c);
    }
    pthread_barrier_wait(j->bar);
    const double t1 = now_s();
    for (uint32_t b = (uint32_t)j->tid; b < j->batch; b += (uint32_t)j->nthreads) {
      const uint8_t* c = j->comp + (size_t)b * j->compStride;
      uint8_t* o = j->out + (size_t)b * j->rowBytes;
      uint32_t sz = 0;
      if (j->ft) (void)dgo_float_decompress(j->ft, c, j->probBits, o, j->size, &sz);
      else (void)dgo_ans_decode(c, j->probBits, o, j->size, &sz);
    }
    pthread_barrier_wait(j->bar);
    const double t2 = now_s();
    if (rep >= 0) { enc += t1 - t0; dec += t2 - t1; }
  }
  if (j->tid == 0) { *j->encSeconds = enc; *j->decSeconds = dec; }
  int bad = 0;
  for (uint32_t b = (uint32_t)j->tid; b < j->batch; b += (uint32_t)j->nthreads) {
    if (memcmp(j->in + (size_t)b * j->inStride, j->out + (size_t)b * j->rowBytes, j->rowBytes) != 0) ++bad;
  }
  __atomic_fetch_add(j->bad, bad, __ATOMIC_RELAXED);
  return NULL;
}

int dgo_bench_roundtrip(
    uint32_t ft, const void* in, uint32_t size, size_t inStride, uint32_t batch, int probBits, int threads, int reps,
    int pin, double* encSeconds, double* decSeconds, int* mismatches) {
  if (threads < 1) threads = 1;
  /* per-element temporaries (up to `size` bytes) from the arena heaps, not from mmap / munmap per call */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  const size_t wordBytes = ft == 3 ? 4 : (ft ? 2 : 1);
  const size_t rowBytes = (size_t)size * wordBytes;
  const size_t compStride = ft ? dgo_float_max_compressed_size(ft, size) : dgo_ans_max_compressed_size(size);
  uint8_t* comp = (uint8_t*)malloc(compStride * batch + 64);
  uint8_t* out = (uint8_t*)malloc(rowBytes * batch + 64);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  bench_t* jobs = (bench_t*)malloc(sizeof(bench_t) * (size_t)threads);
  pthread_barrier_t bar;
  struct bench_gate gate;
  int bad = 0, rc = 0;
  if (!comp || !out || !th || !jobs || pthread_barrier_init(&bar, NULL, (unsigned)threads) != 0) {
    free(comp); free(out); free(th); free(jobs);
    mallopt(M_MMAP_THRESHOLD, 128 * 1024);
    mallopt(M_TRIM_THRESHOLD, 128 * 1024);
    return -1;
  }
  pthread_mutex_init(&gate.mu, NULL);
  pthread_cond_init(&gate.cv, NULL);
  gate.state = 0;
  int started = 0;
  for (int t = 0; t < threads; ++t) {
    bench_t j;
    memset(&j, 0, sizeof(j));
    j.ft = ft; j.in = (const uint8_t*)in; j.inStride = inStride; j.size = size; j.batch = batch; j.probBits = probBits;
    j.comp = comp; j.compStride = compStride; j.out = out; j.rowBytes = rowBytes;
    j.tid = t; j.nthreads = threads; j.reps = reps; j.pin = pin;
    j.bar = &bar; j.encSeconds = encSeconds; j.decSeconds = decSeconds; j.bad = &bad; j.gate = &gate;
    jobs[t] = j;
    if (pthread_create(&th[t], NULL, bench_worker, &jobs[t]) != 0) { rc = -1; break; }
    ++started;
  }
  pthread_mutex_lock(&gate.mu);
  gate.state = rc == 0 ? 1 : 2; /* all threads exist: go; otherwise the started ones leave without touching the barrier */
  pthread_cond_broadcast(&gate.cv);
  pthread_mutex_unlock(&gate.mu);
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&bar);
  pthread_cond_destroy(&gate.cv);
  pthread_mutex_destroy(&gate.mu);
  if (mismatches) *mismatches = bad;
  free(comp); free(out); free(th); free(jobs);
  /* back to glibc's documented defaults: this is somebody's Python process, not ours */
  mallopt(M_MMAP_THRESHOLD, 128 * 1024);
  mallopt(M_TRIM_THRESHOLD, 128 * 1024);
  return rc;
}
