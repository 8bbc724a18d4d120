/*
 * dietgpu_oracle.h -- CPU oracle for the dietgpu rANS / float codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.
 *
 * What this is: a sequential, plain-C restatement of the algorithm that the
 * reference (facebookresearch/dietgpu, CUDA-only) implements in its kernels.
 * Every function cites the reference file:line it follows.
 *
 * PARITY PINNING STATUS: "parity unpinned" for compressed BYTES.
 *   - The reference cannot be built here (CUDA + PTX + CUB + glog, no nvcc),
 *     and it ships no golden bitstreams, so compressed-byte parity with the
 *     reference is not pinned by any upstream vector.
 *   - What IS pinned (tests/test_oracle_known_answers.py): the reference's own
 *     known-answer tests for the normalised pdf (ANSStatisticsTest.cu:127-207),
 *     struct sizes / header layout (GpuANSUtils.cuh:229, GpuFloatUtils.cuh:74),
 *     size formulas (GpuANSEncode.cu:13-25, GpuFloatCompress.cu:23-45), the
 *     round-trip identity on the reference's deterministic generators
 *     (ANSTest.cu:18-31, FloatTest.cu:110-120), compressed size % 16 == 0
 *     (ANSTest.cu:131-135), empty-input archive (ans_test.py:68-77).
 *
 * Bytes that the reference leaves indeterminate (uninitialised header words,
 * per-block pad words, odd blockWords pad entry, non-comp tail padding) are
 * written as ZERO here; see SURVEY.md section 7.1.
 */
#ifndef DIETGPU_ORACLE_H
#define DIETGPU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FloatType, dietgpu/float/GpuFloatCodec.h:21-26 */
enum {
  DGO_FLOAT_UNDEFINED = 0,
  DGO_FLOAT16 = 1,
  DGO_BFLOAT16 = 2,
  DGO_FLOAT32 = 3,
};

/* ---- sizes ---- */
/* getMaxCompressedSize, dietgpu/ans/GpuANSEncode.cu:13-25 */
uint32_t dgo_ans_max_compressed_size(uint32_t uncompressedBytes);
/* getMaxFloatCompressedSize, dietgpu/float/GpuFloatCompress.cu:23-45 */
uint32_t dgo_float_max_compressed_size(uint32_t floatType, uint32_t numFloats);
/* ANSCoalescedHeader::getCompressedOverhead, dietgpu/ans/GpuANSUtils.cuh:68-82 */
uint32_t dgo_ans_compressed_overhead(uint32_t numBlocks);
/* FloatTypeInfo<FT>::getUncompDataSize, dietgpu/float/GpuFloatUtils.cuh:123-203 */
uint32_t dgo_float_uncomp_data_size(uint32_t floatType, uint32_t numFloats);

/* ---- statistics ---- */
/* histogramSingle, dietgpu/ans/GpuANSStatistics.cuh:21-134 */
void dgo_histogram(const uint8_t* in, uint32_t size, uint32_t counts[256]);
/* normalizeProbabilitiesFromHistogram, dietgpu/ans/GpuANSStatistics.cuh:178-367
 * table is [256][4] = {pdf, cdf, magic, shift}. total == 0 leaves table zeroed
 * (reference leaves it unwritten). Entries with pdf == 0 get magic = shift = 0
 * (reference computes an undefined value that is never looked up). */
void dgo_normalize(
    const uint32_t counts[256],
    uint32_t total,
    int probBits,
    uint32_t table[256 * 4]);

/* checksumSingle, dietgpu/ans/GpuChecksum.cuh:26-93 : XOR of all bytes,
 * folded to 8 bits */
uint32_t dgo_checksum(const uint8_t* in, uint32_t size);

/* ---- ANS block level ---- */
/* ansEncodeWarpBlock, dietgpu/ans/GpuANSEncode.cuh:141-211.
 * Returns the number of u16 words written to outWords (capacity 4096 words is
 * always enough); final 32 lane states go to outState. */
uint32_t dgo_ans_encode_block(
    const uint8_t* in,
    uint32_t n,
    int probBits,
    const uint32_t table[256 * 4],
    uint16_t* outWords,
    uint32_t outState[32]);

/* ansDecodeWarpBlock, dietgpu/ans/GpuANSDecode.cuh:161-217 (+:274-297).
 * lut is the 2^probBits packed decode table. Returns 0 on success, nonzero if
 * the stream did not consume exactly to position 0 / start state. */
int dgo_ans_decode_block(
    const uint16_t* words,
    uint32_t numWords,
    const uint32_t state[32],
    uint32_t n,
    int probBits,
    const uint32_t* lut,
    uint8_t* out);

/* ansDecodeTable, dietgpu/ans/GpuANSDecode.cuh:405-476 */
void dgo_ans_decode_table(const uint16_t pdf[256], int probBits, uint32_t* lut);

/* ---- ANS archive level ---- */
/* ansEncodeBatchDevice for one batch element,
 * dietgpu/ans/GpuANSEncode.cuh:674-849 (+ coalesce :515-628).
 * counts may be NULL (histogram computed from the data).
 * Returns compressed size in bytes (always a multiple of 16). */
uint32_t dgo_ans_encode(
    const uint8_t* in,
    uint32_t size,
    int probBits,
    int useChecksum,
    const uint32_t* counts,
    uint8_t* out);

/* ansDecodeKernel for one batch element, dietgpu/ans/GpuANSDecode.cuh:299-403.
 * Returns 0 = ok, 1 = capacity too small (nothing written), <0 = malformed.
 * *outSize receives the uncompressed size recorded in the header. */
int dgo_ans_decode(
    const uint8_t* in,
    int probBits,
    uint8_t* out,
    uint32_t outCapacity,
    uint32_t* outSize);

/* ansGetCompressedInfoKernel, dietgpu/ans/GpuANSInfo.cuh:17-37 */
int dgo_ans_info(
    const uint8_t* in,
    uint32_t* uncompressedSize,
    uint32_t* compressedSize,
    uint32_t* checksum,
    uint32_t* probBits);

/* ---- float codec ---- */
/* FloatTypeInfo<FT>::split / join, dietgpu/float/GpuFloatUtils.cuh:100-204 */
void dgo_float_split(
    uint32_t floatType,
    const void* in,
    uint32_t n,
    uint8_t* comp,
    uint8_t* nonComp /* getUncompDataSize bytes, tail zeroed */);
void dgo_float_join(
    uint32_t floatType,
    const uint8_t* comp,
    const uint8_t* nonComp,
    uint32_t n,
    void* out);

/* floatCompressDevice for one batch element,
 * dietgpu/float/GpuFloatCompress.cuh:446-579. Returns compressed bytes. */
uint32_t dgo_float_compress(
    uint32_t floatType,
    const void* in,
    uint32_t numFloats,
    int probBits,
    int useChecksum,
    uint8_t* out);

/* floatDecompressDevice for one batch element,
 * dietgpu/float/GpuFloatDecompress.cuh:565-738.
 * Returns 0 ok, 1 capacity too small, <0 malformed. *outSize in float words. */
int dgo_float_decompress(
    uint32_t floatType,
    const uint8_t* in,
    int probBits,
    void* out,
    uint32_t outCapacityFloats,
    uint32_t* outSize);

/* floatGetCompressedInfoKernel, dietgpu/float/GpuFloatInfo.cuh:18-41 */
int dgo_float_info(
    const uint8_t* in,
    uint32_t* numFloats,
    uint32_t* floatType,
    uint32_t* checksum,
    uint32_t* compressedSize);

/* ---- batch helpers used by bench.py's cpu_baseline leg (pthreads) ---- */
/* Encodes/decodes `batch` independent rows with `threads` worker threads.
 * in: [batch][inStride] bytes; out: [batch][outStride]. */
void dgo_ans_encode_batch(
    const uint8_t* in, uint32_t size, size_t inStride, uint32_t batch,
    int probBits, uint8_t* out, size_t outStride, uint32_t* outSizes,
    int threads);
void dgo_ans_decode_batch(
    const uint8_t* in, size_t inStride, uint32_t batch, int probBits,
    uint8_t* out, size_t outStride, uint32_t outCapacity, int threads);
void dgo_float_compress_batch(
    uint32_t floatType, const void* in, uint32_t numFloats, size_t inStride,
    uint32_t batch, int probBits, uint8_t* out, size_t outStride,
    uint32_t* outSizes, int threads);
void dgo_float_decompress_batch(
    uint32_t floatType, const uint8_t* in, size_t inStride, uint32_t batch,
    int probBits, void* out, size_t outStride, uint32_t outCapacityFloats,
    int threads);

#ifdef __cplusplus
}
#endif
#endif
