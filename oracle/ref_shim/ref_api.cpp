// extern "C" surface of oracle/_ref/libdietgpu_ref.so: the REFERENCE's own public API
// (dietgpu/ans/GpuANSCodec.h:22-341, dietgpu/float/GpuFloatCodec.h:31-292), compiled from the
// reference's sources and executed on the CPU SIMT emulation of dgemu.h.
// TEST INFRASTRUCTURE ONLY: tests use it to pin oracle/dietgpu_oracle.c to the reference itself.
// "Device" memory is host memory here; every call is synchronous.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "dietgpu/ans/BatchProvider.cuh"
#include "dietgpu/ans/GpuANSCodec.h"
#include "dietgpu/ans/GpuANSStatistics.cuh"
#include "dietgpu/ans/GpuANSUtils.cuh"
#include "dietgpu/float/GpuFloatCodec.h"
#include "dietgpu/float/GpuFloatUtils.cuh"
#include "dietgpu/utils/StackDeviceMemory.h"

using namespace dietgpu;

namespace {
// A fresh, zero-filled temp stack per call (cudaMalloc is calloc here): the reference copies up to 7
// never-written pad words per block out of its scratch memory into the archive
// (GpuANSEncode.cuh:618-627); with zero-filled scratch they are zero, as in the oracle.
size_t stackBytes(uint32_t num, const uint32_t* sizes, uint32_t wordBytes) {
  size_t total = 0, mx = 0;
  for (uint32_t i = 0; i < num; ++i) {
    total += (size_t)sizes[i] * wordBytes;
    mx = std::max(mx, (size_t)sizes[i] * wordBytes);
  }
  // per element: exponent plane + per-block scratch (5248 B per 4 KiB) + tables, sized by the LARGEST element
  return (size_t)num * (2 * mx + 16384) + total + ((size_t)8 << 20);
}
}  // namespace

extern "C" {

const char* dgref_version(void) { return "facebookresearch/dietgpu reference sources on dgemu (CPU SIMT emulation)"; }

// ---- layout / size probes (host-side functions of the reference) -------------
uint32_t dgref_ans_max_compressed_size(uint32_t bytes) { return getMaxCompressedSize(bytes); }
uint32_t dgref_float_max_compressed_size(uint32_t ft, uint32_t n) { return getMaxFloatCompressedSize((FloatType)ft, n); }
uint32_t dgref_ans_compressed_overhead(uint32_t numBlocks) { return ANSCoalescedHeader::getCompressedOverhead(numBlocks); }
uint32_t dgref_sizeof_ans_header(void) { return (uint32_t)sizeof(ANSCoalescedHeader); }
uint32_t dgref_sizeof_float_header(void) { return (uint32_t)sizeof(GpuFloatHeader); }
uint32_t dgref_sizeof_warp_state(void) { return (uint32_t)sizeof(ANSWarpState); }
uint32_t dgref_float_uncomp_data_size(uint32_t ft, uint32_t n) {
  switch ((FloatType)ft) {
    case FloatType::kFloat16: return FloatTypeInfo<FloatType::kFloat16>::getUncompDataSize(n);
    case FloatType::kBFloat16: return FloatTypeInfo<FloatType::kBFloat16>::getUncompDataSize(n);
    case FloatType::kFloat32: return FloatTypeInfo<FloatType::kFloat32>::getUncompDataSize(n);
    default: return 0;
  }
}
// Field accessors of the reference's header struct applied to an archive: {magic word matches, magic word, numBlocks,
// totalUncompressedWords, totalCompressedWords, probBits, useChecksum, checksum, offset of the pdf table,
// offset of the warp states, offset of blockWords, offset of the block data, total compressed size}
void dgref_ans_header_fields(const void* archive, uint32_t out[13]) {
  const ANSCoalescedHeader* h = (const ANSCoalescedHeader*)archive;
  ANSCoalescedHeader* hm = (ANSCoalescedHeader*)archive;
  // no getter upstream: a header written by the reference's own setter is the comparand
  ANSCoalescedHeader expect;
  expect.setMagicAndVersion();
  uint32_t mine = 0, theirs = 0;
  memcpy(&mine, archive, 4);
  memcpy(&theirs, &expect, 4);
  out[0] = mine == theirs ? 1u : 0u;  // first word == what setMagicAndVersion() writes
  out[1] = theirs;
  out[2] = h->getNumBlocks();
  out[3] = h->getTotalUncompressedWords();
  out[4] = h->getTotalCompressedWords();
  out[5] = h->getProbBits();
  out[6] = h->getUseChecksum() ? 1 : 0;
  out[7] = h->getChecksum();
  out[8] = (uint32_t)((const uint8_t*)hm->getSymbolProbs() - (const uint8_t*)archive);
  out[9] = (uint32_t)((const uint8_t*)hm->getWarpStates() - (const uint8_t*)archive);
  out[10] = (uint32_t)((const uint8_t*)hm->getBlockWords(h->getNumBlocks()) - (const uint8_t*)archive);
  out[11] = (uint32_t)((const uint8_t*)hm->getBlockDataStart(h->getNumBlocks()) - (const uint8_t*)archive);
  out[12] = h->getTotalCompressedSize();
}

// ---- statistics building blocks -------------------------------------------------
// ansHistogramBatch (GpuANSStatistics.cuh:384-412) on one buffer of any alignment
void dgref_histogram(const void* in, uint32_t size, uint32_t counts[256]) {
  BatchProviderStride p((void*)in, size, size);
  ansHistogramBatch(1, p, counts, nullptr);
}
// ansCalcWeights / quantizeWeights / normalizeProbabilitiesFromHistogram (GpuANSStatistics.cuh:178-430):
// counts [num][256], totals [num] -> table [num][256] of {pdf, cdf, magic, shift} (16-byte aligned)
void dgref_normalize_batch(uint32_t num, int probBits, const uint32_t* totals, const uint32_t* counts, void* table) {
  std::vector<uint32_t> prefix(num, 0);
  BatchProviderSplitSize sizes(nullptr, totals, prefix.data(), 1);
  ansCalcWeights(num, probBits, sizes, counts, (uint4*)table, nullptr);
}

// ---- the codec itself ---------------------------------------------------------
// Batch of `num` inputs (pointer API).  out[i] must hold dgref_ans_max_compressed_size(inSize[i]) bytes.
void dgref_ans_encode_batch(int probBits, int useChecksum, uint32_t num, const void* const* in, const uint32_t* inSize,
                            void* const* out, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, inSize, 1));
  ansEncodeBatchPointer(res, cfg, num, (const void**)in, inSize, nullptr, (void**)out, outSize, nullptr);
}
// returns 0, or 1 + index of the first element whose checksum did not match
int dgref_ans_decode_batch(int probBits, int useChecksum, uint32_t num, const void* const* in, void* const* out,
                           const uint32_t* outCapacity, uint8_t* outSuccess, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, outCapacity, 1));
  auto st = ansDecodeBatchPointer(res, cfg, num, (const void**)in, (void**)out, outCapacity, outSuccess, outSize, nullptr);
  if (st.error == ANSDecodeError::ChecksumMismatch) return 1 + (st.errorInfo.empty() ? 0 : st.errorInfo[0].first);
  return 0;
}
// `aligned16`: FloatCodecConfig::is16ByteAligned -- what DietGpu.cpp passes for 16-byte aligned tensors; it selects
// the reference's vectorised SplitFloatAligned16 / JoinFloatAligned16 paths (GpuFloatCompress.cuh:85-278,
// GpuFloatDecompress.cuh:25-270).  The caller guarantees the alignment of every in / out pointer then.
void dgref_float_compress_batch(uint32_t ft, int probBits, int useChecksum, int aligned16, uint32_t num,
                                const void* const* in, const uint32_t* inSize, void* const* out, uint32_t* outSize) {
  FloatCodecConfig cfg((FloatType)ft, ANSCodecConfig(probBits, false), aligned16 != 0, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, inSize, 4));
  floatCompress(res, cfg, num, (const void**)in, inSize, (void**)out, outSize, nullptr);
}
int dgref_float_decompress_batch(uint32_t ft, int probBits, int useChecksum, int aligned16, uint32_t num,
                                 const void* const* in, void* const* out, const uint32_t* outCapacity,
                                 uint8_t* outSuccess, uint32_t* outSize) {
  FloatCodecConfig cfg((FloatType)ft, ANSCodecConfig(probBits, false), aligned16 != 0, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, outCapacity, 4));
  auto st = floatDecompress(res, cfg, num, (const void**)in, (void**)out, outCapacity, outSuccess, outSize, nullptr);
  if (st.error == FloatDecompressError::ChecksumMismatch) return 1 + (st.errorInfo.empty() ? 0 : st.errorInfo[0].first);
  return 0;
}

// ---- the other batch providers of the reference (BatchProvider.cuh:39-194) ------------------------------
// ansEncodeBatchStride / ansDecodeBatchStride (GpuANSEncode.cu:27-53, GpuANSDecode.cu:20-45)
void dgref_ans_encode_batch_stride(int probBits, int useChecksum, uint32_t num, const void* in, uint32_t inSize,
                                   uint32_t inStride, void* out, uint32_t outStride, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  std::vector<uint32_t> sizes(num, inSize);
  StackDeviceMemory res(0, stackBytes(num, sizes.data(), 1));
  ansEncodeBatchStride(res, cfg, num, in, inSize, inStride, nullptr, out, outStride, outSize, nullptr);
}
int dgref_ans_decode_batch_stride(int probBits, int useChecksum, uint32_t num, const void* in, uint32_t inStride,
                                  void* out, uint32_t outStride, uint32_t outCapacity, uint8_t* outSuccess,
                                  uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  std::vector<uint32_t> sizes(num, outCapacity);
  StackDeviceMemory res(0, stackBytes(num, sizes.data(), 1));
  auto st = ansDecodeBatchStride(res, cfg, num, in, inStride, out, outStride, outCapacity, outSuccess, outSize, nullptr);
  if (st.error == ANSDecodeError::ChecksumMismatch) return 1 + (st.errorInfo.empty() ? 0 : st.errorInfo[0].first);
  return 0;
}
// ansEncodeBatchSplitSize / ansDecodeBatchSplitSize (GpuANSEncode.cu:115-179, GpuANSDecode.cu:122-193)
void dgref_ans_encode_batch_split_size(int probBits, int useChecksum, uint32_t num, const void* in,
                                       const uint32_t* splitSizes, void* out, uint32_t outStride, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, splitSizes, 1));
  ansEncodeBatchSplitSize(res, cfg, num, in, splitSizes, nullptr, out, outStride, outSize, nullptr);
}
int dgref_ans_decode_batch_split_size(int probBits, int useChecksum, uint32_t num, const void* const* in, void* out,
                                      const uint32_t* splitSizes, uint8_t* outSuccess, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, splitSizes, 1));
  auto st = ansDecodeBatchSplitSize(res, cfg, num, (const void**)in, out, splitSizes, outSuccess, outSize, nullptr);
  if (st.error == ANSDecodeError::ChecksumMismatch) return 1 + (st.errorInfo.empty() ? 0 : st.errorInfo[0].first);
  return 0;
}
// floatCompressSplitSize / floatDecompressSplitSize (GpuFloatCompress.cu:103-159, GpuFloatDecompress.cu:117-179)
void dgref_float_compress_split_size(uint32_t ft, int probBits, int useChecksum, int aligned16, uint32_t num,
                                     const void* in, const uint32_t* splitSizes, void* out, uint32_t outStride,
                                     uint32_t* outSize) {
  FloatCodecConfig cfg((FloatType)ft, ANSCodecConfig(probBits, false), aligned16 != 0, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, splitSizes, 4));
  floatCompressSplitSize(res, cfg, num, in, splitSizes, out, outStride, outSize, nullptr);
}
int dgref_float_decompress_split_size(uint32_t ft, int probBits, int useChecksum, int aligned16, uint32_t num,
                                      const void* const* in, void* out, const uint32_t* splitSizes,
                                      uint8_t* outSuccess, uint32_t* outSize) {
  FloatCodecConfig cfg((FloatType)ft, ANSCodecConfig(probBits, false), aligned16 != 0, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, splitSizes, 4));
  auto st = floatDecompressSplitSize(res, cfg, num, (const void**)in, out, splitSizes, outSuccess, outSize, nullptr);
  if (st.error == FloatDecompressError::ChecksumMismatch) return 1 + (st.errorInfo.empty() ? 0 : st.errorInfo[0].first);
  return 0;
}

// ---- caller-supplied histograms (GpuANSCodec.h:65-164: `histogram_dev`, [num][256] u32, nullable; the encoder then
// skips ansHistogramBatch and normalises what it was given, GpuANSEncode.cuh:692-700) -------------------------------
void dgref_ans_encode_batch_hist(int probBits, int useChecksum, uint32_t num, const void* const* in, const uint32_t* inSize,
                                 const uint32_t* histogram, void* const* out, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, inSize, 1));
  ansEncodeBatchPointer(res, cfg, num, (const void**)in, inSize, histogram, (void**)out, outSize, nullptr);
}
void dgref_ans_encode_batch_stride_hist(int probBits, int useChecksum, uint32_t num, const void* in, uint32_t inSize,
                                        uint32_t inStride, const uint32_t* histogram, void* out, uint32_t outStride,
                                        uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  std::vector<uint32_t> sizes(num, inSize);
  StackDeviceMemory res(0, stackBytes(num, sizes.data(), 1));
  ansEncodeBatchStride(res, cfg, num, in, inSize, inStride, histogram, out, outStride, outSize, nullptr);
}
void dgref_ans_encode_batch_split_size_hist(int probBits, int useChecksum, uint32_t num, const void* in,
                                            const uint32_t* splitSizes, const uint32_t* histogram, void* out,
                                            uint32_t outStride, uint32_t* outSize) {
  ANSCodecConfig cfg(probBits, useChecksum != 0);
  StackDeviceMemory res(0, stackBytes(num, splitSizes, 1));
  ansEncodeBatchSplitSize(res, cfg, num, in, splitSizes, histogram, out, outStride, outSize, nullptr);
}

// ---- header info (GpuANSCodec.h:309-341, GpuFloatCodec.h:252-292); any output may be null ---------------------------
// (the reference ASSERTS getUseChecksum() when a checksum is asked for: only ask on archives made with checksums)
void dgref_ans_get_compressed_info(uint32_t num, const void* const* in, uint32_t* outSizes, uint32_t* outChecksum) {
  StackDeviceMemory res(0, (size_t)1 << 20);
  ansGetCompressedInfo(res, (const void**)in, num, outSizes, outChecksum, nullptr);
}
void dgref_ans_get_compressed_info_device(uint32_t num, const void* const* in_dev, uint32_t* outSizes, uint32_t* outChecksum) {
  StackDeviceMemory res(0, (size_t)1 << 20);
  ansGetCompressedInfoDevice(res, (const void**)in_dev, num, outSizes, outChecksum, nullptr);
}
void dgref_float_get_compressed_info(uint32_t num, const void* const* in, uint32_t* outSizes, uint32_t* outTypes,
                                     uint32_t* outChecksum) {
  StackDeviceMemory res(0, (size_t)1 << 20);
  floatGetCompressedInfo(res, (const void**)in, num, outSizes, outTypes, outChecksum, nullptr);
}
void dgref_float_get_compressed_info_device(uint32_t num, const void* const* in_dev, uint32_t* outSizes, uint32_t* outTypes,
                                            uint32_t* outChecksum) {
  StackDeviceMemory res(0, (size_t)1 << 20);
  floatGetCompressedInfoDevice(res, (const void**)in_dev, num, outSizes, outTypes, outChecksum, nullptr);
}

}  // extern "C"
