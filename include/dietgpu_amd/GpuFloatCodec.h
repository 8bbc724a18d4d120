// dietgpu::floatCompress* / floatDecompress* / floatGetCompressedInfo* with the
// reference's C++ signatures (dietgpu/float/GpuFloatCodec.h:18-292), inline on
// top of the C ABI of ../dietgpu_amd.h.
#pragma once

#include "GpuANSCodec.h"

namespace dietgpu {

enum class FloatType : uint32_t { kUndefined = 0, kFloat16 = 1, kBFloat16 = 2, kFloat32 = 3 };

inline uint32_t getMaxFloatCompressedSize(FloatType floatType, uint32_t size) {
  const uint32_t r = dgpu_float_max_compressed_size((uint32_t)floatType, size);
  if (r == 0) {  // getMaxCompressedSize's CHECK_LE(rawSize, INT32_MAX), GpuANSEncode.cu:22
    fprintf(stderr, "getMaxFloatCompressedSize(%u): exceeds INT32_MAX\n", size);
    abort();
  }
  return r;
}

struct FloatCodecConfig {
  inline FloatCodecConfig() : floatType(FloatType::kFloat16), useChecksum(false), is16ByteAligned(false) {}
  inline FloatCodecConfig(FloatType ft, const ANSCodecConfig& ansConf, bool align, bool checksum = false)
      : floatType(ft), useChecksum(checksum), ansConfig(ansConf), is16ByteAligned(align) {}
  FloatType floatType;
  bool useChecksum;          // float-level checksum; ansConfig.useChecksum must stay false (GpuFloatCodec.h:50)
  ANSCodecConfig ansConfig;
  bool is16ByteAligned;      // accepted for source compatibility; alignment is detected per call
};
using FloatCompressConfig = FloatCodecConfig;
using FloatDecompressConfig = FloatCodecConfig;

enum class FloatDecompressError : uint32_t { None = 0, ChecksumMismatch = 1 };

struct FloatDecompressStatus {
  inline FloatDecompressStatus() : error(FloatDecompressError::None) {}
  FloatDecompressError error;
  std::vector<std::pair<int, std::string>> errorInfo;
};

namespace detail {
inline FloatDecompressStatus toFloatStatus(int rc, int32_t errBatch) {
  FloatDecompressStatus s;
  if (rc == DGPU_ERR_CHECKSUM_MISMATCH) {
    s.error = FloatDecompressError::ChecksumMismatch;
    // every mismatching member, each with the text accumulated so far -- upstream's stringstream is never reset
    // (GpuANSDecode.cuh:579-590)
    (void)errBatch;
    const uint32_t n = dgpu_last_checksum_mismatches(nullptr, nullptr, nullptr, 0);
    std::vector<int32_t> idx(n);
    std::vector<uint32_t> want(n), got(n);
    dgpu_last_checksum_mismatches(idx.data(), want.data(), got.data(), n);
    std::string text;
    for (uint32_t i = 0; i < n; ++i) {
      char buf[160];
      snprintf(buf, sizeof(buf), "Checksum mismatch in batch member %d: expected checksum %x got %x\n", (int)idx[i], want[i], got[i]);
      text += buf;
      s.errorInfo.emplace_back((int)idx[i], text);
    }
  }
  return s;
}
}  // namespace detail

inline void floatCompress(
    StackDeviceMemory& res, const FloatCompressConfig& config, uint32_t numInBatch, const void** in,
    const uint32_t* inSize, void** out, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxSize = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxSize = std::max(maxSize, inSize[i]);
  detail::TempRegion t(res, stream, dgpu_float_compress_temp_bytes((uint32_t)config.floatType, numInBatch, maxSize));
  detail::checkRc(dgpu_float_compress(t.ptr, t.bytes, &t.used, (uint32_t)config.floatType,
                                      config.ansConfig.probBits, config.useChecksum, numInBatch, in, inSize, out,
                                      outSize_dev, stream),
                  "floatCompress");
}

inline void floatCompressSplitSize(
    StackDeviceMemory& res, const FloatCompressConfig& config, uint32_t numInBatch, const void* in_dev,
    const uint32_t* inSplitSizes, void* out_dev, uint32_t outStride, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxSize = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxSize = std::max(maxSize, inSplitSizes[i]);
  detail::TempRegion t(res, stream, dgpu_float_compress_temp_bytes((uint32_t)config.floatType, numInBatch, maxSize));
  detail::checkRc(dgpu_float_compress_split_size(t.ptr, t.bytes, &t.used, (uint32_t)config.floatType,
                                                 config.ansConfig.probBits, config.useChecksum, numInBatch, in_dev,
                                                 inSplitSizes, out_dev, outStride, outSize_dev, stream),
                  "floatCompressSplitSize");
}

inline FloatDecompressStatus floatDecompress(
    StackDeviceMemory& res, const FloatDecompressConfig& config, uint32_t numInBatch, const void** in, void** out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxCap = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxCap = std::max(maxCap, outCapacity[i]);
  detail::TempRegion t(res, stream, dgpu_float_decompress_temp_bytes((uint32_t)config.floatType, numInBatch, maxCap,
                                                                     config.ansConfig.probBits));
  int32_t err = -1;
  int rc = dgpu_float_decompress(t.ptr, t.bytes, &t.used, (uint32_t)config.floatType, config.ansConfig.probBits,
                                 config.useChecksum, numInBatch, in, out, outCapacity, outSuccess_dev, outSize_dev,
                                 stream, &err);
  detail::checkRc(rc, "floatDecompress");
  return detail::toFloatStatus(rc, err);
}

inline FloatDecompressStatus floatDecompressSplitSize(
    StackDeviceMemory& res, const FloatDecompressConfig& config, uint32_t numInBatch, const void** in, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxCap = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxCap = std::max(maxCap, outSplitSizes[i]);
  detail::TempRegion t(res, stream, dgpu_float_decompress_temp_bytes((uint32_t)config.floatType, numInBatch, maxCap,
                                                                     config.ansConfig.probBits));
  int32_t err = -1;
  int rc = dgpu_float_decompress_split_size(t.ptr, t.bytes, &t.used, (uint32_t)config.floatType,
                                            config.ansConfig.probBits, config.useChecksum, numInBatch, in, out_dev,
                                            outSplitSizes, outSuccess_dev, outSize_dev, stream, &err);
  detail::checkRc(rc, "floatDecompressSplitSize");
  return detail::toFloatStatus(rc, err);
}

inline void floatGetCompressedInfo(
    StackDeviceMemory& res, const void** in, uint32_t numInBatch, uint32_t* outSizes_dev, uint32_t* outTypes_dev,
    uint32_t* outChecksum_dev, hipStream_t stream) {
  detail::TempRegion t(res, stream, (size_t)numInBatch * 8 + 256);
  detail::checkRc(dgpu_float_get_compressed_info(t.ptr, t.bytes, in, numInBatch, outSizes_dev, outTypes_dev,
                                                 outChecksum_dev, stream),
                  "floatGetCompressedInfo");
}

inline void floatGetCompressedInfoDevice(
    StackDeviceMemory&, const void** in_dev, uint32_t numInBatch, uint32_t* outSizes_dev, uint32_t* outTypes_dev,
    uint32_t* outChecksum_dev, hipStream_t stream) {
  detail::checkRc(dgpu_float_get_compressed_info_device(in_dev, numInBatch, outSizes_dev, outTypes_dev,
                                                        outChecksum_dev, stream),
                  "floatGetCompressedInfoDevice");
}

}  // namespace dietgpu
