// dietgpu::ansEncodeBatch* / ansDecodeBatch* / ansGetCompressedInfo* with the
// reference's C++ signatures (dietgpu/ans/GpuANSCodec.h:16-341, cudaStream_t ->
// hipStream_t), implemented inline on top of the C ABI of ../dietgpu_amd.h.
// Temp memory: the free part of the caller's StackDeviceMemory is handed to the
// C ABI as a raw region for the duration of the call; the bytes the call used are
// recorded in the stack's high-water mark.
#pragma once

#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../dietgpu_amd.h"
#include "StackDeviceMemory.h"

namespace dietgpu {

constexpr int kANSRequiredAlignment = DGPU_ANS_REQUIRED_ALIGNMENT;
constexpr int kANSDefaultProbBits = DGPU_ANS_DEFAULT_PROB_BITS;

inline uint32_t getMaxCompressedSize(uint32_t uncompressedBytes) {
  const uint32_t r = dgpu_ans_max_compressed_size(uncompressedBytes);
  if (r == 0) {  // CHECK_LE(rawSize, INT32_MAX), GpuANSEncode.cu:22
    fprintf(stderr, "getMaxCompressedSize(%u): exceeds INT32_MAX\n", uncompressedBytes);
    abort();
  }
  return r;
}

struct ANSCodecConfig {
  inline ANSCodecConfig() : probBits(kANSDefaultProbBits), useChecksum(false) {}
  explicit inline ANSCodecConfig(int pb, bool checksum = false) : probBits(pb), useChecksum(checksum) {}
  int probBits;      // 9, 10 or 11
  bool useChecksum;
};

enum class ANSDecodeError : uint32_t { None = 0, ChecksumMismatch = 1 };

struct ANSDecodeStatus {
  inline ANSDecodeStatus() : error(ANSDecodeError::None) {}
  ANSDecodeError error;
  std::vector<std::pair<int, std::string>> errorInfo;
};

namespace detail {
// glog CHECK upstream; abort with the library's message here
inline void checkRc(int rc, const char* what) {
  if (rc != DGPU_OK && rc != DGPU_ERR_CHECKSUM_MISMATCH) {
    fprintf(stderr, "%s failed: %s\n", what, dgpu_last_error());
    abort();
  }
}
inline ANSDecodeStatus toStatus(int rc, int32_t errBatch) {
  ANSDecodeStatus s;
  if (rc == DGPU_ERR_CHECKSUM_MISMATCH) {
    s.error = ANSDecodeError::ChecksumMismatch;
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
// The free part of the caller's stack (at most `want` bytes, the call's upper
// bound) is lent to the C ABI for the duration of the call; what the call really
// needed (`used`, reported by the C ABI) is what counts towards
// getMaxMemoryUsage(), as upstream where every temporary is alloc()ed one by one.
struct TempRegion {
  TempRegion(StackDeviceMemory& r, hipStream_t, size_t want) : res(r), ptr(r.lendFree(want, &bytes)) {}
  ~TempRegion() { res.noteUsage(used); }
  TempRegion(const TempRegion&) = delete;
  TempRegion& operator=(const TempRegion&) = delete;
  StackDeviceMemory& res;
  size_t bytes = 0;
  void* ptr;
  size_t used = 0;
};
}  // namespace detail

inline void ansEncodeBatchStride(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void* in_dev,
    uint32_t inPerBatchSize, uint32_t inPerBatchStride, const uint32_t* histogram_dev, void* out_dev,
    uint32_t outPerBatchStride, uint32_t* outBatchSize_dev, hipStream_t stream) {
  detail::TempRegion t(res, stream, dgpu_ans_encode_temp_bytes(numInBatch, inPerBatchSize));
  detail::checkRc(dgpu_ans_encode_batch_stride(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum,
                                               numInBatch, in_dev, inPerBatchSize, inPerBatchStride, histogram_dev,
                                               out_dev, outPerBatchStride, outBatchSize_dev, stream),
                  "ansEncodeBatchStride");
}

inline void ansEncodeBatchPointer(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void** in,
    const uint32_t* inSize, const uint32_t* histogram_dev, void** out, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxSize = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxSize = std::max(maxSize, inSize[i]);
  detail::TempRegion t(res, stream, dgpu_ans_encode_temp_bytes(numInBatch, maxSize));
  detail::checkRc(dgpu_ans_encode_batch_pointer(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum,
                                                numInBatch, in, inSize, histogram_dev, out, outSize_dev, stream),
                  "ansEncodeBatchPointer");
}

inline void ansEncodeBatchSplitSize(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void* in_dev,
    const uint32_t* inSplitSizes, const uint32_t* histogram_dev, void* out_dev, uint32_t outStride,
    uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxSize = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxSize = std::max(maxSize, inSplitSizes[i]);
  detail::TempRegion t(res, stream, dgpu_ans_encode_temp_bytes(numInBatch, maxSize));
  detail::checkRc(dgpu_ans_encode_batch_split_size(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum,
                                                   numInBatch, in_dev, inSplitSizes, histogram_dev, out_dev, outStride,
                                                   outSize_dev, stream),
                  "ansEncodeBatchSplitSize");
}

inline ANSDecodeStatus ansDecodeBatchStride(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void* in_dev,
    uint32_t inPerBatchStride, void* out_dev, uint32_t outPerBatchStride, uint32_t outPerBatchCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream) {
  detail::TempRegion t(res, stream, dgpu_ans_decode_temp_bytes(numInBatch, outPerBatchCapacity, config.probBits));
  int32_t err = -1;
  int rc = dgpu_ans_decode_batch_stride(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum, numInBatch,
                                        in_dev, inPerBatchStride, out_dev, outPerBatchStride, outPerBatchCapacity,
                                        outSuccess_dev, outSize_dev, stream, &err);
  detail::checkRc(rc, "ansDecodeBatchStride");
  return detail::toStatus(rc, err);
}

inline ANSDecodeStatus ansDecodeBatchPointer(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void** in, void** out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxCap = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxCap = std::max(maxCap, outCapacity[i]);
  detail::TempRegion t(res, stream, dgpu_ans_decode_temp_bytes(numInBatch, maxCap, config.probBits));
  int32_t err = -1;
  int rc = dgpu_ans_decode_batch_pointer(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum,
                                         numInBatch, in, out, outCapacity, outSuccess_dev, outSize_dev, stream, &err);
  detail::checkRc(rc, "ansDecodeBatchPointer");
  return detail::toStatus(rc, err);
}

inline ANSDecodeStatus ansDecodeBatchSplitSize(
    StackDeviceMemory& res, const ANSCodecConfig& config, uint32_t numInBatch, const void** in, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream) {
  uint32_t maxCap = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) maxCap = std::max(maxCap, outSplitSizes[i]);
  detail::TempRegion t(res, stream, dgpu_ans_decode_temp_bytes(numInBatch, maxCap, config.probBits));
  int32_t err = -1;
  int rc = dgpu_ans_decode_batch_split_size(t.ptr, t.bytes, &t.used, config.probBits, config.useChecksum,
                                            numInBatch, in, out_dev, outSplitSizes, outSuccess_dev, outSize_dev,
                                            stream, &err);
  detail::checkRc(rc, "ansDecodeBatchSplitSize");
  return detail::toStatus(rc, err);
}

inline void ansGetCompressedInfo(
    StackDeviceMemory& res, const void** in, uint32_t numInBatch, uint32_t* outSizes_dev,
    uint32_t* outChecksum_dev, hipStream_t stream) {
  detail::TempRegion t(res, stream, (size_t)numInBatch * 8 + 256);
  detail::checkRc(dgpu_ans_get_compressed_info(t.ptr, t.bytes, in, numInBatch, outSizes_dev, outChecksum_dev, stream),
                  "ansGetCompressedInfo");
}

inline void ansGetCompressedInfoDevice(
    StackDeviceMemory&, const void** in_dev, uint32_t numInBatch, uint32_t* outSizes_dev,
    uint32_t* outChecksum_dev, hipStream_t stream) {
  detail::checkRc(dgpu_ans_get_compressed_info_device(in_dev, numInBatch, outSizes_dev, outChecksum_dev, stream),
                  "ansGetCompressedInfoDevice");
}

}  // namespace dietgpu
