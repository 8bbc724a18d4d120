// torch.ops.dietgpu.* on PyTorch-ROCm: the ten ops of dietgpu/DietGpu.cpp with
// the schema strings kept verbatim (DietGpu.cpp:915-937), implemented on top of
// the C ABI of include/dietgpu_amd.h.  Host-only C++ (no kernels here); tensors
// are plumbing for device memory and the current HIP stream.
//
// Same argument validation and error behaviour as the reference
// (TORCH_CHECK -> c10::Error): DietGpu.cpp:149-275 (compress_data_res),
// :310-452 (split size), :454-522 (simple), :530-644 (decompress_data_res),
// :679-816, :818-911.
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <optional>
#include <tuple>
#include <vector>

#include "../../include/dietgpu_amd.h"

namespace dietgpu_amd {
namespace {

constexpr int kDefaultPrecision = 10;  // DietGpu.cpp:114
// The precision the six codec ops use on the calling thread.  The reference's ops fix it at kDefaultPrecision; its C++
// API takes it per call (ANSCodecConfig.probBits, GpuANSCodec.h:28-51).  dietgpu_amd::set_precision (an EXTRA op, in
// its own namespace) lets dietgpu_amd.ops serve `prob_bits` 9 / 11 through this library too instead of through ctypes.
thread_local int tPrecision = kDefaultPrecision;
int precision() { return tPrecision; }

uint32_t floatTypeFromDtype(at::ScalarType t) {
  switch (t) {
    case at::ScalarType::Half: return DGPU_FLOAT16;
    case at::ScalarType::BFloat16: return DGPU_BFLOAT16;
    case at::ScalarType::Float: return DGPU_FLOAT32;
    default:
      TORCH_CHECK(false, "dietgpu: tensor must be float16, bfloat16 or float32");
  }
  return DGPU_FLOAT_UNDEFINED;
}

at::ScalarType dtypeFromFloatType(uint32_t ft) {
  switch (ft) {
    case DGPU_FLOAT16: return at::ScalarType::Half;
    case DGPU_BFLOAT16: return at::ScalarType::BFloat16;
    case DGPU_FLOAT32: return at::ScalarType::Float;
    default:
      TORCH_CHECK(false, "dietgpu: unknown float type in archive");
  }
  return at::ScalarType::Half;
}

std::tuple<int64_t, int64_t> totalAndMaxSize(const std::vector<at::Tensor>& ts) {
  int64_t total = 0, mx = 0;
  for (auto& t : ts) {
    auto n = t.numel();
    TORCH_CHECK((uint64_t)(n * t.element_size()) <= std::numeric_limits<uint32_t>::max());
    total += n;
    mx = std::max<int64_t>(mx, n);
  }
  return {total, mx};
}

// the size queries return 0 beyond the reference's INT32_MAX guard (getMaxCompressedSize's CHECK_LE, GpuANSEncode.cu:22)
uint32_t maxAnsSize(uint64_t bytes) {
  TORCH_CHECK(bytes <= 0xffffffffull, "input of ", bytes, " bytes: sizes are 32-bit (GpuANSCodec.h)");
  const uint32_t r = dgpu_ans_max_compressed_size((uint32_t)bytes);
  TORCH_CHECK(r != 0, "input of ", bytes, " bytes: its maximum compressed size exceeds INT32_MAX (GpuANSEncode.cu:22)");
  return r;
}
uint32_t maxFloatSize(uint32_t ft, uint64_t words) {
  TORCH_CHECK(words <= 0xffffffffull, "tensor of ", words, " words: sizes are 32-bit (GpuFloatCodec.h)");
  const uint32_t r = dgpu_float_max_compressed_size(ft, (uint32_t)words);
  TORCH_CHECK(r != 0, "tensor of ", words, " words: the maximum compressed size exceeds INT32_MAX (GpuANSEncode.cu:22)");
  return r;
}

void* streamOf(int dev) { return (void*)c10::hip::getCurrentHIPStream(dev).stream(); }

void check(int rc, const char* what, bool isFloat) {
  // (DietGpu.cpp:626-633 / 801-808 raise this text; the members are what upstream's errorInfo lists)
  TORCH_CHECK(rc != DGPU_ERR_CHECKSUM_MISMATCH, isFloat ? "floatDecompress" : "ANSDecode",
              ": checksum mismatch seen on decoded data; archive cannot be unpacked\n", dgpu_last_error());
  TORCH_CHECK(rc == DGPU_OK, what, " failed: ", dgpu_last_error());
}

struct Temp {
  void* ptr = nullptr;
  size_t bytes = 0;
};
Temp tempOf(const std::optional<at::Tensor>& t, int dev) {
  Temp r;
  if (t) {
    TORCH_CHECK(t->device().is_cuda());
    TORCH_CHECK(t->is_contiguous());
    TORCH_CHECK(t->get_device() == dev);
    r.ptr = t->data_ptr();
    r.bytes = (size_t)t->numel() * t->element_size();
  }
  return r;
}

void validateCompOut(
    const std::optional<at::Tensor>& outCompressed, const std::optional<at::Tensor>& outSizes,
    int64_t rows, int64_t cols, int dev, const at::Device& device, at::Tensor& comp, at::Tensor& sizes) {
  if (outCompressed) {
    TORCH_CHECK(outCompressed->dtype() == at::kByte);
    TORCH_CHECK(outCompressed->device().is_cuda());
    TORCH_CHECK(outCompressed->is_contiguous());
    TORCH_CHECK(outCompressed->dim() == 2);
    TORCH_CHECK(outCompressed->size(0) >= rows);
    TORCH_CHECK(outCompressed->size(1) >= cols);
    TORCH_CHECK(outCompressed->get_device() == dev);
    comp = *outCompressed;
  } else {
    comp = at::empty({rows, cols}, at::TensorOptions().device(device).dtype(at::kByte));
  }
  if (outSizes) {
    TORCH_CHECK(outSizes->dtype() == at::kInt);
    TORCH_CHECK(outSizes->device().is_cuda());
    TORCH_CHECK(outSizes->dim() == 1);
    TORCH_CHECK(outSizes->is_contiguous());
    TORCH_CHECK(outSizes->size(0) >= rows);
    TORCH_CHECK(outSizes->get_device() == dev);
    sizes = *outSizes;
  } else {
    sizes = at::empty({rows}, at::TensorOptions().device(device).dtype(at::kInt));
  }
}

void validateStatus(const std::optional<at::Tensor>& outStatus, const std::optional<at::Tensor>& outSizes, int64_t n, int dev) {
  if (outStatus) {
    TORCH_CHECK(outStatus->is_contiguous());
    TORCH_CHECK(outStatus->device().is_cuda());
    TORCH_CHECK(outStatus->dtype() == at::kByte);
    TORCH_CHECK(outStatus->numel() == n);
    TORCH_CHECK(outStatus->get_device() == dev);
  }
  if (outSizes) {
    TORCH_CHECK(outSizes->is_contiguous());
    TORCH_CHECK(outSizes->device().is_cuda());
    TORCH_CHECK(outSizes->dtype() == at::kInt);
    TORCH_CHECK(outSizes->numel() == n);
    TORCH_CHECK(outSizes->get_device() == dev);
  }
}

}  // namespace

// ---- size queries (DietGpu.cpp:116-143) --------------------------------------
std::tuple<int64_t, int64_t> max_float_compressed_output_size(const std::vector<at::Tensor>& ts) {
  TORCH_CHECK(!ts.empty());
  auto sz = totalAndMaxSize(ts);
  return {(int64_t)ts.size(),
          (int64_t)maxFloatSize(floatTypeFromDtype(ts[0].scalar_type()), (uint64_t)std::get<1>(sz))};
}

int64_t max_float_compressed_size(const at::Tensor& dtype, int64_t size) {
  return maxFloatSize(floatTypeFromDtype(dtype.scalar_type()), (uint64_t)size);
}

std::tuple<int64_t, int64_t> max_any_compressed_output_size(const std::vector<at::Tensor>& ts) {
  TORCH_CHECK(!ts.empty());
  auto sz = totalAndMaxSize(ts);
  return {(int64_t)ts.size(), (int64_t)maxAnsSize((uint64_t)std::get<1>(sz) * ts[0].element_size())};
}

int64_t max_any_compressed_size(int64_t bytes) { return maxAnsSize((uint64_t)bytes); }

// ---- compress ------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor, int64_t> compress_data(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, bool checksum,
    const std::optional<at::Tensor>& tempMem, const std::optional<at::Tensor>& outCompressed,
    const std::optional<at::Tensor>& outCompressedSizes) {
  TORCH_CHECK(!tIns.empty());
  int dev = tIns.front().get_device();
  c10::hip::HIPGuard guard(dev);
  Temp tmp = tempOf(tempMem, dev);

  auto maxOut = compressAsFloat ? max_float_compressed_output_size(tIns) : max_any_compressed_output_size(tIns);
  for (auto& t : tIns) {
    TORCH_CHECK(t.device().is_cuda());
    TORCH_CHECK(t.is_contiguous());
    TORCH_CHECK(t.get_device() == dev);
    if (compressAsFloat) {
      TORCH_CHECK(t.dtype() == tIns[0].dtype());
      floatTypeFromDtype(t.scalar_type());
    }
  }
  at::Tensor comp, sizes;
  validateCompOut(outCompressed, outCompressedSizes, (int64_t)tIns.size(), std::get<1>(maxOut), dev, tIns[0].device(), comp, sizes);

  const size_t n = tIns.size();
  std::vector<const void*> inPtrs(n);
  std::vector<uint32_t> inSize(n);
  std::vector<void*> compPtrs(n);
  for (size_t i = 0; i < n; ++i) {
    inPtrs[i] = tIns[i].data_ptr();
    inSize[i] = compressAsFloat ? tIns[i].numel() : tIns[i].numel() * tIns[i].element_size();
    compPtrs[i] = (uint8_t*)comp.data_ptr() + i * comp.size(1);
  }
  size_t used = 0;
  if (compressAsFloat) {
    check(dgpu_float_compress(tmp.ptr, tmp.bytes, &used, floatTypeFromDtype(tIns[0].scalar_type()), precision(),
                              checksum, (uint32_t)n, inPtrs.data(), inSize.data(), compPtrs.data(),
                              (uint32_t*)sizes.data_ptr(), streamOf(dev)),
          "floatCompress", true);
  } else {
    check(dgpu_ans_encode_batch_pointer(tmp.ptr, tmp.bytes, &used, precision(), checksum, (uint32_t)n,
                                        inPtrs.data(), inSize.data(), nullptr, compPtrs.data(),
                                        (uint32_t*)sizes.data_ptr(), streamOf(dev)),
          "ansEncodeBatchPointer", false);
  }
  return {std::move(comp), std::move(sizes), (int64_t)used};
}

std::tuple<std::vector<at::Tensor>, at::Tensor, int64_t> compress_data_split_size(
    bool compressAsFloat, const at::Tensor& tIn, const at::Tensor& tSplitSizes, bool checksum,
    const std::optional<at::Tensor>& tempMem, const std::optional<at::Tensor>& outCompressed,
    const std::optional<at::Tensor>& outCompressedSizes) {
  int dev = tIn.get_device();
  c10::hip::HIPGuard guard(dev);
  Temp tmp = tempOf(tempMem, dev);

  TORCH_CHECK(tIn.device().is_cuda());
  TORCH_CHECK(tIn.is_contiguous());
  uint32_t ft = compressAsFloat ? floatTypeFromDtype(tIn.scalar_type()) : DGPU_FLOAT_UNDEFINED;
  if (!compressAsFloat) {
    TORCH_CHECK(uintptr_t(tIn.data_ptr()) % DGPU_ANS_REQUIRED_ALIGNMENT == 0,
                "All splits should start on a 16 byte boundary; start pointer is not aligned");
  }
  auto numInBatch = tSplitSizes.numel();
  TORCH_CHECK(tSplitSizes.is_contiguous());
  TORCH_CHECK(tSplitSizes.device().is_cpu());
  TORCH_CHECK(tSplitSizes.dtype() == at::kInt);
  uint32_t maxSize = 0;
  for (int64_t i = 0; i < numInBatch; ++i) {
    auto size = ((const int32_t*)tSplitSizes.data_ptr())[i];
    TORCH_CHECK(size > 0);
    maxSize = std::max((uint32_t)size, maxSize);
    if (!compressAsFloat && i != numInBatch - 1) {
      TORCH_CHECK(size % DGPU_ANS_REQUIRED_ALIGNMENT == 0,
                  "All splits should start on a 16 byte boundary; the size of an interior split is not a multiple of 16 bytes");
    }
  }
  int64_t maxCompressedBytes = compressAsFloat ? maxFloatSize(ft, maxSize) : maxAnsSize(maxSize);
  at::Tensor comp, sizes;
  validateCompOut(outCompressed, outCompressedSizes, numInBatch, maxCompressedBytes, dev, tIn.device(), comp, sizes);

  size_t used = 0;
  if (compressAsFloat) {
    check(dgpu_float_compress_split_size(tmp.ptr, tmp.bytes, &used, ft, precision(), checksum, (uint32_t)numInBatch,
                                         tIn.data_ptr(), (const uint32_t*)tSplitSizes.data_ptr(), comp.data_ptr(),
                                         (uint32_t)comp.size(1), (uint32_t*)sizes.data_ptr(), streamOf(dev)),
          "floatCompressSplitSize", true);
  } else {
    check(dgpu_ans_encode_batch_split_size(tmp.ptr, tmp.bytes, &used, precision(), checksum, (uint32_t)numInBatch,
                                           tIn.data_ptr(), (const uint32_t*)tSplitSizes.data_ptr(), nullptr,
                                           comp.data_ptr(), (uint32_t)comp.size(1), (uint32_t*)sizes.data_ptr(),
                                           streamOf(dev)),
          "ansEncodeBatchSplitSize", false);
  }
  // compressedMatrixToTensors (DietGpu.cpp:77-104): views narrowed to the reported sizes
  at::Tensor sizesHost = sizes.to(at::kCPU);
  auto flat = comp.view({comp.numel()});
  std::vector<at::Tensor> out(numInBatch);
  for (int64_t i = 0; i < numInBatch; ++i) {
    out[i] = flat.narrow(0, i * comp.size(1), ((const int32_t*)sizesHost.data_ptr())[i]);
  }
  return {std::move(out), std::move(sizes), (int64_t)used};
}

std::vector<at::Tensor> compress_data_simple(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, bool checksum, const std::optional<int64_t>& tempMem) {
  TORCH_CHECK(!tIns.empty());
  std::optional<at::Tensor> scratch;
  if (tempMem && *tempMem > 0) {
    scratch = at::empty({*tempMem}, at::TensorOptions().device(tIns[0].device()).dtype(at::kByte));
  }
  auto comp = compress_data(compressAsFloat, tIns, checksum, scratch, std::nullopt, std::nullopt);
  auto& mat = std::get<0>(comp);
  at::Tensor sizeHost = std::get<1>(comp).to(at::kCPU);
  std::vector<at::Tensor> out;
  for (size_t i = 0; i < tIns.size(); ++i) {
    auto n = ((const int32_t*)sizeHost.data_ptr())[i];
    out.emplace_back(mat[i].narrow(0, 0, n).clone());
  }
  return out;
}

// ---- decompress ----------------------------------------------------------------
int64_t decompress_data_impl(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, const std::vector<at::Tensor>& tOuts, bool checksum,
    Temp tmp, const std::optional<at::Tensor>& outStatus, const std::optional<at::Tensor>& outSizes) {
  TORCH_CHECK(!tIns.empty());
  TORCH_CHECK(tIns.size() == tOuts.size());
  int dev = tIns.front().get_device();
  c10::hip::HIPGuard guard(dev);
  const size_t n = tIns.size();
  std::vector<const void*> inPtrs(n);
  std::vector<void*> outPtrs(n);
  std::vector<uint32_t> outCapacity(n);
  std::vector<uint32_t> inBytes(n);  // the tensors say how many bytes each archive may occupy
  for (size_t i = 0; i < n; ++i) {
    auto& tIn = tIns[i];
    auto& tOut = tOuts[i];
    TORCH_CHECK(tIn.device().is_cuda());
    TORCH_CHECK(tIn.get_device() == dev);
    TORCH_CHECK(tIn.is_contiguous());
    TORCH_CHECK(tOut.device().is_cuda());
    TORCH_CHECK(tOut.get_device() == dev);
    TORCH_CHECK(tOut.is_contiguous());
    TORCH_CHECK(tIn.dtype() == at::kByte);
    if (compressAsFloat) floatTypeFromDtype(tOut.scalar_type());
    inPtrs[i] = tIn.data_ptr();
    inBytes[i] = (uint32_t)std::min<int64_t>(tIn.numel(), std::numeric_limits<uint32_t>::max());
    outPtrs[i] = tOut.data_ptr();
    auto cap = compressAsFloat ? tOut.numel() : tOut.numel() * tOut.element_size();
    TORCH_CHECK((uint64_t)cap <= std::numeric_limits<uint32_t>::max());
    outCapacity[i] = (uint32_t)cap;
  }
  validateStatus(outStatus, outSizes, (int64_t)n, dev);
  size_t used = 0;
  int32_t err = -1;
  if (compressAsFloat) {
    check(dgpu_float_decompress_bounded(tmp.ptr, tmp.bytes, &used, floatTypeFromDtype(tOuts[0].scalar_type()), precision(),
                                checksum, (uint32_t)n, inPtrs.data(), inBytes.data(), outPtrs.data(), outCapacity.data(),
                                outStatus ? (uint8_t*)outStatus->data_ptr() : nullptr,
                                outSizes ? (uint32_t*)outSizes->data_ptr() : nullptr, streamOf(dev), &err),
          "floatDecompress", true);
  } else {
    check(dgpu_ans_decode_batch_pointer_bounded(tmp.ptr, tmp.bytes, &used, precision(), checksum, (uint32_t)n,
                                        inPtrs.data(), inBytes.data(), outPtrs.data(), outCapacity.data(),
                                        outStatus ? (uint8_t*)outStatus->data_ptr() : nullptr,
                                        outSizes ? (uint32_t*)outSizes->data_ptr() : nullptr, streamOf(dev), &err),
          "ansDecodeBatchPointer", false);
  }
  return (int64_t)used;
}

int64_t decompress_data(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, const std::vector<at::Tensor>& tOuts, bool checksum,
    const std::optional<at::Tensor>& tempMem, const std::optional<at::Tensor>& outStatus,
    const std::optional<at::Tensor>& outSizes) {
  TORCH_CHECK(!tIns.empty());
  int dev = tIns.front().get_device();
  return decompress_data_impl(compressAsFloat, tIns, tOuts, checksum, tempOf(tempMem, dev), outStatus, outSizes);
}

int64_t decompress_data_split_size(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, at::Tensor& tOut, const at::Tensor& tSplitSizes,
    bool checksum, const std::optional<at::Tensor>& tempMem, const std::optional<at::Tensor>& outStatus,
    const std::optional<at::Tensor>& outSizes) {
  TORCH_CHECK(!tIns.empty());
  int dev = tIns.front().get_device();
  c10::hip::HIPGuard guard(dev);
  Temp tmp = tempOf(tempMem, dev);
  auto numInBatch = tSplitSizes.numel();
  TORCH_CHECK(tSplitSizes.is_contiguous());
  TORCH_CHECK(tSplitSizes.device().is_cpu());
  TORCH_CHECK(tSplitSizes.dtype() == at::kInt);
  TORCH_CHECK(numInBatch == (int64_t)tIns.size());
  std::vector<const void*> inPtrs(numInBatch);
  std::vector<uint32_t> splitSizes(numInBatch);
  std::vector<uint32_t> inBytes(numInBatch);
  for (int64_t i = 0; i < numInBatch; ++i) {
    auto& tIn = tIns[i];
    TORCH_CHECK(tIn.device().is_cuda());
    TORCH_CHECK(tIn.get_device() == dev);
    TORCH_CHECK(tIn.is_contiguous());
    TORCH_CHECK(tIn.dtype() == at::kByte);
    inPtrs[i] = tIn.data_ptr();
    inBytes[i] = (uint32_t)std::min<int64_t>(tIn.numel(), std::numeric_limits<uint32_t>::max());
    auto size = ((const int32_t*)tSplitSizes.data_ptr())[i];
    TORCH_CHECK(size > 0);
    splitSizes[i] = size;
  }
  TORCH_CHECK(tOut.device().is_cuda());
  TORCH_CHECK(tOut.get_device() == dev);
  TORCH_CHECK(tOut.is_contiguous());
  if (compressAsFloat) floatTypeFromDtype(tOut.scalar_type());
  validateStatus(outStatus, outSizes, numInBatch, dev);
  size_t used = 0;
  int32_t err = -1;
  if (compressAsFloat) {
    check(dgpu_float_decompress_split_size_bounded(tmp.ptr, tmp.bytes, &used, floatTypeFromDtype(tOut.scalar_type()),
                                           precision(), checksum, (uint32_t)numInBatch, inPtrs.data(), inBytes.data(),
                                           tOut.data_ptr(), splitSizes.data(),
                                           outStatus ? (uint8_t*)outStatus->data_ptr() : nullptr,
                                           outSizes ? (uint32_t*)outSizes->data_ptr() : nullptr, streamOf(dev), &err),
          "floatDecompressSplitSize", true);
  } else {
    check(dgpu_ans_decode_batch_split_size_bounded(tmp.ptr, tmp.bytes, &used, precision(), checksum, (uint32_t)numInBatch,
                                           inPtrs.data(), inBytes.data(), tOut.data_ptr(), splitSizes.data(),
                                           outStatus ? (uint8_t*)outStatus->data_ptr() : nullptr,
                                           outSizes ? (uint32_t*)outSizes->data_ptr() : nullptr, streamOf(dev), &err),
          "ansDecodeBatchSplitSize", false);
  }
  return (int64_t)used;
}

std::vector<at::Tensor> decompress_data_simple(
    bool compressAsFloat, const std::vector<at::Tensor>& tIns, bool checksum, const std::optional<int64_t>& tempMem) {
  TORCH_CHECK(!tIns.empty());
  int dev = tIns.front().get_device();
  c10::hip::HIPGuard guard(dev);
  const size_t n = tIns.size();
  at::Tensor scratch;
  Temp tmp;
  if (tempMem && *tempMem >= 256) {  // kSDMAlignment, DietGpu.cpp:831-834
    scratch = at::empty({*tempMem}, at::TensorOptions().device(tIns[0].device()).dtype(at::kByte));
    tmp.ptr = scratch.data_ptr();
    tmp.bytes = (size_t)*tempMem;
  }
  std::vector<const void*> inPtrs(n);
  for (size_t i = 0; i < n; ++i) {
    TORCH_CHECK(tIns[i].device().is_cuda());
    TORCH_CHECK(tIns[i].get_device() == dev);
    TORCH_CHECK(tIns[i].is_contiguous());
    inPtrs[i] = tIns[i].data_ptr();
  }
  auto opts = at::TensorOptions().device(tIns[0].device()).dtype(at::kInt);
  at::Tensor sizes = at::empty({(int64_t)n}, opts), types = at::zeros({(int64_t)n}, opts);
  if (compressAsFloat) {
    check(dgpu_float_get_compressed_info(tmp.ptr, tmp.bytes, inPtrs.data(), (uint32_t)n, (uint32_t*)sizes.data_ptr(),
                                         (uint32_t*)types.data_ptr(), nullptr, streamOf(dev)),
          "floatGetCompressedInfo", true);
  } else {
    check(dgpu_ans_get_compressed_info(tmp.ptr, tmp.bytes, inPtrs.data(), (uint32_t)n, (uint32_t*)sizes.data_ptr(),
                                       nullptr, streamOf(dev)),
          "ansGetCompressedInfo", false);
  }
  at::Tensor hs = sizes.to(at::kCPU), ht = types.to(at::kCPU);
  std::vector<at::Tensor> tOuts;
  for (size_t i = 0; i < n; ++i) {
    auto size = ((const int32_t*)hs.data_ptr())[i];
    auto type = ((const int32_t*)ht.data_ptr())[i];
    if (compressAsFloat) {
      TORCH_CHECK(type == ((const int32_t*)ht.data_ptr())[0]);  // must be a consistent dtype
      tOuts.emplace_back(at::empty({size}, at::TensorOptions().device(tIns[0].device()).dtype(dtypeFromFloatType(type))));
    } else {
      tOuts.emplace_back(at::empty({size}, at::TensorOptions().device(tIns[0].device()).dtype(at::kByte)));
    }
  }
  decompress_data_impl(compressAsFloat, tIns, tOuts, checksum, tmp, std::nullopt, std::nullopt);
  return tOuts;
}

void set_precision(int64_t probBits) {
  TORCH_CHECK(probBits == 9 || probBits == 10 || probBits == 11, "probBits must be 9, 10 or 11");
  tPrecision = (int)probBits;
}

}  // namespace dietgpu_amd

// Schema strings verbatim from DietGpu.cpp:915-937
TORCH_LIBRARY_FRAGMENT(dietgpu, m) {
  m.def("max_float_compressed_output_size(Tensor[] ts) -> (int, int)");
  m.def("max_float_compressed_size(Tensor dtype, int size) -> int");
  m.def("max_any_compressed_output_size(Tensor[] ts) -> (int, int)");
  m.def("max_any_compressed_size(int bytes) -> int");
  m.def(
      "compress_data(bool compress_as_float, Tensor[] ts_in, bool checksum=False, Tensor? temp_mem=None, Tensor? out_compressed=None, Tensor? out_compressed_bytes=None) -> (Tensor, Tensor, int)");
  m.def(
      "compress_data_split_size(bool compress_as_float, Tensor t_in, Tensor t_in_split_sizes, bool checksum=False, Tensor? temp_mem=None, Tensor? out_compressed=None, Tensor? out_compressed_bytes=None) -> (Tensor[], Tensor, int)");
  m.def(
      "compress_data_simple(bool compress_as_float, Tensor[] ts_in, bool checksum=False, int? temp_mem=67108864) -> Tensor[]");
  m.def(
      "decompress_data(bool compress_as_float, Tensor[] ts_in, Tensor[] ts_out, bool checksum=False, Tensor? temp_mem=None, Tensor? out_status=None, Tensor? out_decompressed_words=None) -> (int)");
  m.def(
      "decompress_data_split_size(bool compress_as_float, Tensor[] ts_in, Tensor t_out, Tensor t_out_split_sizes, bool checksum=False, Tensor? temp_mem=None, Tensor? out_status=None, Tensor? out_decompressed_words=None) -> (int)");
  m.def(
      "decompress_data_simple(bool compress_as_float, Tensor[] ts_in, bool checksum=False, int? temp_mem=67108864) -> Tensor[]");
}

// The version of the C ABI this file was compiled against, and whether the ops below were registered.  Plain C symbols:
// dietgpu_amd/ops.py compares them through ctypes after loading the library.  Nothing here throws -- torch.ops.load_library
// is a dlopen, and a C++ exception leaving a static initialiser under dlopen ends in std::terminate, not in Python.
extern "C" uint32_t dgpu_torch_built_abi(void) { return DGPU_ABI_VERSION; }
static bool gOpsRegistered = false;
extern "C" int dgpu_torch_ops_registered(void) { return gOpsRegistered ? 1 : 0; }
// a libdietgpu_amd.so from another build of the sources than this file was compiled against: register nothing
static bool coreLibraryMatches() { return dgpu_abi_version() == DGPU_ABI_VERSION; }

TORCH_LIBRARY(dietgpu_amd, m) {
  if (!coreLibraryMatches()) return;
  m.def("set_precision(int prob_bits) -> ()", &dietgpu_amd::set_precision);
}

TORCH_LIBRARY(dietgpu, m) {
  if (!coreLibraryMatches()) return;  // (the schemas above stay without implementations; ops.py does not use them)
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::max_float_compressed_output_size"), TORCH_FN(dietgpu_amd::max_float_compressed_output_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::max_float_compressed_size"), TORCH_FN(dietgpu_amd::max_float_compressed_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::max_any_compressed_output_size"), TORCH_FN(dietgpu_amd::max_any_compressed_output_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::max_any_compressed_size"), TORCH_FN(dietgpu_amd::max_any_compressed_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::compress_data"), TORCH_FN(dietgpu_amd::compress_data));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::compress_data_split_size"), TORCH_FN(dietgpu_amd::compress_data_split_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::compress_data_simple"), TORCH_FN(dietgpu_amd::compress_data_simple));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::decompress_data"), TORCH_FN(dietgpu_amd::decompress_data));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::decompress_data_split_size"), TORCH_FN(dietgpu_amd::decompress_data_split_size));
  m.impl(TORCH_SELECTIVE_NAME("dietgpu::decompress_data_simple"), TORCH_FN(dietgpu_amd::decompress_data_simple));
  gOpsRegistered = true;
}
