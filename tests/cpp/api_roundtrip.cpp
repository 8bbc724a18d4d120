// Round trip through the dietgpu:: C++ API (include/dietgpu_amd/*.h) the way the
// reference's gtests drive it (ANSTest.cu:84-170, FloatTest.cu:122-232): raw
// device pointers, a StackDeviceMemory and a non-blocking stream.
// Build: hipcc -std=c++17 -Iinclude tests/cpp/api_roundtrip.cpp -Ldietgpu_amd/lib -ldietgpu_amd
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "dietgpu_amd/GpuANSCodec.h"
#include "dietgpu_amd/GpuFloatCodec.h"

using namespace dietgpu;

#define HIP(x) DIETGPU_HIP_VERIFY(x)

static int failures = 0;
#define EXPECT(c)                                           \
  do {                                                      \
    if (!(c)) {                                             \
      printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c);  \
      ++failures;                                           \
    }                                                       \
  } while (0)

template <typename T>
T* toDev(const std::vector<T>& v, hipStream_t s) {
  T* d = nullptr;
  HIP(hipMalloc((void**)&d, std::max<size_t>(v.size() * sizeof(T), 16)));
  HIP(hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return d;
}

int main() {
  hipStream_t stream;
  HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  auto res = makeStackMemory(64 << 20);

  // --- ANS, batch of 3, checksum on (ANSTest.cu: BatchPointer) ---
  std::mt19937 gen(10);
  std::exponential_distribution<float> dist(20.0f);
  std::vector<uint32_t> sizes = {10000, 10013, 4096};
  std::vector<std::vector<uint8_t>> data;
  std::vector<const void*> in;
  std::vector<void*> comp, out;
  std::vector<uint32_t> cap;
  for (auto n : sizes) {
    std::vector<uint8_t> v(n);
    for (auto& b : v) b = (uint8_t)(std::min(dist(gen), 1.0f) * 255.0f);
    in.push_back(toDev(v, stream));
    void* c; HIP(hipMalloc(&c, getMaxCompressedSize(n))); comp.push_back(c);
    void* o; HIP(hipMalloc(&o, n)); out.push_back(o);
    cap.push_back(n);
    data.push_back(std::move(v));
  }
  uint32_t* outSize_dev; HIP(hipMalloc((void**)&outSize_dev, 3 * 4));
  uint8_t* success_dev; HIP(hipMalloc((void**)&success_dev, 3));
  uint32_t* decSize_dev; HIP(hipMalloc((void**)&decSize_dev, 3 * 4));
  ANSCodecConfig cfg(10, true);
  ansEncodeBatchPointer(res, cfg, 3, in.data(), sizes.data(), nullptr, comp.data(), outSize_dev, stream);
  auto st = ansDecodeBatchPointer(res, cfg, 3, (const void**)comp.data(), out.data(), cap.data(), success_dev, decSize_dev, stream);
  EXPECT(st.error == ANSDecodeError::None);
  uint32_t hs[3]; uint8_t ok[3];
  HIP(hipMemcpyAsync(hs, outSize_dev, 12, hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(ok, success_dev, 3, hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  for (int i = 0; i < 3; ++i) {
    std::vector<uint8_t> back(sizes[i]);
    HIP(hipMemcpy(back.data(), out[i], sizes[i], hipMemcpyDeviceToHost));
    EXPECT(ok[i] == 1);
    EXPECT(hs[i] % 16 == 0);
    EXPECT(back == data[i]);
  }
  // the high-water mark is what the call needed, not the upper bound it was offered
  EXPECT(res.getMaxMemoryUsage() > 0);
  EXPECT(res.getMaxMemoryUsage() <= dgpu_ans_encode_temp_bytes(3, 10013));
  EXPECT(res.getSizeAvailable() == res.getSizeTotal());  // nothing stays reserved
  {
    // a stack that is too small: the call still succeeds (library-owned overflow memory, with a warning)
    auto tiny = makeStackMemory(1024);
    ansEncodeBatchPointer(tiny, cfg, 3, in.data(), sizes.data(), nullptr, comp.data(), outSize_dev, stream);
    uint32_t hs2[3];
    HIP(hipMemcpyAsync(hs2, outSize_dev, 12, hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < 3; ++i) EXPECT(hs2[i] == hs[i]);
    EXPECT(tiny.getMaxMemoryUsage() > 1024);
  }

  {
    // checksum mismatch in TWO of the three members: errorInfo lists both (GpuANSDecode.cuh:581-590), each with
    // the text accumulated so far, as upstream
    for (int i : {0, 2}) {
      uint32_t word;
      HIP(hipMemcpy(&word, (char*)comp[i] + 20, 4, hipMemcpyDeviceToHost));  // ANSCoalescedHeader::checksum
      word ^= 0x5a;
      HIP(hipMemcpy((char*)comp[i] + 20, &word, 4, hipMemcpyHostToDevice));
    }
    auto bad = ansDecodeBatchPointer(res, cfg, 3, (const void**)comp.data(), out.data(), cap.data(), success_dev, decSize_dev, stream);
    EXPECT(bad.error == ANSDecodeError::ChecksumMismatch);
    EXPECT(bad.errorInfo.size() == 2);
    if (bad.errorInfo.size() == 2) {
      EXPECT(bad.errorInfo[0].first == 0 && bad.errorInfo[1].first == 2);
      EXPECT(bad.errorInfo[0].second.find("batch member 0") != std::string::npos);
      EXPECT(bad.errorInfo[1].second.find("batch member 0") != std::string::npos &&
             bad.errorInfo[1].second.find("batch member 2") != std::string::npos);
    }
  }

  // --- float codec, bf16, batch of 2 (FloatTest.cu: Batch) ---
  std::normal_distribution<float> nd;
  std::vector<uint32_t> fsizes = {8192 + 5, 30000};
  std::vector<std::vector<uint16_t>> fdata;
  std::vector<const void*> fin;
  std::vector<void*> fcomp, fout;
  for (auto n : fsizes) {
    std::vector<uint16_t> v(n);
    for (auto& w : v) { float f = nd(gen); uint32_t x; memcpy(&x, &f, 4); w = (uint16_t)(x >> 16); }
    fin.push_back(toDev(v, stream));
    void* c; HIP(hipMalloc(&c, getMaxFloatCompressedSize(FloatType::kBFloat16, n))); fcomp.push_back(c);
    void* o; HIP(hipMalloc(&o, n * 2)); fout.push_back(o);
    fdata.push_back(std::move(v));
  }
  FloatCodecConfig fcfg(FloatType::kBFloat16, ANSCodecConfig(10, false), false, true);
  floatCompress(res, fcfg, 2, fin.data(), fsizes.data(), fcomp.data(), outSize_dev, stream);
  auto fst = floatDecompress(res, fcfg, 2, (const void**)fcomp.data(), fout.data(), fsizes.data(), success_dev, decSize_dev, stream);
  EXPECT(fst.error == FloatDecompressError::None);
  HIP(hipStreamSynchronize(stream));
  for (int i = 0; i < 2; ++i) {
    std::vector<uint16_t> back(fsizes[i]);
    HIP(hipMemcpy(back.data(), fout[i], fsizes[i] * 2, hipMemcpyDeviceToHost));
    EXPECT(back == fdata[i]);
  }
  printf(failures ? "api_roundtrip: %d FAILURES\n" : "api_roundtrip: OK\n", failures);
  return failures ? 1 : 0;
}
