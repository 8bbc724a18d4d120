// Round trip through the dietgpu:: C++ API (include/dietgpu_amd/*.h) the way the
// reference's gtests drive it (ANSTest.cu:84-170, FloatTest.cu:122-232): raw
// device pointers, a StackDeviceMemory and a non-blocking stream.
// Build: hipcc -std=c++17 -Iinclude tests/cpp/api_roundtrip.cpp -Ldietgpu_amd/lib -ldietgpu_amd
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "dietgpu_amd/DeviceUtils.h"
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

template <typename T>
std::vector<T> toHost(const T* d, size_t n, hipStream_t s) {
  std::vector<T> v(n);
  HIP(hipMemcpyAsync(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, s));
  HIP(hipStreamSynchronize(s));
  return v;
}

// The stride, split-size and info entry points of the C++ mirror, and `histogram_dev` (GpuANSCodec.h:65-164,
// 265-341, GpuFloatCodec.h:166-292), driven the way ANSTest.cu / FloatTest.cu drive them: stream and device helpers
// from DeviceUtils.h, one StackDeviceMemory for every call.
static void strideSplitSizeAndInfo(StackDeviceMemory& res) {
  DeviceScope scope(getCurrentDevice());
  auto stream = CudaStream::makeNonBlocking();  // (the reference's spelling; = HipStream)
  EXPECT(getNumDevices() >= 1 && getMaxThreadsCurrentDevice() >= 256);
  std::mt19937 gen(77);
  std::exponential_distribution<float> dist(30.0f);
  const uint32_t B = 4, n = 4096 * 3 + 8, stride = n + 20;  // rows 4-byte aligned, not 16
  std::vector<uint8_t> host(B * stride);
  for (auto& b : host) b = (uint8_t)(std::min(dist(gen), 1.0f) * 255.0f);
  uint8_t* in_dev = toDev(host, stream);
  EXPECT(getDeviceForAddress(in_dev) == getCurrentDevice() && getDeviceForAddress(host.data()) == -1);
  const uint32_t compStride = getMaxCompressedSize(n);
  uint8_t *comp_dev, *out_dev, *success_dev;
  uint32_t *size_dev, *decSize_dev, *hist_dev, *info_dev;
  HIP(hipMalloc((void**)&comp_dev, (size_t)B * compStride));
  HIP(hipMalloc((void**)&out_dev, (size_t)B * stride));
  HIP(hipMalloc((void**)&success_dev, B));
  HIP(hipMalloc((void**)&size_dev, B * 4));
  HIP(hipMalloc((void**)&decSize_dev, B * 4));
  HIP(hipMalloc((void**)&info_dev, 3 * B * 4));
  // histograms computed on the host, + 1 on every bin: a caller-supplied histogram that covers the data
  std::vector<uint32_t> hist(B * 256, 1);
  for (uint32_t b = 0; b < B; ++b)
    for (uint32_t i = 0; i < n; ++i) hist[b * 256 + host[b * stride + i]]++;
  hist_dev = toDev(hist, stream);

  ANSCodecConfig cfg(11, true);
  // stride batch without and with the caller's histogram (the second archive is the one decoded below)
  ansEncodeBatchStride(res, cfg, B, in_dev, n, stride, nullptr, comp_dev, compStride, size_dev, stream);
  auto sizesOwn = toHost(size_dev, B, stream);
  ansEncodeBatchStride(res, cfg, B, in_dev, n, stride, hist_dev, comp_dev, compStride, size_dev, stream);
  auto sizesGiven = toHost(size_dev, B, stream);
  for (uint32_t b = 0; b < B; ++b) EXPECT(sizesGiven[b] % 16 == 0 && sizesOwn[b] % 16 == 0 && sizesGiven[b] < n);
  HIP(hipMemsetAsync(out_dev, 0xcd, (size_t)B * stride, stream));
  auto st = ansDecodeBatchStride(res, cfg, B, comp_dev, compStride, out_dev, stride, n, success_dev, decSize_dev, stream);
  EXPECT(st.error == ANSDecodeError::None);
  {
    auto back = toHost(out_dev, (size_t)B * stride, stream);
    auto ok = toHost(success_dev, B, stream);
    auto dsz = toHost(decSize_dev, B, stream);
    for (uint32_t b = 0; b < B; ++b) {
      EXPECT(ok[b] == 1 && dsz[b] == n);
      EXPECT(memcmp(&back[b * stride], &host[b * stride], n) == 0);
      EXPECT(back[b * stride + n] == 0xcd);  // nothing beyond the element
    }
  }
  // info: host array of device pointers, then the same array on the device
  {
    std::vector<const void*> arch(B);
    for (uint32_t b = 0; b < B; ++b) arch[b] = comp_dev + (size_t)b * compStride;
    ansGetCompressedInfo(res, arch.data(), B, info_dev, info_dev + B, stream);
    auto got = toHost(info_dev, 2 * B, stream);
    const void** arch_dev = (const void**)toDev(arch, stream);
    HIP(hipMemsetAsync(info_dev, 0, 2 * B * 4, stream));
    ansGetCompressedInfoDevice(res, arch_dev, B, info_dev, info_dev + B, stream);
    auto got2 = toHost(info_dev, 2 * B, stream);
    for (uint32_t b = 0; b < B; ++b) {
      uint32_t ck = 0;
      for (uint32_t i = 0; i < n; ++i) ck ^= host[b * stride + i];
      EXPECT(got[b] == n && got[B + b] == ck && got2[b] == n && got2[B + b] == ck);
    }
    HIP(hipFree((void*)arch_dev));
  }
  // split sizes: ONE input buffer, elements at the prefix sums (interior sizes multiples of 4), caller's histogram
  {
    std::vector<uint32_t> split = {4096 * 2, 100, 0, 4096 + 4};
    std::vector<uint32_t> h2(B * 256, 0);
    uint32_t pos = 0, total = 0;
    for (uint32_t b = 0; b < B; ++b) {
      for (uint32_t i = 0; i < split[b]; ++i) h2[b * 256 + host[pos + i]]++;
      pos += split[b];
    }
    total = pos;
    uint32_t* h2_dev = toDev(h2, stream);
    ansEncodeBatchSplitSize(res, cfg, B, in_dev, split.data(), h2_dev, comp_dev, compStride, size_dev, stream);
    std::vector<const void*> arch(B);
    for (uint32_t b = 0; b < B; ++b) arch[b] = comp_dev + (size_t)b * compStride;
    HIP(hipMemsetAsync(out_dev, 0xcd, (size_t)B * stride, stream));
    auto st2 = ansDecodeBatchSplitSize(res, cfg, B, arch.data(), out_dev, split.data(), success_dev, decSize_dev, stream);
    EXPECT(st2.error == ANSDecodeError::None);
    auto back = toHost(out_dev, total + 1, stream);
    auto dsz = toHost(decSize_dev, B, stream);
    EXPECT(memcmp(back.data(), host.data(), total) == 0 && back[total] == 0xcd);
    for (uint32_t b = 0; b < B; ++b) EXPECT(dsz[b] == split[b]);
    HIP(hipFree(h2_dev));
  }
  // floats through the split-size pair + float info (sizes in float words)
  {
    std::normal_distribution<float> nd;
    std::vector<uint32_t> split = {4096 + 8, 16, 4096 * 2};
    uint32_t total = 0;
    for (auto v : split) total += v;
    std::vector<uint16_t> words(total);
    for (auto& w : words) { float f = nd(gen); uint32_t x; memcpy(&x, &f, 4); w = (uint16_t)(x >> 16); }
    uint16_t* w_dev = toDev(words, stream);
    const uint32_t fstride = getMaxFloatCompressedSize(FloatType::kBFloat16, 4096 * 2);
    uint8_t* fcomp;
    uint16_t* fout;
    HIP(hipMalloc((void**)&fcomp, (size_t)3 * fstride));
    HIP(hipMalloc((void**)&fout, (size_t)total * 2));
    FloatCodecConfig fcfg(FloatType::kBFloat16, ANSCodecConfig(10, false), false, true);
    floatCompressSplitSize(res, fcfg, 3, w_dev, split.data(), fcomp, fstride, size_dev, stream);
    std::vector<const void*> arch = {fcomp, fcomp + fstride, fcomp + 2 * (size_t)fstride};
    auto fst = floatDecompressSplitSize(res, fcfg, 3, arch.data(), fout, split.data(), success_dev, decSize_dev, stream);
    EXPECT(fst.error == FloatDecompressError::None);
    EXPECT(toHost(fout, total, stream) == words);
    floatGetCompressedInfo(res, arch.data(), 3, info_dev, info_dev + 3, info_dev + 6, stream);
    auto got = toHost(info_dev, 9, stream);
    const void** arch_dev = (const void**)toDev(arch, stream);
    HIP(hipMemsetAsync(info_dev, 0, 9 * 4, stream));
    floatGetCompressedInfoDevice(res, arch_dev, 3, info_dev, info_dev + 3, info_dev + 6, stream);
    auto got2 = toHost(info_dev, 9, stream);
    for (int i = 0; i < 3; ++i) {
      EXPECT(got[i] == split[i] && got[3 + i] == (uint32_t)FloatType::kBFloat16);
      EXPECT(got2[i] == got[i] && got2[3 + i] == got[3 + i] && got2[6 + i] == got[6 + i]);
    }
    HIP(hipFree((void*)arch_dev)); HIP(hipFree(w_dev)); HIP(hipFree(fcomp)); HIP(hipFree(fout));
  }
  // events: a timed pair around a call, and one stream waiting on another (DeviceUtils.h)
  {
    CudaEvent t0(stream, true);
    ansEncodeBatchStride(res, cfg, B, in_dev, n, stride, nullptr, comp_dev, compStride, size_dev, stream);
    CudaEvent t1(stream, true);
    EXPECT(t1.timeFrom(t0) > 0.f);
    auto other = HipStream::make();
    streamWait({other.get()}, {stream.get()});
    HIP(hipStreamSynchronize(other));
  }
  EXPECT(res.getSizeAvailable() == res.getSizeTotal());
  HIP(hipFree(in_dev)); HIP(hipFree(comp_dev)); HIP(hipFree(out_dev)); HIP(hipFree(success_dev));
  HIP(hipFree(size_dev)); HIP(hipFree(decSize_dev)); HIP(hipFree(hist_dev)); HIP(hipFree(info_dev));
}

int main() {
  hipStream_t stream;
  HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  auto res = makeStackMemory(64 << 20);
  {
    auto res2 = makeStackMemory(64 << 20);  // (its own stack: the high-water mark of `res` is checked below)
    strideSplitSizeAndInfo(res2);
  }

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
