// Host-side device helpers with the surface of dietgpu/utils/DeviceUtils.h:55-216 -- the subset every caller of
// the codec API touches (the reference's gtests, DietGpu.cpp:159,289): current-device queries, DeviceScope, and
// RAII wrappers for streams and events.  HIP-backed, header-only, host-only (plain C++17: no device code).
//
// Names: the classes are HipStream / HipEvent; `CudaStream` / `CudaEvent` are the spellings the reference's callers
// use for the same two classes and are provided as type aliases, so that switching the includes is all a caller
// written against dietgpu/utils/DeviceUtils.h has to do.  (Aliases of these two class names only: the HIP runtime
// is called directly and no CUDA runtime symbol is emulated.)
//
// Not carried over: the profiler start/stop and unified-memory probes (unused by the codec path), and the glog
// CHECK machinery -- errors abort with a message on stderr, like StackDeviceMemory.h here.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace dietgpu {

#define DIETGPU_HIP_CHECK(X)                                                                                         \
  do {                                                                                                               \
    hipError_t dgpuErr__ = (X);                                                                                      \
    if (dgpuErr__ != hipSuccess) {                                                                                   \
      fprintf(stderr, "HIP error %s: %s at %s:%d\n", hipGetErrorName(dgpuErr__), hipGetErrorString(dgpuErr__),       \
              __FILE__, __LINE__);                                                                                   \
      abort();                                                                                                       \
    }                                                                                                                \
  } while (0)

inline std::string errorToString(hipError_t err) { return hipGetErrorString(err); }
inline std::string errorToName(hipError_t err) { return hipGetErrorName(err); }

inline int getCurrentDevice() {
  int dev = -1;
  DIETGPU_HIP_CHECK(hipGetDevice(&dev));
  return dev;
}
inline void setCurrentDevice(int device) { DIETGPU_HIP_CHECK(hipSetDevice(device)); }
inline int getNumDevices() {
  int n = 0;
  const hipError_t err = hipGetDeviceCount(&n);
  if (err == hipErrorNoDevice) return 0;  // (DeviceUtils.cpp:38-48 treats "no device" as zero devices)
  DIETGPU_HIP_CHECK(err);
  return n;
}

// Switches the current device for the lifetime of the object (-1: leaves it alone).
class DeviceScope {
 public:
  explicit DeviceScope(int device) {
    if (device >= 0) {
      const int cur = getCurrentDevice();
      if (cur != device) {
        prev_ = cur;
        setCurrentDevice(device);
      }
    }
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
  ~DeviceScope() {
    if (prev_ >= 0) setCurrentDevice(prev_);
  }

 private:
  int prev_ = -1;
};

inline void synchronizeAllDevices() {
  const int n = getNumDevices();
  for (int d = 0; d < n; ++d) {
    DeviceScope scope(d);
    DIETGPU_HIP_CHECK(hipDeviceSynchronize());
  }
}

// Cached device properties (one query per device and process; guarded, as DeviceUtils.cpp:64-80).
inline const hipDeviceProp_t& getDeviceProperties(int device) {
  static std::mutex mu;
  static std::unordered_map<int, hipDeviceProp_t> cache;
  std::lock_guard<std::mutex> guard(mu);
  auto it = cache.find(device);
  if (it == cache.end()) {
    hipDeviceProp_t prop;
    DIETGPU_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    it = cache.emplace(device, prop).first;
  }
  return it->second;
}
inline const hipDeviceProp_t& getCurrentDeviceProperties() { return getDeviceProperties(getCurrentDevice()); }
inline int getMaxThreads(int device) { return getDeviceProperties(device).maxThreadsPerBlock; }
inline int getMaxThreadsCurrentDevice() { return getMaxThreads(getCurrentDevice()); }
inline size_t getMaxSharedMemPerBlock(int device) { return getDeviceProperties(device).sharedMemPerBlock; }
inline size_t getMaxSharedMemPerBlockCurrentDevice() { return getMaxSharedMemPerBlock(getCurrentDevice()); }

// Device that owns `p`, or -1 for host / unregistered memory.
inline int getDeviceForAddress(const void* p) {
  if (!p) return -1;
  hipPointerAttribute_t attr;
  const hipError_t err = hipPointerGetAttributes(&attr, p);
  if (err != hipSuccess) {
    (void)hipGetLastError();  // plain host memory is "invalid value", not a failure
    return -1;
  }
  return attr.type == hipMemoryTypeDevice ? attr.device : -1;
}

// An event recorded on a stream at construction.
class HipEvent {
 public:
  explicit HipEvent(hipStream_t stream, bool timer = false) {
    DIETGPU_HIP_CHECK(hipEventCreateWithFlags(&event_, timer ? hipEventDefault : hipEventDisableTiming));
    DIETGPU_HIP_CHECK(hipEventRecord(event_, stream));
  }
  HipEvent(const HipEvent&) = delete;
  HipEvent& operator=(const HipEvent&) = delete;
  HipEvent(HipEvent&& other) noexcept : event_(std::exchange(other.event_, nullptr)) {}
  HipEvent& operator=(HipEvent&& other) noexcept {
    if (this != &other) {
      destroy();
      event_ = std::exchange(other.event_, nullptr);
    }
    return *this;
  }
  ~HipEvent() { destroy(); }

  hipEvent_t get() { return event_; }
  void streamWaitOnEvent(hipStream_t stream) { DIETGPU_HIP_CHECK(hipStreamWaitEvent(stream, event_, 0)); }
  void cpuWaitOnEvent() { DIETGPU_HIP_CHECK(hipEventSynchronize(event_)); }
  // milliseconds from `from` (both created with timer = true) to this event; waits for this one
  float timeFrom(HipEvent& from) {
    cpuWaitOnEvent();
    float ms = 0.f;
    DIETGPU_HIP_CHECK(hipEventElapsedTime(&ms, from.event_, event_));
    return ms;
  }

 private:
  void destroy() {
    if (event_) DIETGPU_HIP_CHECK(hipEventDestroy(event_));
    event_ = nullptr;
  }
  hipEvent_t event_ = nullptr;
};

// An owned stream on the current device.
class HipStream {
 public:
  explicit HipStream(unsigned flags = hipStreamDefault) { DIETGPU_HIP_CHECK(hipStreamCreateWithFlags(&stream_, flags)); }
  HipStream(const HipStream&) = delete;
  HipStream& operator=(const HipStream&) = delete;
  HipStream(HipStream&& other) noexcept : stream_(std::exchange(other.stream_, nullptr)) {}
  HipStream& operator=(HipStream&& other) noexcept {
    if (this != &other) {
      destroy();
      stream_ = std::exchange(other.stream_, nullptr);
    }
    return *this;
  }
  ~HipStream() { destroy(); }

  hipStream_t get() { return stream_; }
  operator hipStream_t() { return stream_; }
  static HipStream make() { return HipStream(hipStreamDefault); }
  static HipStream makeNonBlocking() { return HipStream(hipStreamNonBlocking); }

 private:
  void destroy() {
    if (stream_) DIETGPU_HIP_CHECK(hipStreamDestroy(stream_));
    stream_ = nullptr;
  }
  hipStream_t stream_ = nullptr;
};

// the reference's spellings of the two classes above (dietgpu/utils/DeviceUtils.h:127-186)
using CudaEvent = HipEvent;
using CudaStream = HipStream;

// Every stream of `waiting` waits for everything enqueued so far on every stream of `waitOn`.
template <typename L1, typename L2>
void streamWaitBase(const L1& waiting, const L2& waitOn) {
  std::vector<HipEvent> marks;
  for (hipStream_t s : waitOn) marks.emplace_back(s);
  for (hipStream_t s : waiting) {
    for (HipEvent& e : marks) e.streamWaitOnEvent(s);
  }
}
template <typename L1>
void streamWait(const L1& a, const std::initializer_list<hipStream_t>& b) {
  streamWaitBase(a, b);
}
template <typename L2>
void streamWait(const std::initializer_list<hipStream_t>& a, const L2& b) {
  streamWaitBase(a, b);
}
inline void streamWait(const std::initializer_list<hipStream_t>& a, const std::initializer_list<hipStream_t>& b) {
  streamWaitBase(a, b);
}

}  // namespace dietgpu
