// HIP-backed temporary-memory stack with the surface of dietgpu's
// StackDeviceMemory (dietgpu/utils/StackDeviceMemory.h:19-301): a LIFO bump
// allocator over a caller-supplied or self-allocated device slab with 256-byte
// granularity, hipMalloc fallback (with a stderr warning) when the slab
// overflows, RAII reservations released in reverse order, and a high-water mark.
// Host-only, header-only; not thread-safe (as upstream).
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

namespace dietgpu {

#define DIETGPU_HIP_VERIFY(X)                                                         \
  do {                                                                                \
    hipError_t err__ = (X);                                                           \
    if (err__ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)err__, hipGetErrorString(err__), __FILE__, __LINE__); \
      abort();                                                                        \
    }                                                                                 \
  } while (0)

constexpr size_t kDefaultStackSize = 256 * 1024 * 1024;  // StackDeviceMemory.h:20
constexpr size_t kSDMAlignment = 256;                     // StackDeviceMemory.h:22

class StackDeviceMemory;

enum class AllocType { Temporary, Permanent };

template <typename T>
struct GpuMemoryReservation {
  GpuMemoryReservation() = default;
  GpuMemoryReservation(StackDeviceMemory* r, int dev, hipStream_t str, void* p, size_t n, size_t szAlloc)
      : res(r), device(dev), stream(str), ptr(p), num(n), sizeAllocated(szAlloc) {}
  GpuMemoryReservation(GpuMemoryReservation&& m) noexcept { moveFrom(m); }
  GpuMemoryReservation& operator=(GpuMemoryReservation&& m) {
    release();
    moveFrom(m);
    return *this;
  }
  GpuMemoryReservation(const GpuMemoryReservation&) = delete;
  GpuMemoryReservation& operator=(const GpuMemoryReservation&) = delete;
  ~GpuMemoryReservation() { release(); }

  T* data() { return reinterpret_cast<T*>(ptr); }
  const T* data() const { return reinterpret_cast<const T*>(ptr); }

  // Unlike upstream (StackDeviceMemory.h:105-112, which reads the pageable
  // destination without waiting) this synchronises the stream before returning.
  std::vector<T> copyToHost(hipStream_t s) const {
    std::vector<T> out(num);
    DIETGPU_HIP_VERIFY(hipMemcpyAsync(out.data(), data(), num * sizeof(T), hipMemcpyDeviceToHost, s));
    DIETGPU_HIP_VERIFY(hipStreamSynchronize(s));
    return out;
  }

  void release();

  StackDeviceMemory* res = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  void* ptr = nullptr;
  size_t num = 0;
  size_t sizeAllocated = 0;

 private:
  void moveFrom(GpuMemoryReservation& m) {
    res = m.res; device = m.device; stream = m.stream; ptr = m.ptr; num = m.num; sizeAllocated = m.sizeAllocated;
    m.res = nullptr; m.ptr = nullptr; m.num = 0; m.sizeAllocated = 0;
  }
};

class StackDeviceMemory {
 public:
  // Allocate a new slab on `device` (StackDeviceMemory.cpp:28-52)
  StackDeviceMemory(int device, size_t allocPerDevice) : device_(device) {
    size_ = allocPerDevice ? std::max(allocPerDevice, kSDMAlignment) : 0;
    if (size_) {
      int prev = 0;
      DIETGPU_HIP_VERIFY(hipGetDevice(&prev));
      DIETGPU_HIP_VERIFY(hipSetDevice(device_));
      DIETGPU_HIP_VERIFY(hipMalloc((void**)&owned_, size_));
      DIETGPU_HIP_VERIFY(hipSetDevice(prev));
      start_ = owned_;
    }
    head_ = start_;
  }
  // Use memory the caller owns (StackDeviceMemory.cpp:54-75)
  StackDeviceMemory(int device, void* p, size_t size) : device_(device) {
    start_ = (char*)p;
    size_ = p ? size : 0;
    head_ = start_;
  }
  ~StackDeviceMemory() {
    if (owned_) {
      int prev = 0;
      (void)hipGetDevice(&prev);
      (void)hipSetDevice(device_);
      (void)hipFree(owned_);
      (void)hipSetDevice(prev);
    }
  }
  StackDeviceMemory(const StackDeviceMemory&) = delete;
  StackDeviceMemory& operator=(const StackDeviceMemory&) = delete;

  int getDevice() const { return device_; }

  template <typename T>
  GpuMemoryReservation<T> alloc(hipStream_t stream, size_t num, AllocType type = AllocType::Temporary) {
    size_t sizeToAlloc = (num * sizeof(T) + kSDMAlignment - 1) / kSDMAlignment * kSDMAlignment;
    sizeToAlloc = std::max(sizeToAlloc, kSDMAlignment);
    return GpuMemoryReservation<T>(this, device_, stream, allocPointer(stream, sizeToAlloc, type), num, sizeToAlloc);
  }

  template <typename T>
  GpuMemoryReservation<T> copyAlloc(hipStream_t stream, const T* ptr, size_t num, AllocType type = AllocType::Temporary) {
    auto mem = alloc<T>(stream, num, type);  // upstream over-allocates num * sizeof(T) ELEMENTS (:167-168)
    DIETGPU_HIP_VERIFY(hipMemcpyAsync(mem.data(), ptr, num * sizeof(T), hipMemcpyDefault, stream));
    return mem;
  }
  template <typename T>
  GpuMemoryReservation<T> copyAlloc(hipStream_t stream, const std::vector<T>& v, AllocType type = AllocType::Temporary) {
    return copyAlloc<T>(stream, v.data(), v.size(), type);
  }

  void* allocPointer(hipStream_t, size_t size, AllocType type) {
    if (type == AllocType::Permanent || size > getSizeAvailable()) {
      // overflow: hipMalloc, which synchronises (StackDeviceMemory.cpp:119-139)
      void* p = nullptr;
      DIETGPU_HIP_VERIFY(hipMalloc(&p, size));
      if (type == AllocType::Temporary) {
        fprintf(stderr,
                "WARNING: StackDeviceMemory: attempting to allocate %zu bytes with %zu bytes available; "
                "calling hipMalloc which is synchronizing. If possible, increase the temporary memory\n",
                size, getSizeAvailable());
      }
      overflow_[p] = size;
      overflowSize_ += size;
      maxSeen_ = std::max(maxSeen_, (size_t)(head_ - start_) + overflowSize_);
      return p;
    }
    void* out = head_;
    head_ += size;
    maxSeen_ = std::max(maxSeen_, (size_t)(head_ - start_) + overflowSize_);
    return out;
  }

  void deallocPointer(int, hipStream_t stream, size_t size, void* p) {
    auto it = overflow_.find(p);
    if (it != overflow_.end()) {
      DIETGPU_HIP_VERIFY(hipStreamSynchronize(stream));
      DIETGPU_HIP_VERIFY(hipFree(p));
      overflowSize_ -= it->second;
      overflow_.erase(it);
      return;
    }
    // allocations must be returned in reverse order (StackDeviceMemory.cpp:181)
    if ((char*)p + size != head_) {
      fprintf(stderr, "StackDeviceMemory: allocations must be freed in LIFO order\n");
      abort();
    }
    head_ = (char*)p;
  }

  size_t getSizeAvailable() const { return size_ - (size_t)(head_ - start_); }
  size_t getSizeTotal() const { return size_; }
  size_t getMaxMemoryUsage() const { return maxSeen_; }
  void resetMaxMemoryUsage() { maxSeen_ = 0; }
  // Used by the codec wrappers: the unused part of the stack (at most `want` bytes) as a raw
  // region for ONE call -- nothing is reserved, the caller must not alloc() until that call has
  // been enqueued (the upstream codec functions hold their reservations the same way) -- and
  // the bytes the call turned out to need, for the high-water mark.
  void* lendFree(size_t want, size_t* got) {
    *got = std::min(want, getSizeAvailable());
    return *got ? (void*)head_ : nullptr;
  }
  void noteUsage(size_t bytes) { maxSeen_ = std::max(maxSeen_, (size_t)(head_ - start_) + overflowSize_ + bytes); }

 private:
  int device_;
  char* owned_ = nullptr;
  char* start_ = nullptr;
  char* head_ = nullptr;
  size_t size_ = 0;
  size_t maxSeen_ = 0;
  size_t overflowSize_ = 0;
  std::unordered_map<void*, size_t> overflow_;
};

template <typename T>
void GpuMemoryReservation<T>::release() {
  if (ptr) {
    res->deallocPointer(device, stream, sizeAllocated, ptr);
    res = nullptr;
    ptr = nullptr;
    num = 0;
    sizeAllocated = 0;
  }
}

inline StackDeviceMemory makeStackMemory(size_t bytes = kDefaultStackSize) {
  int dev = 0;
  DIETGPU_HIP_VERIFY(hipGetDevice(&dev));
  return StackDeviceMemory(dev, bytes);
}

}  // namespace dietgpu
