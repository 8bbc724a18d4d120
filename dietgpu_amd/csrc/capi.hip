// Host side of the C ABI (include/dietgpu_amd.h): argument checking, temp
// memory carving, one pinned-memory parameter upload per call, and the kernel
// launch sequences.  Everything is enqueued on the caller's stream; the only
// host synchronisation is checksum verification on decode (as upstream,
// GpuANSDecode.cuh:557-591) and the overflow-allocation fallback.
#include "../../include/dietgpu_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "format.h"
#include "kernels_decode.h"
#include "kernels_encode.h"
#include "kernels_pairs.h"
#include "kernels_stats.h"

using namespace dgpu;

namespace {

thread_local std::string g_lastError;
// every batch member whose checksum did not match in this thread's last decode call (GpuANSDecode.cuh:581-590)
struct ChecksumMismatch {
  int32_t batch;
  uint32_t expected, got;
};
thread_local std::vector<ChecksumMismatch> g_mismatches;
thread_local const char* g_captureHint = nullptr;  // why a call cannot be captured into a HIP graph (set next to the error)

int fail(int code, const std::string& msg) {
  g_lastError = msg;
  return code;
}

#define DGPU_HIP(expr)                                                              \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      return fail(DGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                               \
  } while (0)

#define DGPU_REQUIRE(cond, msg)                                      \
  do {                                                               \
    if (!(cond)) return fail(DGPU_ERR_INVALID_ARGUMENT, (msg));      \
  } while (0)

constexpr size_t kTempAlign = 256;  // kSDMAlignment, StackDeviceMemory.h:22

size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (used by
// bench.py for the roofline figure; off by default, zero cost when off).
struct ProfSpan {
  const char* name;
  hipEvent_t start, stop;
};
struct ProfState {
  std::mutex mu;
  bool enabled = false;
  std::vector<ProfSpan> open;
  std::map<std::string, std::pair<uint64_t, double>> acc;  // name -> (launches, total ms)
};
ProfState& prof() {
  static ProfState* p = new ProfState();
  return *p;
}

class KernelTimer {
 public:
  KernelTimer(const char* name, hipStream_t stream) : stream_(stream) {
    ProfState& p = prof();
    if (!p.enabled) return;
    span_.name = name;
    // no system-scope fence at the events: it would write the L2 back around every kernel that is being timed
    if (hipEventCreateWithFlags(&span_.start, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&span_.stop, hipEventDisableSystemFence) != hipSuccess) return;
    (void)hipEventRecord(span_.start, stream_);
    active_ = true;
  }
  ~KernelTimer() {
    if (!active_) return;
    (void)hipEventRecord(span_.stop, stream_);
    ProfState& p = prof();
    std::lock_guard<std::mutex> g(p.mu);
    p.open.push_back(span_);
  }

 private:
  hipStream_t stream_;
  ProfSpan span_{};
  bool active_ = false;
};

#define DGPU_LAUNCH(name, stream, ...)   \
  do {                                   \
    KernelTimer kt_(name, stream);       \
    hipLaunchKernelGGL(__VA_ARGS__);     \
  } while (0)

// ---------------------------------------------------------------------------
// Library-owned device state per (device, stream): everything a call needs that
// must outlive it or be zero when it starts.
//   * overflow slab: when the caller's temp memory is missing or too small the
//     reference falls back to cudaMalloc + cudaFree around every call
//     (synchronising, StackDeviceMemory.cpp:119-139).  Here the overflow comes
//     from a grow-only slab that is kept between calls: calls on one stream
//     execute in order, so every call can carve the slab from offset 0 without
//     synchronising.  A slab that has to grow in the middle of a call is RETIRED,
//     never freed under the call: allocations handed out earlier in the same call
//     live in it and their kernels have not even been launched yet.  Retired slabs
//     are freed by the NEXT call that needs overflow memory, after a stream
//     synchronise.
//   * arrival counters / accumulate-mode histogram counters of the histogram ->
//     normalisation hand-off, zero at rest.
// A call that touches this state holds `busy` from its first use until it has
// enqueued everything: two host threads enqueueing on the same stream would
// otherwise carve the same slab bytes for two calls whose kernels interleave.
// The registry is bounded: beyond kMaxStreams entries per process the idle ones
// are released after a device synchronise (streams come and go in long-running
// processes; their handles cannot be observed dying).
constexpr uint32_t kAccElements = 64;  // batches up to this size may use the accumulate-by-atomics histogram
struct StreamState {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex busy;          // held by the call using this state
  uint64_t lastUse = 0;
  // overflow slab
  uint8_t* slab = nullptr;
  size_t slabCap = 0;
  std::vector<void*> retired;
  // slabs a CAPTURED call carved memory from: a HIP graph replays into them, so they outlive their retirement (until
  // dgpu_release_graph_state); slabInGraph: the current slab is such a slab
  bool slabInGraph = false;
  std::vector<void*> graphSlabs;
  // 65536 arrival counters + kAccElements x 256 histogram counters + 16384 spill-pool flags (zero at rest)
  uint32_t* counters = nullptr;
  // a call was CAPTURED into a HIP graph with pointers into this state (slab, counters): the graph replays without
  // passing through the library, so the state is never trimmed or released implicitly (dgpu_release_graph_state)
  bool graphPinned = false;

  void releaseDeviceMemory() {
    for (void* p : retired) (void)hipFree(p);
    retired.clear();
    for (void* p : graphSlabs) (void)hipFree(p);
    graphSlabs.clear();
    slabInGraph = false;
    if (slab) (void)hipFree(slab);
    if (counters) (void)hipFree(counters);
    slab = nullptr;
    slabCap = 0;
    counters = nullptr;
  }
};

class StreamRegistry {
 public:
  static constexpr size_t kMaxStreams = 32;
  // Returns the state of (current device, stream) with its `busy` mutex HELD, creating the state if necessary.
  // `busy` is taken (try_lock) while the registry mutex is held, so trimLocked() / release() -- which only drop
  // states whose `busy` they can take -- can never delete a state between its lookup and its use.  A state that is
  // busy (another host thread is enqueueing on the same stream) is waited for with the registry mutex released.
  StreamState* acquire(hipStream_t stream, hipError_t* err) {
    for (;;) {
      {
        std::lock_guard<std::mutex> g(mu_);
        int dev = 0;
        *err = hipGetDevice(&dev);
        if (*err != hipSuccess) return nullptr;
        auto key = std::make_pair(dev, stream);
        auto it = states_.find(key);
        if (it == states_.end()) {
          if (states_.size() >= kMaxStreams) trimLocked();
          StreamState* s = new StreamState();
          s->device = dev;
          s->stream = stream;
          it = states_.emplace(key, s).first;
        }
        if (it->second->busy.try_lock()) {
          it->second->lastUse = ++clock_;
          return it->second;
        }
      }
      std::this_thread::yield();
    }
  }
  // Frees the device memory kept for `stream` on the current device (all streams
  // of all devices if `all`).  Synchronises first.  Returns the number of states released.
  // States captured into a HIP graph (graphPinned) are only released when `includeGraphPinned`.
  int release(hipStream_t stream, bool all, bool includeGraphPinned = false) {
    std::lock_guard<std::mutex> g(mu_);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::vector<std::map<std::pair<int, hipStream_t>, StreamState*>::iterator> victims;
    for (auto it = states_.begin(); it != states_.end(); ++it) {
      StreamState* s = it->second;
      const bool match = all || (it->first.first == dev && it->first.second == stream);
      if (match && (includeGraphPinned || !s->graphPinned) && s->busy.try_lock()) victims.push_back(it);
    }
    return dropLocked(victims, dev);
  }
  size_t size() {
    std::lock_guard<std::mutex> g(mu_);
    return states_.size();
  }

 private:
  typedef std::map<std::pair<int, hipStream_t>, StreamState*>::iterator Iter;
  // `victims` hold their `busy`.  ONE device synchronise per device (nothing enqueued earlier can still be using
  // the memory afterwards); a device that cannot be synchronised keeps its memory (leaked, never freed under work).
  int dropLocked(const std::vector<Iter>& victims, int restoreDev) {
    std::map<int, bool> synced;
    for (Iter it : victims) {
      const int d = it->second->device;
      if (!synced.count(d)) synced[d] = hipSetDevice(d) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    int n = 0;
    for (Iter it : victims) {
      StreamState* s = it->second;
      if (synced[s->device] && hipSetDevice(s->device) == hipSuccess) {
        s->releaseDeviceMemory();
      } else {
        (void)hipGetLastError();
        s->slab = nullptr;  // cannot prove idleness: leak rather than free under running kernels
        s->counters = nullptr;
        s->retired.clear();
        s->graphSlabs.clear();
      }
      s->busy.unlock();
      delete s;
      states_.erase(it);
      ++n;
    }
    (void)hipSetDevice(restoreDev);
    return n;
  }
  // Drops the least recently used half of the idle states that no HIP graph refers to.
  void trimLocked() {
    std::vector<std::pair<uint64_t, Iter>> idle;
    for (auto it = states_.begin(); it != states_.end(); ++it) {
      if (!it->second->graphPinned) idle.push_back({it->second->lastUse, it});
    }
    std::sort(idle.begin(), idle.end(), [](const std::pair<uint64_t, Iter>& a, const std::pair<uint64_t, Iter>& b) { return a.first < b.first; });
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::vector<Iter> victims;
    for (auto& e : idle) {
      if (victims.size() >= kMaxStreams / 2) break;
      if (e.second->second->busy.try_lock()) victims.push_back(e.second);
    }
    dropLocked(victims, cur);
  }
  std::mutex mu_;
  uint64_t clock_ = 0;
  std::map<std::pair<int, hipStream_t>, StreamState*> states_;
};

StreamRegistry& streamRegistry() {
  static StreamRegistry* r = new StreamRegistry();  // intentionally leaked: no teardown-order issues
  return *r;
}

// One call's hold on the stream state (taken lazily: calls whose temp memory
// suffices and that need no library-owned counters never touch it).
class StreamLease {
 public:
  explicit StreamLease(hipStream_t stream) : stream_(stream) {}
  StreamLease(const StreamLease&) = delete;
  StreamLease& operator=(const StreamLease&) = delete;
  ~StreamLease() {
    if (state_) state_->busy.unlock();
  }
  StreamState* state(hipError_t* err) {
    if (!state_) {
      StreamState* s = streamRegistry().acquire(stream_, err);  // returns with `busy` held
      if (!s) return nullptr;
      state_ = s;
      if (capturing()) s->graphPinned = true;
    }
    *err = hipSuccess;
    return state_;
  }
  hipStream_t stream() const { return stream_; }
  // Is the caller's stream being captured into a HIP graph?  Then nothing may synchronise or allocate, and every
  // library-owned address the kernels receive must stay valid for the life of the graph.
  bool capturing() {
    if (capturing_ < 0) {
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      capturing_ = (hipStreamIsCapturing(stream_, &st) == hipSuccess && st == hipStreamCaptureStatusActive) ? 1 : 0;
      if (capturing_ == 0) (void)hipGetLastError();
    }
    return capturing_ == 1;
  }

 private:
  hipStream_t stream_;
  StreamState* state_ = nullptr;
  int capturing_ = -1;
};

// Temp memory: bump allocation out of the caller's region; what does not fit
// comes from the stream's overflow slab with a warning (StackDeviceMemory.cpp:119-139).
class TempArena {
 public:
  TempArena(void* base, size_t bytes, StreamLease& lease)
      : base_((uint8_t*)base), bytes_(base ? bytes : 0), lease_(lease) {
    // honour the 256-byte granularity even if the caller's pointer is odd
    size_t mis = ((uintptr_t)base_) % kTempAlign;
    if (base_ && mis) {
      size_t skip = kTempAlign - mis;
      if (skip >= bytes_) {
        base_ = nullptr;
        bytes_ = 0;
      } else {
        base_ += skip;
        bytes_ -= skip;
      }
    }
  }

  template <typename T>
  T* alloc(size_t count, hipError_t* err) {
    size_t need = std::max(alignUp(count * sizeof(T), kTempAlign), kTempAlign);
    requested_ += need;
    if (head_ + need <= bytes_) {
      T* p = (T*)(base_ + head_);
      head_ += need;
      return p;
    }
    // no warning when the caller chose to pass no temp memory at all
    if (!warned_ && bytes_ > 0) {
      fprintf(stderr,
              "dietgpu_amd: WARNING: temp memory too small (%zu bytes given, > %zu needed); "
              "using library-owned overflow memory\n",
              bytes_, requested_);
      warned_ = true;
    }
    return (T*)overflow(need, err);
  }

  size_t requested() const { return requested_; }

 private:
  void* overflow(size_t need, hipError_t* err) {
    StreamState* s = lease_.state(err);
    if (!s) return nullptr;
    if (!overflowUsed_) {
      overflowUsed_ = true;
      // first overflow allocation of this call: slabs retired by EARLIER calls can go
      // (their kernels precede everything this call enqueues on the stream) -- not while the stream is being
      // captured: a synchronise would invalidate the capture, they wait for the next plain call.  Slabs that a
      // captured call carved memory from are not in this list: a graph replays with their address (graphSlabs).
      if (!s->retired.empty() && !lease_.capturing()) {
        *err = hipStreamSynchronize(lease_.stream());
        if (*err != hipSuccess) return nullptr;
        for (void* p : s->retired) (void)hipFree(p);
        s->retired.clear();
      }
    }
    if (overflowHead_ + need > s->slabCap) {
      if (lease_.capturing()) {
        // growing the slab means hipMalloc in the middle of a stream capture
        *err = hipErrorStreamCaptureUnsupported;
        g_captureHint = "temp memory: the library-owned overflow slab would have to grow while the stream is being "
                        "captured into a HIP graph; pass enough temp memory, or run the same call once before capturing";
        return nullptr;
      }
      const size_t cap = std::max<size_t>(std::max(2 * s->slabCap, need + (need >> 2)), (size_t)8 << 20);
      void* p = nullptr;
      *err = hipMalloc(&p, cap);
      if (*err != hipSuccess) return nullptr;
      // the old slab may hold allocations of THIS call: retire it, never free it here; one that a HIP graph replays
      // into stays until dgpu_release_graph_state() ("neither evicted nor trimmed nor released", dietgpu_amd.h)
      if (s->slab) (s->slabInGraph ? s->graphSlabs : s->retired).push_back(s->slab);
      s->slabInGraph = false;
      s->slab = (uint8_t*)p;
      s->slabCap = cap;
      overflowHead_ = 0;
    }
    if (lease_.capturing()) s->slabInGraph = true;
    void* out = s->slab + overflowHead_;
    overflowHead_ += need;
    return out;
  }

  uint8_t* base_;
  size_t bytes_;
  StreamLease& lease_;
  size_t head_ = 0;
  size_t overflowHead_ = 0;
  size_t requested_ = 0;
  bool warned_ = false;
  bool overflowUsed_ = false;
};

#define DGPU_ALLOC(var, T, arena, count)                                     \
  T* var = nullptr;                                                          \
  do {                                                                       \
    hipError_t e_ = hipSuccess;                                              \
    var = (arena).alloc<T>((count), &e_);                                    \
    if (!var) return fail(DGPU_ERR_HIP, std::string("temp alloc: ") + (e_ == hipErrorStreamCaptureUnsupported && g_captureHint ? g_captureHint : hipGetErrorString(e_))); \
  } while (0)

// ---------------------------------------------------------------------------
// Host->device parameter upload (pointer / size arrays).
//
// The pointer-array entry points take HOST arrays (as the reference's do) that
// the kernels need in device memory.  A call's arrays are packed into one block
// and looked up in a small per-device cache of blocks already resident on the
// device: a training or collective loop compresses the same buffers step after
// step, and then nothing is uploaded at all.  A miss copies the block with one
// hipMemcpyAsync from pinned memory on the caller's stream (a ~3 us blit ahead of
// the first kernel; a side-stream copy + event wait measured slower).
//
// Lifetime rules:
//   * an entry is pinned (not evictable) while a call is being enqueued with it
//   * a miss records `released` on the caller's stream when the call has been
//     enqueued; eviction waits for it (it is 8 calls old by then)
//   * a hit records nothing; the entry remembers every stream that hit it, and
//     evicting it synchronises those streams first (rare: LRU)
//   * a hit from a stream other than the uploading one waits on `copied`
std::atomic<bool> g_paramCacheEnabled{true};  // dgpu_debug_set_param_cache

class ParamCache {
 public:
  struct Entry {
    void* host = nullptr;  // pinned; also the comparison copy
    void* dev = nullptr;
    size_t cap = 0;
    size_t bytes = 0;
    uint64_t hash = 0;
    uint64_t lastUse = 0;
    int pins = 0;
    hipEvent_t copied = nullptr;
    hipEvent_t released = nullptr;
    hipStream_t uploadStream = nullptr;
    std::vector<hipStream_t> hitStreams;  // every stream that hit this entry since its upload
    bool syncAllBeforeReuse = false;      // completion could not be tracked: device synchronise before reuse
    bool everUsed = false;
    bool graphPinned = false;             // a HIP graph holds entry->dev: never evicted (dgpu_release_graph_state)
  };

  // Returns a pinned entry holding `block`; *miss tells the caller to record `released`.
  // `capturing`: the stream is being captured into a HIP graph.  The graph will replay with entry->dev baked into
  // its kernel arguments without ever passing through acquire() again, so the entry is pinned for good; and nothing
  // here may synchronise, allocate or blit then, so only a block that is already resident (a hit) can be captured.
  hipError_t acquire(const uint8_t* block, size_t bytes, hipStream_t stream, Entry** out, bool* miss, bool capturing = false) {
    const uint64_t h = hashBlock(block, bytes);
    std::lock_guard<std::mutex> g(mu_);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::vector<Entry*>& entries = perDevice_[dev];
    ++clock_;
    for (Entry* en : entries) {
      if (!g_paramCacheEnabled.load()) break;  // measurement hook: upload on every call
      if (en->everUsed && en->hash == h && en->bytes == bytes && memcmp(en->host, block, bytes) == 0) {
        if (stream != en->uploadStream && hipEventQuery(en->copied) != hipSuccess) {
          e = hipStreamWaitEvent(stream, en->copied, 0);
          if (e != hipSuccess) return e;
        }
        en->lastUse = clock_;
        en->pins++;
        if (capturing && !en->graphPinned) {
          en->graphPinned = true;
          en->pins++;  // never released: the entry stays where it is for the life of the graph
        }
        if (std::find(en->hitStreams.begin(), en->hitStreams.end(), stream) == en->hitStreams.end()) {
          en->hitStreams.push_back(stream);
        }
        *out = en;
        *miss = false;
        return hipSuccess;
      }
    }
    if (capturing) {
      g_captureHint = "the call's pointer / size arrays are not resident on the device yet and cannot be uploaded while the "
                      "stream is being captured into a HIP graph: run the same call once before capturing";
      return hipErrorStreamCaptureUnsupported;
    }
    // miss: least recently used unpinned entry, or a new one while the cache is small
    Entry* victim = nullptr;
    if (entries.size() >= kEntries) {
      for (Entry* en : entries) {
        if (en->pins == 0 && (!victim || en->lastUse < victim->lastUse)) victim = en;
      }
    }
    if (!victim) {
      victim = new Entry();
      entries.push_back(victim);
    }
    if (victim->everUsed) {
      // kernels of every call that used this block must be done with it: hits record
      // nothing, so synchronise each stream that hit it (rare: LRU eviction of a block
      // that was still in use a few calls ago)
      bool needDeviceSync = victim->syncAllBeforeReuse;
      for (hipStream_t hs : victim->hitStreams) {
        if (needDeviceSync) break;
        if (hipStreamSynchronize(hs) != hipSuccess) {
          (void)hipGetLastError();
          needDeviceSync = true;  // the stream may be gone
        }
      }
      if (needDeviceSync) {
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
      } else {
        e = hipEventSynchronize(victim->released);
        if (e != hipSuccess) return e;
      }
    }
    if (victim->cap < bytes) {
      if (victim->host) (void)hipHostFree(victim->host);
      if (victim->dev) (void)hipFree(victim->dev);
      victim->host = victim->dev = nullptr;
      victim->cap = 0;
      const size_t cap = std::max<size_t>(alignUp(bytes, 4096), 16384);
      e = hipHostMalloc(&victim->host, cap, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      e = hipMalloc(&victim->dev, cap);
      if (e != hipSuccess) return e;
      victim->cap = cap;
    }
    if (!victim->copied) {
      // ordering only (a later stream's kernels after the blit; the host asking whether the call is done): no
      // system-scope fence, which costs a cache write-back per record and slows the kernels that follow
      e = hipEventCreateWithFlags(&victim->copied, hipEventDisableTiming | hipEventDisableSystemFence);
      if (e != hipSuccess) return e;
      e = hipEventCreateWithFlags(&victim->released, hipEventDisableTiming | hipEventDisableSystemFence);
      if (e != hipSuccess) return e;
    }
    memcpy(victim->host, block, bytes);
    victim->bytes = bytes;
    victim->hash = h;
    victim->everUsed = false;  // not matchable until the copy has been enqueued
    e = hipMemcpyAsync(victim->dev, victim->host, bytes, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipEventRecord(victim->copied, stream);
    if (e != hipSuccess) return e;
    victim->everUsed = true;
    victim->uploadStream = stream;
    victim->hitStreams.clear();
    victim->syncAllBeforeReuse = false;
    victim->lastUse = clock_;
    victim->pins++;
    *out = victim;
    *miss = true;
    return hipSuccess;
  }

  // Makes the blocks HIP graphs were captured with evictable again (the caller has destroyed those graphs).
  int releaseGraphPins() {
    std::lock_guard<std::mutex> g(mu_);
    int n = 0;
    for (auto& kv : perDevice_) {
      for (Entry* en : kv.second) {
        if (en->graphPinned) {
          en->graphPinned = false;
          en->pins--;
          ++n;
        }
      }
    }
    return n;
  }

  void release(Entry* en, bool miss, hipStream_t stream) {
    std::lock_guard<std::mutex> g(mu_);
    if (miss) {
      if (hipEventRecord(en->released, stream) != hipSuccess) {
        // cannot track completion: never reuse this entry without a full sync
        en->syncAllBeforeReuse = true;
      }
    }
    en->pins--;
  }

 private:
  static constexpr size_t kEntries = 16;
  // (four independent lanes: a batch of tens of thousands of tensors has a parameter block of a megabyte, and one
  // multiply-xor chain over it was most of such a call's host time)
  static uint64_t hashBlock(const uint8_t* p, size_t n) {
    uint64_t h0 = 0x9e3779b97f4a7c15ull ^ n, h1 = 0xc2b2ae3d27d4eb4full, h2 = 0x165667b19e3779f9ull, h3 = 0x27d4eb2f165667c5ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
      uint64_t w[4];
      memcpy(w, p + i, 32);
      h0 = (h0 ^ w[0]) * 0xff51afd7ed558ccdull;
      h1 = (h1 ^ w[1]) * 0xff51afd7ed558ccdull;
      h2 = (h2 ^ w[2]) * 0xff51afd7ed558ccdull;
      h3 = (h3 ^ w[3]) * 0xff51afd7ed558ccdull;
      h0 ^= h0 >> 32;
      h1 ^= h1 >> 32;
      h2 ^= h2 >> 32;
      h3 ^= h3 >> 32;
    }
    uint64_t h = h0 ^ (h1 * 3u) ^ (h2 * 5u) ^ (h3 * 7u);
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      memcpy(&w, p + i, 8);
      h = (h ^ w) * 0xff51afd7ed558ccdull;
      h ^= h >> 32;
    }
    for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
  }
  std::mutex mu_;
  uint64_t clock_ = 0;
  std::map<int, std::vector<Entry*>> perDevice_;
};

ParamCache& paramCache() {
  static ParamCache* c = new ParamCache();  // intentionally leaked: no teardown-order issues
  return *c;
}

// Unpins the parameter block of a call (and, after an upload, records when the
// call's kernels are done with it) once everything has been enqueued.
class ParamLease {
 public:
  ParamLease() = default;
  ParamLease(const ParamLease&) = delete;
  ParamLease& operator=(const ParamLease&) = delete;
  ~ParamLease() {
    if (entry_) paramCache().release(entry_, miss_, stream_);
  }
  void bind(ParamCache::Entry* e, bool miss, hipStream_t stream) {
    entry_ = e;
    miss_ = miss;
    stream_ = stream;
  }

 private:
  ParamCache::Entry* entry_ = nullptr;
  bool miss_ = false;
  hipStream_t stream_ = nullptr;
};

BatchView viewStride(const void* base, uint64_t stride, uint32_t uniformSize) {
  BatchView v;
  v.ptrs = nullptr;
  v.base = (uint64_t)(uintptr_t)base;
  v.stride = stride;
  v.sizes = nullptr;
  v.uniformSize = uniformSize;
  return v;
}

BatchView viewPointers(const uint64_t* ptrs_dev, const uint32_t* sizes_dev, uint32_t uniformSize) {
  BatchView v;
  v.ptrs = ptrs_dev;
  v.base = 0;
  v.stride = 0;
  v.sizes = sizes_dev;
  v.uniformSize = uniformSize;
  return v;
}

// Workgroups per element of the internal histogram pass.  The hand-off to the
// normalisation costs a few microseconds per workgroup (write-through + arrival
// atomic), so workgroups should be long; two per CU already saturate HBM.  With
// a large batch that is a few long workgroups per element, with a single large
// tensor up to 256 of them.
// Raw bytes carry twice the symbols (LDS atomics, address arithmetic) per byte of traffic: three workgroups per CU
// (256 x 1 MiB Zipf bytes: 53.9 -> 51.3 us; the exponent histogram loses 2 us with three).
constexpr uint32_t kHistTargetWgs = 512, kHistTargetWgsRaw = 768, kHistAccWgs = 512, kHistAccMaxBatch = 2;
// (The counters of an element in 16 sets instead of one -- so that 512 workgroups do not queue on 256 addresses -- measured
// SLOWER: 1 x 128 Mi bf16 histogram 49.3 us against 47.9, 1 x 16 Mi 18.4-19.5 against 16.6-16.9
// (profiles/r06_ab_hist_acc_sets_*.txt): summing the sets costs the normalising workgroup more than the queue does.)
uint32_t histPartsFor(uint32_t B, uint32_t maxBytes, bool raw) {
  const uint32_t bySize = divUp(std::max(maxBytes, 1u), 32u * 1024u);
  const uint32_t byBatch = divUp(raw ? kHistTargetWgsRaw : kHistTargetWgs, std::max(B, 1u));
  return std::max(1u, std::min(std::min(bySize, byBatch), 256u));
}
// One or two large elements: the counts of an element's workgroups meet in 256 library-owned
// atomic counters instead of per-workgroup partial histograms, so an element can be spread over
// ~512 workgroups without the normalising workgroup having to sum 512 partial histograms (a single
// 256 MiB tensor: 79 -> 52 us; more workgroups than that lose to contention on the counters).
bool histAccumulates(uint32_t B, uint32_t maxBytes, bool raw);
uint32_t histPartsAccFor(uint32_t B, uint32_t maxBytes) {
  const uint32_t bySize = divUp(std::max(maxBytes, 1u), 64u * 1024u);
  const uint32_t byBatch = divUp(kHistAccWgs, std::max(B, 1u));
  return std::max(1u, std::min(bySize, byBatch));
}

uint32_t gridX(uint32_t maxBytes, uint32_t bytesPerBlock, uint32_t cap) {
  uint32_t x = divUp(std::max(maxBytes, 1u), bytesPerBlock);
  return std::max(1u, std::min(x, cap));
}

bool validProbBits(int p) { return p == 9 || p == 10 || p == 11; }
bool validFloatType(uint32_t ft) { return ft == kFloat16 || ft == kBFloat16 || ft == kFloat32; }

// getMaxCompressedSize, GpuANSEncode.cu:13-25 (block SIZE passed as block COUNT [sic]).  Upstream CHECKs the result
// against INT32_MAX (GpuANSEncode.cu:22: inputs beyond 419 321 blocks = 1 717 538 816 bytes abort); here such a size
// yields 0 and the encode entry points reject it.
constexpr uint32_t kMaxEncodableBytes = 419321u * kBlockSize;
uint32_t maxCompressedSizeHost(uint32_t bytes) {
  uint32_t blocks = divUp(bytes, kBlockSize);
  size_t raw = ansOverhead(kBlockSize);
  raw += (size_t)roundUp(kBlockSize + kBlockSize / 4, 16) * blocks;
  raw = alignUp(raw, 16);
  if (raw > (size_t)INT32_MAX) return 0u;
  return (uint32_t)raw;
}
bool encodableSize(uint32_t symbols) { return symbols <= kMaxEncodableBytes; }

// Device batch description assembled by each entry point.
struct DeviceBatch {
  BatchView in;
  BatchView out;
  uint32_t maxSize = 0;
};

// Layout of the packed parameter block: [in ptrs][out ptrs][sizes]
struct HostParams {
  std::vector<uint64_t> inPtrs, outPtrs;
  std::vector<uint32_t> sizes;
  std::vector<uint32_t> inBytes;  // decode, *_bounded entry points: bytes available per compressed input
  std::vector<uint32_t> work;     // work lists of a batch whose elements differ widely in size (RaggedPlan)
};

bool streamIsCapturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st == hipStreamCaptureStatusActive;
}

int uploadParams(
    ParamLease& lease, hipStream_t stream, const HostParams& hp,
    const uint64_t** inPtrs_dev, const uint64_t** outPtrs_dev, const uint32_t** sizes_dev,
    const uint32_t** inBytes_dev = nullptr, const uint32_t** work_dev = nullptr) {
  const size_t nIn = hp.inPtrs.size(), nOut = hp.outPtrs.size(), nSz = hp.sizes.size(), nIb = hp.inBytes.size();
  const size_t nWk = work_dev ? hp.work.size() : 0;
  const size_t workAt = (nIn + nOut) * 8 + alignUp(nSz * 4, 8) + alignUp(nIb * 4, 8);
  const size_t bytes = workAt + alignUp(nWk * 4, 8);
  if (bytes == 0) return DGPU_OK;
  static thread_local std::vector<uint8_t> block;
  block.assign(bytes, 0);
  uint8_t* h = block.data();
  if (nIn) memcpy(h, hp.inPtrs.data(), nIn * 8);
  if (nOut) memcpy(h + nIn * 8, hp.outPtrs.data(), nOut * 8);
  if (nSz) memcpy(h + (nIn + nOut) * 8, hp.sizes.data(), nSz * 4);
  if (nIb) memcpy(h + (nIn + nOut) * 8 + alignUp(nSz * 4, 8), hp.inBytes.data(), nIb * 4);
  if (nWk) memcpy(h + workAt, hp.work.data(), nWk * 4);
  ParamCache::Entry* entry = nullptr;
  bool miss = false;
  {
    const bool capturing = streamIsCapturing(stream);
    const hipError_t ae = paramCache().acquire(h, bytes, stream, &entry, &miss, capturing);
    if (ae == hipErrorStreamCaptureUnsupported && capturing && g_captureHint) return fail(DGPU_ERR_HIP, std::string("HIP graph capture: ") + g_captureHint);
    DGPU_HIP(ae);
  }
  lease.bind(entry, miss, stream);
  uint8_t* dev = (uint8_t*)entry->dev;
  *inPtrs_dev = nIn ? (const uint64_t*)dev : nullptr;
  *outPtrs_dev = nOut ? (const uint64_t*)(dev + nIn * 8) : nullptr;
  *sizes_dev = nSz ? (const uint32_t*)(dev + (nIn + nOut) * 8) : nullptr;
  if (inBytes_dev) *inBytes_dev = nIb ? (const uint32_t*)(dev + (nIn + nOut) * 8 + alignUp(nSz * 4, 8)) : nullptr;
  if (work_dev) *work_dev = nWk ? (const uint32_t*)(dev + workAt) : nullptr;
  return DGPU_OK;
}


// A pointer batch whose addresses form an arithmetic progression and whose sizes are all equal IS a stride batch
// (the rows of one tensor, the rows of the output matrix the tensor API allocates, any batch of one): it needs no
// parameter block on the device at all -- no cache lookup, and no blit + event records when the cache would miss
// (+9 us per call).  sizesOnOut: the sizes are output capacities (decode) rather than input sizes (encode).
bool progression(const std::vector<uint64_t>& p, uint64_t* stride) {
  *stride = p.size() > 1 ? p[1] - p[0] : 0;
  for (size_t i = 1; i < p.size(); ++i) {
    if (p[i] - p[i - 1] != *stride) return false;
  }
  return !p.empty();
}
bool asStrideViews(const HostParams& hp, bool sizesOnOut, BatchView* in, BatchView* out) {
  if (!hp.inBytes.empty() || hp.inPtrs.size() != hp.outPtrs.size()) return false;
  uint64_t inStride = 0, outStride = 0;
  if (!progression(hp.inPtrs, &inStride) || !progression(hp.outPtrs, &outStride)) return false;
  uint32_t u = hp.sizes.empty() ? 0u : hp.sizes[0];
  for (uint32_t sz : hp.sizes) {
    if (sz != u) return false;
  }
  *in = viewStride((const void*)(uintptr_t)hp.inPtrs[0], inStride, sizesOnOut ? 0u : u);
  *out = viewStride((const void*)(uintptr_t)hp.outPtrs[0], outStride, sizesOnOut ? u : 0u);
  return true;
}

// ---------------------------------------------------------------------------
// Launch sequences
// ---------------------------------------------------------------------------

uint32_t numComputeUnits() {
  static std::mutex mu;
  static std::map<int, uint32_t> cus;
  std::lock_guard<std::mutex> g(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  hipDeviceProp_t prop;
  uint32_t n = 256;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = (uint32_t)prop.multiProcessorCount;
  cus[dev] = n;
  return n;
}

// The encoder runs as persistent workgroups: as many as fit on the chip at once
// (or fewer, if there are fewer tiles).  Float inputs use the small-stage /
// spilling variant (6 workgroups per CU), raw bytes the worst-case stage (3 per CU; kernels_encode.h).
constexpr bool encodeSpills(uint32_t ft) { return ft != 0; }

// (`resident` also sizes the spill-slot pool of the hardware-dispatched float encoders, whose wavefronts hold a pair
// of slots across their look-back wait: the pool must cover whichever form of the kernel is launched, so where both
// forms exist the larger occupancy of the two counts -- today they compile to the same register count.)
template <int P, uint32_t FT, uint32_t TB>
uint32_t encodeGridPFT(uint32_t tickets) {
  static const uint32_t perCu = [] {
    int n = 0, m = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &n, (k_ans_encode<P, FT, encodeSpills(FT), TB, true>), encThreads(TB), encLdsBytes(P, encodeSpills(FT), FT, TB)) != hipSuccess || n < 1) {
      n = 1;
    }
    if constexpr (!(encodeSpills(FT) && TB >= kBlocksPerTile)) {  // (8-block float tiles exist in the persistent form only)
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
              &m, (k_ans_encode<P, FT, encodeSpills(FT), TB, false>), encThreads(TB), encLdsBytes(P, encodeSpills(FT), FT, TB)) != hipSuccess) {
        m = 0;
      }
    }
    return (uint32_t)std::max(n, m);
  }();
  return std::max(1u, std::min(tickets, perCu * numComputeUnits()));
}
// ... the persistent 8-block bf16 / fp32 encoder with the wide stage (kernels_encode.h, kSpillStageWordsWide)
template <int P, uint32_t FT>
uint32_t encodeGridWidePF(uint32_t tickets) {
  if constexpr (FT == kBFloat16 || FT == kFloat32) {
    static const uint32_t perCu = [] {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (k_ans_encode<P, FT, true, kBlocksPerTile, true, true>), encThreads(kBlocksPerTile),
                                                       encLdsBytes(P, true, FT, kBlocksPerTile, true)) != hipSuccess || n < 1) {
        n = 1;
      }
      return (uint32_t)n;
    }();
    return std::max(1u, std::min(tickets, perCu * numComputeUnits()));
  } else {
    return encodeGridPFT<P, FT, kBlocksPerTile>(tickets);
  }
}
// Batches of single-block elements: two ELEMENTS per wavefront (kernels_pairs.h) instead of one with an idle half.
template <int P, uint32_t FT>
uint32_t encodePairGridPF(uint32_t elements) {
  static const uint32_t perCu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (k_ans_encode_pair<P, FT, encodeSpills(FT)>), 64,
                                                     encPairLdsBytes(P, encodeSpills(FT), FT)) != hipSuccess || n < 1) {
      n = 1;
    }
    return (uint32_t)n;
  }();
  // (workgroups = wavefronts that can be resident at once: the size of the spill-slot pool; the LAUNCH is one
  // workgroup per pair, encodeCommon)
  return std::max(1u, std::min((elements + 1u) / 2u, perCu * numComputeUnits()));
}
template <int P, uint32_t FT>
uint32_t encodeGridPF(uint32_t tickets, uint32_t tileBlocks, bool wide) {
  if (tileBlocks == kBlocksPerSingleTile) return encodePairGridPF<P, FT>(tickets);  // (one ticket per element)
  if (wide && tileBlocks == kBlocksPerTile) return encodeGridWidePF<P, FT>(tickets);
  return tileBlocks == kBlocksPerTinyTile ? encodeGridPFT<P, FT, kBlocksPerTinyTile>(tickets)
      : tileBlocks == kBlocksPerSmallTile ? encodeGridPFT<P, FT, kBlocksPerSmallTile>(tickets)
                                          : encodeGridPFT<P, FT, kBlocksPerTile>(tickets);
}

template <int P, uint32_t FT, bool kPersistent>
int launchEncodePFD(const EncodeArgs& a, uint32_t tileBlocks, uint32_t grid, bool wide, hipStream_t stream) {
  constexpr bool kSpill = encodeSpills(FT);
  if constexpr (FT == kBFloat16 || FT == kFloat32) {
    if (wide && tileBlocks == kBlocksPerTile) {
      DGPU_LAUNCH("k_ans_encode", stream, (k_ans_encode<P, FT, true, kBlocksPerTile, true, true>), dim3(grid), dim3(kBlocksPerTile * 32),
                  encLdsBytes(P, true, FT, kBlocksPerTile, true), stream, a);
      DGPU_HIP(hipGetLastError());
      return DGPU_OK;
    }
  }
  if (tileBlocks == kBlocksPerTinyTile) {
    DGPU_LAUNCH("k_ans_encode", stream, (k_ans_encode<P, FT, kSpill, kBlocksPerTinyTile, kPersistent>), dim3(grid), dim3(kBlocksPerTinyTile * 32),
                encLdsBytes(P, kSpill, FT, kBlocksPerTinyTile), stream, a);
  } else if (tileBlocks == kBlocksPerSmallTile) {
    DGPU_LAUNCH("k_ans_encode", stream, (k_ans_encode<P, FT, kSpill, kBlocksPerSmallTile, kPersistent>), dim3(grid), dim3(kBlocksPerSmallTile * 32),
                encLdsBytes(P, kSpill, FT, kBlocksPerSmallTile), stream, a);
  } else {
    // (8-block float tiles exist in the persistent form only: encoderHardwareDispatch never asks for the other)
    constexpr bool kPers = kPersistent || kSpill;
    DGPU_LAUNCH("k_ans_encode", stream, (k_ans_encode<P, FT, kSpill, kBlocksPerTile, kPers>), dim3(grid), dim3(kBlocksPerTile * 32),
                encLdsBytes(P, kSpill, FT, kBlocksPerTile), stream, a);
  }
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}
// `hwDispatch`: grid == a.numTickets, one workgroup per tile (k_ans_encode<..., kPersistent = false>)
template <int P, uint32_t FT>
int launchEncodePF(const EncodeArgs& a, uint32_t tileBlocks, uint32_t grid, bool hwDispatch, bool wide, hipStream_t stream) {
  constexpr bool kSpill = encodeSpills(FT);
  if (tileBlocks == kBlocksPerSingleTile) {
    DGPU_LAUNCH("k_ans_encode_pair", stream, (k_ans_encode_pair<P, FT, kSpill>), dim3(grid), dim3(64), encPairLdsBytes(P, kSpill, FT), stream, a);
    DGPU_HIP(hipGetLastError());
    return DGPU_OK;
  }
  if (hwDispatch) return launchEncodePFD<P, FT, false>(a, tileBlocks, grid, wide, stream);
  return launchEncodePFD<P, FT, true>(a, tileBlocks, grid, wide, stream);
}

#define DGPU_ENCODE_DISPATCH(P_, FT_, EXPR)                                   \
  switch (FT_) {                                                              \
    case 0:                                                                   \
      switch (P_) { case 9: { constexpr int kP = 9; constexpr uint32_t kFT = 0; EXPR; } break;          \
                    case 10: { constexpr int kP = 10; constexpr uint32_t kFT = 0; EXPR; } break;        \
                    default: { constexpr int kP = 11; constexpr uint32_t kFT = 0; EXPR; } break; }      \
      break;                                                                  \
    case kFloat16:                                                            \
      switch (P_) { case 9: { constexpr int kP = 9; constexpr uint32_t kFT = kFloat16; EXPR; } break;   \
                    case 10: { constexpr int kP = 10; constexpr uint32_t kFT = kFloat16; EXPR; } break; \
                    default: { constexpr int kP = 11; constexpr uint32_t kFT = kFloat16; EXPR; } break; } \
      break;                                                                  \
    case kBFloat16:                                                           \
      switch (P_) { case 9: { constexpr int kP = 9; constexpr uint32_t kFT = kBFloat16; EXPR; } break;  \
                    case 10: { constexpr int kP = 10; constexpr uint32_t kFT = kBFloat16; EXPR; } break; \
                    default: { constexpr int kP = 11; constexpr uint32_t kFT = kBFloat16; EXPR; } break; } \
      break;                                                                  \
    default:                                                                  \
      switch (P_) { case 9: { constexpr int kP = 9; constexpr uint32_t kFT = kFloat32; EXPR; } break;   \
                    case 10: { constexpr int kP = 10; constexpr uint32_t kFT = kFloat32; EXPR; } break; \
                    default: { constexpr int kP = 11; constexpr uint32_t kFT = kFloat32; EXPR; } break; } \
      break;                                                                  \
  }

uint32_t encodeGrid(int P, uint32_t ft, uint32_t tileBlocks, uint32_t tickets, bool wide) {
  uint32_t g = 1;
  DGPU_ENCODE_DISPATCH(P, ft, g = (encodeGridPF<kP, kFT>(tickets, tileBlocks, wide)));
  return g;
}
// The wide stage (five workgroups per CU, no flushes on N(0,1) exponents) for persistent 8-block bf16 / fp32 tiles of
// batches whose elements have few tiles; elements of many tiles keep six workgroups per CU in flight behind their
// in-order commit (kernels_encode.h, kSpillStageWordsWide; profiles/r06_ab_encoder_five_per_cu_*.txt).
constexpr uint32_t kWideStageMaxTiles = 32;
bool encoderWideStage(uint32_t floatType, uint32_t tileBlocks, uint32_t maxTiles) {
  return (floatType == kBFloat16 || floatType == kFloat32) && tileBlocks == kBlocksPerTile && maxTiles <= kWideStageMaxTiles;
}

int launchEncode(int P, uint32_t ft, const EncodeArgs& a, uint32_t tileBlocks, uint32_t grid, bool hwDispatch, bool wide, hipStream_t stream) {
  int rc = DGPU_OK;
  DGPU_ENCODE_DISPATCH(P, ft, rc = (launchEncodePF<kP, kFT>(a, tileBlocks, grid, hwDispatch, wide, stream)));
  return rc;
}

// blocks per encoder tile for a batch whose largest element has `maxSize` symbols
uint32_t encTileBlocksFor(uint32_t maxSize) {
  const uint32_t blocks = divUp(maxSize, kBlockSize);
  return blocks <= kBlocksPerSingleTile ? kBlocksPerSingleTile
      : blocks <= kBlocksPerTinyTile    ? kBlocksPerTinyTile
      : blocks <= kBlocksPerSmallTile   ? kBlocksPerSmallTile
                                        : kBlocksPerTile;
}
uint32_t tilesFor(uint32_t maxSize) { return divUp(divUp(maxSize, kBlockSize), encTileBlocksFor(maxSize)); }

uint32_t absentWorkgroupModulo();  // test hook, defined with the C ABI below

// How the workgroups of the tiled encoder come to their tiles: persistent workgroups with a static ticket map, or one
// workgroup per tile, dispatched by the hardware in ticket order.  Measured on MI355X
// (profiles/r05_ab_encoder_hw_dispatch.txt, r05_ab_small_tiles_hw_dispatch.txt, r05_ab_pair_encoder_hw_dispatch.txt):
//   * raw bytes, 256 x 1 MiB: 152.7 -> 141.5 us under hardware dispatch (8192 tiles on 768 resident workgroups: the
//     compute-bound row loop of a slow CU no longer holds a fixed share of them);
//   * float tiles of 2 / 4 blocks (batches of elements of <= 16 Ki words): 118 -> 96 us / 98 -> 88 us;
//   * 8-block float tiles: nothing on 256 x 512 Ki, 1-3 % slower on few large tensors: they stay persistent;
//   * k_ans_encode_pair (single-block elements) always runs one workgroup per pair: 133.7 -> 96.8 us.
// -1 = this policy; dgpu_debug_set_encoder_dispatch / DGPU_ENC_DISPATCH force 0 (persistent) or 1 (hardware) where the
// kernel exists in both forms (tests, A/B runs).
std::atomic<int> g_encDispatch{[] {
  const char* e = getenv("DGPU_ENC_DISPATCH");
  return e && *e ? atoi(e) : -1;
}()};
bool encoderHardwareDispatch(uint32_t numTickets, uint32_t resident, uint32_t floatType, uint32_t tileBlocks) {
  if (encodeSpills(floatType) && tileBlocks >= kBlocksPerTile) return false;  // (no hardware-dispatched build of those)
  const int m = g_encDispatch.load();
  if (m >= 0) return m != 0;
  return numTickets > resident;  // more tiles than slots: let the hardware balance them
}

// Batches whose elements differ widely in size -- the tensors of a model in one call: a few matrices, many vectors.
// The grids of the histogram, the encoder and the decoder are rectangles laid out for the LARGEST element (as
// upstream's are); with one 32 Mi-word tensor next to 255 small ones that is 262 144 encoder tickets of which 1 279
// exist, spread over the persistent workgroups by a static map that hands the large tensor's tiles to three of them,
// and two histogram workgroups for its 64 MiB (tools/ragged_probe.py: 5.7 ms per compress call against 56 us + 37 us
// for the two size classes on their own).  The host knows the sizes (they arrive as host arrays), so for such a batch
// it lists the work that exists -- HostParams::work, uploaded with the pointers -- and the kernels take their
// (element, tile / part) from the list instead of from the rectangle:
//   * tiles: the encoder's element by element, the large elements first (a tile's predecessor has the ticket before
//     its own, and descriptors and claim words exist for the listed tiles only); the decoder's, which do not depend on
//     one another, tile-major;
//   * histogram parts: every element cut into parts of histPartBytes (chosen for the usual number of workgroups over
//     the WHOLE batch), element-major, so that an element's partial histograms are consecutive.
// Used when at least a fifth of the rectangle's tiles do not exist (256 bf16 tensors of 0.06 .. 1 Mi words: compress
// 228 -> 181 us; of 0.5 .. 1 Mi: 247 -> 229; of 0.85 .. 1 Mi the rectangle is 3 % faster: profiles/r05_ab_work_lists.txt);
// dgpu_debug_set_work_lists forces it on (1: whenever the batch has a size array) or off (0) for tests.
inline uint64_t divUp64(uint64_t a, uint64_t b) { return (a + b - 1u) / b; }
inline uint64_t roundUp64(uint64_t a, uint64_t b) { return divUp64(a, b) * b; }
struct RaggedPlan {
  bool use = false;
  // HostParams::work of an encode call: [numTiles] x {element << 16 | tile}, [numHistParts] x {element << 16 | part},
  // [B] x {the element's first ticket = index of its first descriptor and claim word}
  uint32_t numTiles = 0;
  uint32_t numHistParts = 0;
  uint32_t histPartBytes = 0;
};
std::atomic<int> g_workLists{[] {
  const char* e = getenv("DGPU_WORK_LISTS");
  return e && *e ? atoi(e) : -1;
}()};
// tiles of `tileSymbols` symbols; minTiles: 1 where an element without symbols still needs its first tile (decode)
// elementMajor (encoder): an element's tiles are consecutive -- descriptors and claim words are then indexed by the
// ticket -- elements in descending size (the large ones start first, the small ones fill the end); *tileBase receives
// each element's first ticket.  Otherwise (decoder: no dependence between tiles) tile-major.
bool planTileList(const std::vector<uint32_t>& sizes, uint32_t tileSymbols, uint32_t maxTiles, uint32_t minTiles, std::vector<uint32_t>* work,
                  std::vector<uint32_t>* tileBase = nullptr) {
  const int mode = g_workLists.load();
  const size_t B = sizes.size();
  if (mode == 0 || B == 0 || B > 65535u || maxTiles > 65536u || tileSymbols == 0) return false;
  std::vector<uint32_t> tiles(B);
  uint64_t total = 0;
  for (size_t b = 0; b < B; ++b) {
    tiles[b] = std::max(divUp(sizes[b], tileSymbols), minTiles);
    total += tiles[b];
  }
  if (mode != 1 && (B < 2 || maxTiles < 2 || total * 5u > (uint64_t)B * maxTiles * 4u)) return false;
  if (total > 0x7fffffffull) return false;
  std::vector<uint32_t> order(B);
  for (size_t b = 0; b < B; ++b) order[b] = (uint32_t)b;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return tiles[x] > tiles[y]; });
  const size_t first = work->size();
  work->reserve(first + (size_t)total);
  if (tileBase) {
    tileBase->assign(B, 0u);
    for (size_t i = 0; i < B; ++i) {
      const uint32_t b = order[i];
      (*tileBase)[b] = (uint32_t)(work->size() - first);
      for (uint32_t r = 0; r < tiles[b]; ++r) work->push_back((b << 16) | r);
    }
  } else {
    for (uint32_t r = 0; r < maxTiles; ++r) {
      for (size_t i = 0; i < B && tiles[order[i]] > r; ++i) work->push_back((order[i] << 16) | r);
    }
  }
  return true;
}
constexpr uint32_t kHistTargetWgsForLists = 512, kHistTargetWgsForListsRaw = 768;
void planHistList(const std::vector<uint32_t>& sizes, uint32_t wordBytes, bool raw, RaggedPlan* plan, std::vector<uint32_t>* work) {
  uint64_t totalBytes = 0;
  for (uint32_t sz : sizes) totalBytes += (uint64_t)sz * wordBytes;
  const uint64_t target = raw ? kHistTargetWgsForListsRaw : kHistTargetWgsForLists;
  const uint64_t partBytes = std::max<uint64_t>(32u * 1024u, roundUp64(divUp64(totalBytes, target), 16u * 1024u));
  plan->histPartBytes = (uint32_t)std::min<uint64_t>(partBytes, 0x40000000ull);
  const size_t before = work->size();
  for (size_t b = 0; b < sizes.size(); ++b) {
    const uint64_t bytes = (uint64_t)sizes[b] * wordBytes;
    const uint32_t parts = (uint32_t)std::max<uint64_t>(1u, divUp64(bytes, plan->histPartBytes));
    for (uint32_t p = 0; p < parts; ++p) work->push_back(((uint32_t)b << 16) | p);
  }
  plan->numHistParts = (uint32_t)(work->size() - before);
}
// Work lists of an encode call: false = the rectangles.
bool planEncode(const std::vector<uint32_t>& sizes, uint32_t floatType, uint32_t maxSize, bool needHist, RaggedPlan* plan, std::vector<uint32_t>* work) {
  const uint32_t tileBlocks = encTileBlocksFor(maxSize);
  if (tileBlocks == kBlocksPerSingleTile) return false;  // single-block batches: one wavefront per element, nothing to list
  std::vector<uint32_t> tileBase;
  if (!planTileList(sizes, tileBlocks * kBlockSize, tilesFor(maxSize), 0u, work, &tileBase)) return false;
  plan->use = true;
  plan->numTiles = (uint32_t)work->size();
  if (needHist) planHistList(sizes, floatType ? floatWordBytes(floatType) : 1u, floatType == 0, plan, work);
  work->insert(work->end(), tileBase.begin(), tileBase.end());
  return true;
}

// SIZE CLASSES inside one batch.  The tile geometry of a call -- pairs of single-block elements per wavefront, tiles of 2,
// 4 or 8 blocks -- used to be chosen once, from the largest element (encTileBlocksFor(maxSize); upstream does the same:
// one grid laid out for maxSize, GpuANSEncode.cuh:753-771).  One large tensor next to thousands of small ones then ran
// every small element on an 8-block tile -- seven of its eight half-waves idle, where the pair kernels are 1.4-2 x faster
// on such elements.  The host already lists the work of such a batch; it now lists it PER CLASS and launches each class
// on the kernels of its own geometry, the classes one after the other on the caller's stream (large elements first).
// All kernels index the batch's arrays by the element's own index, so a class is nothing but its lists: tiles and
// histogram parts (EncodeArgs::workMap, HistFuse::workMap) or, for the single-block class, the elements to pair up.
// A class of fewer than kMinClassElements elements joins the next larger one (a launch costs more than their idle
// lanes), and a batch whose smaller classes together hold fewer than kMinSplitElements elements is not split at all:
// in ONE launch its few small elements run beside the large ones' tiles (1 x 32 Mi + 255 x 2 Ki bf16: 71 us together
// against 56 + 38 one after the other, profiles/r05_ab_work_lists.txt).
constexpr uint32_t kMinClassElements = 32, kMinSplitElements = 256;
struct EncodeClass {
  uint32_t tileBlocks = 0;
  uint32_t maxSize = 0;                                       // of the class's elements
  uint32_t tilesAt = 0, numTiles = 0;                         // tiles of >= 2 blocks: [numTiles] element << 16 | tile, element by element
  uint32_t histAt = 0, numHistParts = 0, histPartBytes = 0;   // ... [numHistParts] element << 16 | part
  uint32_t tileBaseAt = 0;                                    // ... [B] first ticket of each of the class's elements
  uint32_t elemsAt = 0, numElems = 0;                         // single-block class: [numElems] the elements
};
std::atomic<int> g_sizeClasses{[] {
  const char* e = getenv("DGPU_SIZE_CLASSES");
  return e && *e ? atoi(e) : -1;
}()};
// Splits the batch into size classes (false: one geometry for the call, as before).  `classOf(size)` -> blocks per tile /
// workgroup of an element of that size; `pairsOk`: the single-block class has kernels of its own (not float32 encode).
template <typename ClassOf>
bool classifyBySize(const std::vector<uint32_t>& sizes, ClassOf classOf, bool pairsOk, std::vector<uint32_t>* classOfElem,
                    std::vector<uint32_t>* classesOut) {
  const int mode = g_sizeClasses.load();
  const size_t B = sizes.size();
  if (mode == 0 || g_workLists.load() == 0 || B < 2 || B > 65535u) return false;
  std::map<uint32_t, uint32_t> count;
  classOfElem->resize(B);
  for (size_t b = 0; b < B; ++b) {
    uint32_t c = classOf(sizes[b]);
    if (c == 1u && !pairsOk) c = 2u;
    (*classOfElem)[b] = c;
    count[c]++;
  }
  if (count.size() < 2) return false;
  // small classes join the next larger one that exists
  for (auto it = count.begin(); it != count.end();) {
    auto next = std::next(it);
    if (next != count.end() && it->second < (mode == 1 ? 1u : kMinClassElements)) {
      for (size_t b = 0; b < B; ++b) {
        if ((*classOfElem)[b] == it->first) (*classOfElem)[b] = next->first;
      }
      next->second += it->second;
      it = count.erase(it);
    } else {
      it = next;
    }
  }
  if (count.size() < 2) return false;
  uint32_t small = 0;
  for (auto& kv : count) {
    if (kv.first != count.rbegin()->first) small += kv.second;
  }
  if (mode != 1 && small < kMinSplitElements) return false;
  classesOut->clear();
  for (auto it = count.rbegin(); it != count.rend(); ++it) classesOut->push_back(it->first);  // large elements first
  return true;
}
bool planEncodeClasses(const std::vector<uint32_t>& sizes, uint32_t floatType, std::vector<EncodeClass>* classes, std::vector<uint32_t>* work) {
  std::vector<uint32_t> classOfElem, order;
  if (!classifyBySize(sizes, [](uint32_t sz) { return encTileBlocksFor(sz); }, floatType != kFloat32, &classOfElem, &order)) return false;
  const size_t B = sizes.size();
  const uint32_t wordBytes = floatType ? floatWordBytes(floatType) : 1u;
  classes->clear();
  for (uint32_t c : order) {
    EncodeClass k;
    k.tileBlocks = c;
    std::vector<uint32_t> elems;
    for (size_t b = 0; b < B; ++b) {
      if (classOfElem[b] == c) {
        elems.push_back((uint32_t)b);
        k.maxSize = std::max(k.maxSize, sizes[b]);
      }
    }
    if (c == kBlocksPerSingleTile) {
      k.elemsAt = (uint32_t)work->size();
      k.numElems = (uint32_t)elems.size();
      work->insert(work->end(), elems.begin(), elems.end());
    } else {
      // tiles element by element, the class's larger elements first; histogram parts sized for the usual number of
      // workgroups over the CLASS (the classes run one after the other, each should fill the chip)
      std::stable_sort(elems.begin(), elems.end(), [&](uint32_t x, uint32_t y) { return sizes[x] > sizes[y]; });
      const uint32_t tileSymbols = c * kBlockSize;
      std::vector<uint32_t> tileBase(B, 0u);
      k.tilesAt = (uint32_t)work->size();
      uint64_t classBytes = 0;
      for (uint32_t b : elems) {
        tileBase[b] = (uint32_t)(work->size() - k.tilesAt);
        const uint32_t tiles = divUp(sizes[b], tileSymbols);
        if (tiles > 65536u) {
          work->clear();
          classes->clear();
          return false;
        }
        for (uint32_t r = 0; r < tiles; ++r) work->push_back((b << 16) | r);
        classBytes += (uint64_t)sizes[b] * wordBytes;
      }
      k.numTiles = (uint32_t)(work->size() - k.tilesAt);
      const uint64_t target = floatType == 0 ? kHistTargetWgsForListsRaw : kHistTargetWgsForLists;
      const uint64_t partBytes = std::max<uint64_t>(32u * 1024u, roundUp64(divUp64(classBytes, target), 16u * 1024u));
      k.histPartBytes = (uint32_t)std::min<uint64_t>(partBytes, 0x40000000ull);
      k.histAt = (uint32_t)work->size();
      for (uint32_t b : elems) {
        const uint32_t parts = (uint32_t)std::max<uint64_t>(1u, divUp64((uint64_t)sizes[b] * wordBytes, k.histPartBytes));
        if (parts > 65536u) {
          work->clear();
          classes->clear();
          return false;
        }
        for (uint32_t q = 0; q < parts; ++q) work->push_back((b << 16) | q);
      }
      k.numHistParts = (uint32_t)(work->size() - k.histAt);
      k.tileBaseAt = (uint32_t)work->size();
      work->insert(work->end(), tileBase.begin(), tileBase.end());
    }
    classes->push_back(k);
  }
  return true;
}

// ... and the decoder's classes, from the output capacities (its tiles do not depend on one another: tile-major lists)
struct DecodeClass {
  uint32_t tileBlocks = 0, maxBlocks = 0;
  uint32_t tilesAt = 0, numTiles = 0;   // tiles of >= 2 blocks: [numTiles] element << 16 | tile
  uint32_t elemsAt = 0, numElems = 0;   // single-block class: [numElems] the elements
};
uint32_t decTileBlocksFor(uint32_t maxBlocks);
bool planDecodeClasses(const std::vector<uint32_t>& caps, std::vector<DecodeClass>* classes, std::vector<uint32_t>* work) {
  std::vector<uint32_t> classOfElem, order;
  if (!classifyBySize(caps, [](uint32_t cap) { return decTileBlocksFor(divUp(cap, kBlockSize)); }, true, &classOfElem, &order)) return false;
  const size_t B = caps.size();
  classes->clear();
  for (uint32_t c : order) {
    DecodeClass k;
    k.tileBlocks = c;
    std::vector<uint32_t> elems;
    for (size_t b = 0; b < B; ++b) {
      if (classOfElem[b] == c) {
        elems.push_back((uint32_t)b);
        k.maxBlocks = std::max(k.maxBlocks, divUp(caps[b], kBlockSize));
      }
    }
    if (c == 1u) {
      k.elemsAt = (uint32_t)work->size();
      k.numElems = (uint32_t)elems.size();
      work->insert(work->end(), elems.begin(), elems.end());
    } else {
      std::stable_sort(elems.begin(), elems.end(), [&](uint32_t x, uint32_t y) { return caps[x] > caps[y]; });
      const uint32_t tileSymbols = c * kBlockSize;
      const uint32_t maxTiles = std::max(1u, divUp(k.maxBlocks, c));
      if (maxTiles > 65536u) {
        work->clear();
        classes->clear();
        return false;
      }
      k.tilesAt = (uint32_t)work->size();
      for (uint32_t r = 0; r < maxTiles; ++r) {
        for (uint32_t b : elems) {
          if (std::max(divUp(caps[b], tileSymbols), 1u) <= r) break;  // (descending capacities)
          work->push_back((b << 16) | r);
        }
      }
      k.numTiles = (uint32_t)(work->size() - k.tilesAt);
    }
    classes->push_back(k);
  }
  return true;
}

// A training or collective loop compresses the same list of tensors step after step: the last plan of each kind is kept
// per host thread and reused when the sizes (and everything else the plan depends on) are the same -- planning a batch
// of 32 769 tensors costs ~100 us of host time, comparing its sizes 10.
template <typename Class>
struct ClassPlanCache {
  std::vector<uint32_t> sizes, work;
  std::vector<Class> classes;
  uint32_t floatType = 0xffffffffu;
  int modeClasses = -2, modeLists = -2;
  bool valid = false, split = false;
  bool matches(const std::vector<uint32_t>& sz, uint32_t ft) const {
    return valid && floatType == ft && modeClasses == g_sizeClasses.load() && modeLists == g_workLists.load() && sizes.size() == sz.size() &&
        (sz.empty() || memcmp(sizes.data(), sz.data(), sz.size() * 4u) == 0);
  }
  void remember(const std::vector<uint32_t>& sz, uint32_t ft, bool didSplit, const std::vector<Class>& cl, const std::vector<uint32_t>& wk) {
    sizes = sz, floatType = ft, split = didSplit, classes = cl, work = wk;
    modeClasses = g_sizeClasses.load(), modeLists = g_workLists.load();
    valid = true;
  }
};
bool planEncodeClassesCached(const std::vector<uint32_t>& sizes, uint32_t floatType, std::vector<EncodeClass>* classes, std::vector<uint32_t>* work) {
  if (sizes.size() < kMinSplitElements) return planEncodeClasses(sizes, floatType, classes, work);  // (cheap to plan, and rarely split)
  static thread_local ClassPlanCache<EncodeClass> cache;
  if (!cache.matches(sizes, floatType)) {
    std::vector<EncodeClass> cl;
    std::vector<uint32_t> wk;
    const bool split = planEncodeClasses(sizes, floatType, &cl, &wk);
    cache.remember(sizes, floatType, split, cl, wk);
  }
  if (!cache.split) return false;
  *classes = cache.classes;
  *work = cache.work;
  return true;
}
bool planDecodeClassesCached(const std::vector<uint32_t>& caps, std::vector<DecodeClass>* classes, std::vector<uint32_t>* work) {
  if (caps.size() < kMinSplitElements) return planDecodeClasses(caps, classes, work);
  static thread_local ClassPlanCache<DecodeClass> cache;
  if (!cache.matches(caps, 0u)) {
    std::vector<DecodeClass> cl;
    std::vector<uint32_t> wk;
    const bool split = planDecodeClasses(caps, &cl, &wk);
    cache.remember(caps, 0u, split, cl, wk);
  }
  if (!cache.split) return false;
  *classes = cache.classes;
  *work = cache.work;
  return true;
}

bool histAccumulates(uint32_t B, uint32_t maxBytes, bool raw) {
  return B <= kHistAccMaxBatch && histPartsAccFor(B, maxBytes) > histPartsFor(B, maxBytes, raw);
}

// Library-owned arrival counters for the histogram -> normalisation hand-off
// (HistFuse): 65536 u32 per (device, stream), zero at rest -- the kernel that uses
// them puts them back to zero.  Keyed by stream because calls on one stream are
// ordered while calls on different streams may overlap.
constexpr size_t kCounterWordsArrive = 65536;
constexpr size_t kCounterWordsAcc = (size_t)kAccElements * kNumSymbols;
constexpr size_t kCounterWordsSpill = 16384;  // flags of the hardware-dispatched encoders' spill-slot pool (kernels_encode.h: SpillPool)
int arrivalCounters(StreamLease& lease, uint32_t** out, uint32_t** acc, uint32_t** spillFlags = nullptr) {
  hipError_t e = hipSuccess;
  StreamState* s = lease.state(&e);
  if (!s) return fail(DGPU_ERR_HIP, std::string("stream state: ") + hipGetErrorString(e));
  if (!s->counters) {
    if (lease.capturing()) {
      return fail(DGPU_ERR_HIP, "HIP graph capture: the stream's hand-off counters do not exist yet and cannot be allocated while "
                                "the stream is being captured: run the same call once before capturing");
    }
    uint32_t* p = nullptr;
    const size_t words = kCounterWordsArrive + kCounterWordsAcc + kCounterWordsSpill;
    DGPU_HIP(hipMalloc((void**)&p, words * sizeof(uint32_t)));
    // once per (device, stream), ordered on the caller's stream ahead of the kernels that use the
    // counters (a plain hipMemset runs on the null stream, which non-blocking streams do not wait for)
    hipError_t me = hipMemsetAsync(p, 0, words * sizeof(uint32_t), lease.stream());
    if (me != hipSuccess) {
      (void)hipFree(p);
      return fail(DGPU_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(me));
    }
    s->counters = p;
  }
  *out = s->counters;
  *acc = s->counters + kCounterWordsArrive;
  if (spillFlags) *spillFlags = s->counters + kCounterWordsArrive + kCounterWordsAcc;
  return DGPU_OK;
}

// Cache policy of the histogram pass's input loads (format.h): non-temporal by default; ordinary (allocating) loads
// on request (dgpu_set_histogram_load_policy) for pipelines in which the codec's own dirty lines are what fills the
// memory-side cache when the pass starts.
std::atomic<int> g_histLoadPolicy{-1};  // -1: the compile-time default per input type, 0: non-temporal, 1: ordinary
bool histogramLoadsNonTemporal(uint32_t ft) {
  const int m = g_histLoadPolicy.load();
  (void)ft;
  return m < 0 ? kNtHistLoads : m == 0;
}

// Shared tail of every encode entry point: [checksum] -> histogram (+ fused
// normalisation) -> encode.  `in` holds raw bytes (floatType == 0: the ANS
// archive is the whole output) or float words (floatType != 0: the encoder
// splits them on the fly, the archive is a float archive).  No memset is needed
// on the common path: histogram workgroups store partial histograms, the last
// one of each element sums and normalises them and clears the tile descriptors +
// ticket for the encode kernel.
// (DGPU_TWO_LEVEL_LOOKBACK=0: the single level everywhere -- A/B runs and tests)
std::atomic<int> g_twoLevelLookback{[] {
  const char* e = getenv("DGPU_TWO_LEVEL_LOOKBACK");
  return e && *e ? atoi(e) : 1;
}()};
struct EncodeShared {
  uint32_t* checksumTemp = nullptr;  // [B] the batch's checksums (computed by the first class's call)
  uint4* table = nullptr;            // [B][256] encoder tables, indexed by the element's own index
  uint16_t* spill = nullptr;         // spill slots of the float encoders (every class's kernel is done with them when the next starts)
  size_t spillWords = 0;
};
int encodeCommon(
    TempArena& arena, StreamLease& lease, hipStream_t stream, int P, bool useChecksum, uint32_t B,
    const BatchView& in, const BatchView& archives, uint32_t floatType, uint32_t maxSize,
    const uint32_t* hist_dev /*may be null*/, uint32_t* outSize_dev,
    uint32_t outCapacity = 0xffffffffu /* bytes at every archive pointer; block data beyond it is dropped */,
    const RaggedPlan* plan = nullptr, const uint32_t* work_dev = nullptr /* the plan's lists on the device */,
    const EncodeClass* cls = nullptr /* one size class of the batch (its lists in work_dev); maxSize is the class's */,
    EncodeShared* shared = nullptr /* what the classes of one call share */) {
  const uint32_t wordBytes = floatType ? floatWordBytes(floatType) : 1u;
  const uint32_t tileBlocks = cls ? cls->tileBlocks : encTileBlocksFor(maxSize);
  const uint32_t maxTiles = cls ? std::max(1u, divUp(divUp(maxSize, kBlockSize), tileBlocks)) : tilesFor(maxSize);

  uint32_t* checksumTemp = shared ? shared->checksumTemp : nullptr;
  if (useChecksum && !checksumTemp) {
    DGPU_ALLOC(ck, uint32_t, arena, B);
    checksumTemp = ck;
    if (shared) shared->checksumTemp = ck;
    DGPU_HIP(hipMemsetAsync(checksumTemp, 0, (size_t)B * 4, stream));
    // Float quirk kept from the reference (GpuFloatCompress.cuh:466-468): the
    // size in float WORDS is consumed as a BYTE count by the checksum.
    dim3 grid(gridX(maxSize, 64 * 1024, 64), B);
    DGPU_LAUNCH("k_checksum", stream, k_checksum, grid, dim3(256), 0, stream, in, (const uint32_t*)nullptr, (const uint8_t*)nullptr, checksumTemp);
    DGPU_HIP(hipGetLastError());
  }


  // Encoder tables [B][256] x 16 bytes, normalisation -> encoder.  Not for batches of single-block elements: there
  // the table would be as many bytes as the element's symbols, and k_ans_encode_pair derives it from the pdf table in
  // the archive header instead.
  uint4* table = shared ? shared->table : nullptr;
  if (tileBlocks != kBlocksPerSingleTile && !table) {
    DGPU_ALLOC(tb, uint4, arena, (size_t)B * kNumSymbols);
    table = tb;
    if (shared) shared->table = tb;
  }
  // Work lists (descriptors and claim words for the tiles that exist only): a batch whose elements differ widely in
  // size (plan), or one size class of a batch (cls)
  const bool lists = cls ? tileBlocks != kBlocksPerSingleTile : (plan && plan->use && work_dev);
  const uint32_t numListedTiles = !lists ? 0u : (cls ? cls->numTiles : plan->numTiles);
  const uint32_t numListedHistParts = !lists ? 0u : (cls ? cls->numHistParts : plan->numHistParts);
  const uint32_t listedHistPartBytes = !lists ? 0u : (cls ? cls->histPartBytes : plan->histPartBytes);
  const uint32_t* tilesList = !lists ? nullptr : (cls ? work_dev + cls->tilesAt : work_dev);
  const uint32_t* histPartsList = !lists ? nullptr : (cls ? work_dev + cls->histAt : work_dev + (size_t)plan->numTiles);
  const uint32_t* tileBaseList = !lists ? nullptr : (cls ? work_dev + cls->tileBaseAt : work_dev + (size_t)plan->numTiles + plan->numHistParts);
  // (the single-block class of a batch: the elements to pair up)
  const uint32_t* elemMap = (cls && tileBlocks == kBlocksPerSingleTile) ? work_dev + cls->elemsAt : nullptr;
  const uint32_t numElems = elemMap ? cls->numElems : B;
  DGPU_ALLOC(tileDesc, uint64_t, arena, lists ? std::max<size_t>(numListedTiles, 1u) : (size_t)B * std::max(maxTiles, 1u));
  DGPU_ALLOC(claims, uint32_t, arena, lists ? std::max<size_t>(numListedTiles, 1u) : (size_t)B * std::max(maxTiles, 1u));
  // second level of the look-back for elements of more than 64 tiles (kernels_encode.h, lookBackTwoLevel): rectangles only
  const uint32_t lookbackGroups = (floatType != 0 && !lists && tileBlocks != kBlocksPerSingleTile && maxTiles > kLookbackGroup && g_twoLevelLookback.load() != 0)
      ? divUp(maxTiles, kLookbackGroup) : 0u;
  uint64_t* groupWords = nullptr;
  if (lookbackGroups) {
    DGPU_ALLOC(gw, uint64_t, arena, (size_t)B * lookbackGroups * (kGroupArriveStride + 1u));
    groupWords = gw;
  }

  // The encoder's grid.  `resident` = the workgroups of the kernel that fit on the chip at once.  8-block float tiles
  // run as `resident` persistent workgroups that walk the tickets with a static map; raw bytes and float tiles of 2 / 4
  // blocks run one workgroup per tile when there are more tiles than that, dispatched by the hardware in ticket order
  // (encoderHardwareDispatch); k_ans_encode_pair always runs one workgroup per pair.  Spill slots (float inputs):
  // [resident][slots per workgroup] -- a persistent workgroup's own, or a pool handed out through spillFlags.
  const uint32_t numTickets = lists ? numListedTiles : (elemMap ? numElems : B * maxTiles);
  const bool wideStage = encoderWideStage(floatType, tileBlocks, maxTiles);
  const uint32_t resident = maxTiles > 0 ? encodeGrid(P, floatType, tileBlocks, numTickets, wideStage) : 0u;
  const bool hwDispatch = tileBlocks != kBlocksPerSingleTile && encoderHardwareDispatch(numTickets, resident, floatType, tileBlocks);
  uint16_t* spill = nullptr;
  uint32_t* spillFlags = nullptr;
  uint32_t spillPairs = 0;
  if (maxTiles > 0 && encodeSpills(floatType)) {
    // (single-block batches: two slots per workgroup, one per element of its pair)
    const uint32_t slotsPerWg = tileBlocks == kBlocksPerSingleTile ? 2u : tileBlocks;
    // (the size classes of one call run one after the other on the stream: they share the region of the first one
    // that is large enough)
    const size_t spillWords = (size_t)resident * slotsPerWg * encSpillSlotWords(P);
    if (shared && shared->spill && shared->spillWords >= spillWords) {
      spill = shared->spill;
    } else {
      DGPU_ALLOC(sp, uint16_t, arena, spillWords);
      spill = sp;
      if (shared) shared->spill = sp, shared->spillWords = spillWords;
    }
    if (tileBlocks == kBlocksPerSingleTile || hwDispatch) {
      // one workgroup per pair / tile: the slots are a POOL with a pair for every wavefront that can be resident,
      // handed out through library-owned flags that are zero at rest
      uint32_t *arrive = nullptr, *acc = nullptr;
      int rc = arrivalCounters(lease, &arrive, &acc, &spillFlags);
      if (rc) return rc;
      spillPairs = resident * slotsPerWg / 2u;
      DGPU_REQUIRE(spillPairs <= kCounterWordsSpill, "more resident encoder wavefronts than spill-pool flags");
    }
  }

  NormalizeArgs n;
  n.sizes = in;
  n.hist = hist_dev;
  n.histAcc = nullptr;
  n.histParts = 1;
  n.probBits = P;
  n.encTable = table;
  n.refTable = nullptr;
  n.out = archives;
  n.writeHeader = 1;
  n.floatType = floatType;
  // ANS-level checksums are not used in float mode (GpuFloatCodec.h:50)
  n.useChecksum = (useChecksum && !floatType) ? 1 : 0;
  n.checksum = (useChecksum && !floatType) ? checksumTemp : nullptr;
  n.outSize = outSize_dev;
  n.floatUseChecksum = (useChecksum && floatType) ? 1 : 0;
  n.tileDesc = tileDesc;
  n.maxTiles = maxTiles;
  n.claims = claims;
  n.numInBatch = B;
  n.tileBase = tileBaseList;
  n.tileSymbols = tileBlocks * kBlockSize;
  n.groupWords = groupWords;
  n.groupWordsPerElement = lookbackGroups * (kGroupArriveStride + 1u);

  if (!hist_dev && tileBlocks == kBlocksPerSingleTile && maxTiles > 0 && floatType != kFloat32) {
    // batches of single-block elements: one wavefront counts and normalises an element (kernels_pairs.h); no partial
    // histograms, no arrival counters.  (Measured on 32768 elements, profiles/r04_ab_single_block_elements.txt:
    // bf16 75.5 -> 65.5 us, fp16 80.5 -> 73.7; float32 -- 16 bytes of input per symbol and lane -- 69 -> 75.5, so
    // float32 keeps the workgroup per element.)
    const dim3 grid(divUp(numElems, kSingleStatWaves)), block(64u * kSingleStatWaves);
#define DGPU_STATS_SINGLE(FT)                                                                                           \
    if (histogramLoadsNonTemporal(floatType)) {                                                                         \
      DGPU_LAUNCH("k_stats_single", stream, (k_stats_single<FT, true>), grid, block, 0, stream, in, n, elemMap, numElems); \
    } else {                                                                                                            \
      DGPU_LAUNCH("k_stats_single", stream, (k_stats_single<FT, false>), grid, block, 0, stream, in, n, elemMap, numElems); \
    }
    switch (floatType) {
      case 0: DGPU_STATS_SINGLE(0u) break;
      case kFloat16: DGPU_STATS_SINGLE(kFloat16) break;
      default: DGPU_STATS_SINGLE(kBFloat16) break;
    }
#undef DGPU_STATS_SINGLE
    DGPU_HIP(hipGetLastError());
  } else if (!hist_dev) {
    const bool histList = lists && numListedHistParts != 0;
    const bool accumulate = !histList && histAccumulates(B, maxSize * wordBytes, floatType == 0);
    dim3 grid(accumulate ? histPartsAccFor(B, maxSize * wordBytes) : histPartsFor(B, maxSize * wordBytes, floatType == 0), B);
    if (histList) grid = dim3(numListedHistParts);  // one workgroup per listed (element, part)
    uint32_t* histTemp = nullptr;
    if (!accumulate) {
      DGPU_ALLOC(ht, uint32_t, arena, (size_t)(histList ? 1u : B) * grid.x * kNumSymbols);
      histTemp = ht;
    }
    HistFuse fuse;
    fuse.workMap = histList ? histPartsList : nullptr;
    fuse.partBytes = histList ? listedHistPartBytes : 0u;
    uint32_t* acc = nullptr;
    int rc = arrivalCounters(lease, &fuse.arrive, &acc);
    if (rc) return rc;
    fuse.acc = accumulate ? acc : nullptr;
    n.hist = histTemp;
    n.histAcc = fuse.acc;
    n.histParts = histList ? 1u : grid.x;
    fuse.norm = n;
    // bins with 32 lane slots unless a workgroup sees too little data to pay for zeroing / folding them
    const bool smallBins = histList ? listedHistPartBytes <= 64u * 1024u : (uint64_t)maxSize * wordBytes / grid.x <= 64u * 1024u;
#define DGPU_HIST_LAUNCH_NT(S, NT)                                                                                \
    switch (floatType) {                                                                                          \
      case 0:                                                                                                     \
        DGPU_LAUNCH("k_histogram", stream, (k_histogram<S, NT>), grid, dim3(256), 0, stream, in, histTemp, 1u, fuse); \
        break;                                                                                                    \
      case kFloat16:                                                                                              \
        DGPU_LAUNCH("k_float_histogram", stream, (k_float_histogram<kFloat16, S, NT>), grid, dim3(256), 0, stream, in, histTemp, 1u, fuse); \
        break;                                                                                                    \
      case kBFloat16:                                                                                             \
        DGPU_LAUNCH("k_float_histogram", stream, (k_float_histogram<kBFloat16, S, NT>), grid, dim3(256), 0, stream, in, histTemp, 1u, fuse); \
        break;                                                                                                    \
      default:                                                                                                    \
        DGPU_LAUNCH("k_float_histogram", stream, (k_float_histogram<kFloat32, S, NT>), grid, dim3(256), 0, stream, in, histTemp, 1u, fuse); \
        break;                                                                                                    \
    }
#define DGPU_HIST_LAUNCH(S)                 \
    if (histogramLoadsNonTemporal(floatType)) { \
      DGPU_HIST_LAUNCH_NT(S, true)          \
    } else {                                \
      DGPU_HIST_LAUNCH_NT(S, false)         \
    }
    if (smallBins) {
      DGPU_HIST_LAUNCH(kHistSlotsSmall)
    } else {
      DGPU_HIST_LAUNCH(kHistSlotsLarge)
    }
#undef DGPU_HIST_LAUNCH
#undef DGPU_HIST_LAUNCH_NT
    DGPU_HIP(hipGetLastError());
  } else {
    // caller-supplied histogram: stand-alone normalisation
    DGPU_LAUNCH("k_normalize", stream, k_normalize, dim3(B), dim3(256), 0, stream, n);
    DGPU_HIP(hipGetLastError());
  }
  if (maxTiles > 0) {
    // (k_ans_encode_pair: one workgroup per pair of elements; numTickets counts elements there)
    const uint32_t grid = tileBlocks == kBlocksPerSingleTile ? (numTickets + 1u) / 2u : (hwDispatch ? numTickets : resident);
    EncodeArgs e;
    e.in = in;
    e.out = archives;
    e.encTable = table;
    e.maxTiles = maxTiles;
    e.numInBatch = B;
    e.numTickets = numTickets;
    e.workMap = lists ? tilesList : elemMap;
    e.tileDesc = tileDesc;
    e.claims = claims;
    e.groupWords = groupWords;
    e.groupsPerElement = lookbackGroups;
    e.absentModulo = absentWorkgroupModulo();
    {
      // tiles of ONE element that are in flight at once: the resident workgroups over the elements of this launch
      const uint32_t elems = lists ? std::max(1u, numTickets / std::max(maxTiles, 1u)) : std::max(numElems, 1u);
      e.pollLong = std::min(maxTiles, std::max(resident, 1u) / elems) > 12u ? 1u : 0u;
    }
    e.spill = spill;
    e.spillFlags = spillFlags;
    e.spillPairs = spillPairs;
    e.outSize = outSize_dev;
    e.outCapacity = outCapacity;
    e.useChecksum = (useChecksum && floatType) ? 1 : 0;
    e.checksum = (useChecksum && floatType) ? checksumTemp : nullptr;
    int rc = launchEncode(P, floatType, e, tileBlocks, grid, hwDispatch, wideStage, stream);
    if (rc) return rc;
  }
  return DGPU_OK;
}

int ansEncodeImpl(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int P, int useChecksum, uint32_t B,
    const HostParams* hp /*null => stride views below*/, const BatchView* strideIn,
    const BatchView* strideOut, uint32_t maxSize, const uint32_t* histogram_dev,
    uint32_t* outSize_dev, hipStream_t stream) {
  DGPU_REQUIRE(validProbBits(P), "probBits must be 9, 10 or 11");
  DGPU_REQUIRE(B <= 65535u, "numInBatch must be <= 65535");
  DGPU_REQUIRE(encodableSize(maxSize), "input larger than 1717538816 bytes: its maximum compressed size exceeds INT32_MAX (GpuANSEncode.cu:22)");
  if (tempUsed) *tempUsed = 0;
  if (B == 0) return DGPU_OK;

  StreamLease streamLease(stream);
  TempArena arena(temp_dev, tempBytes, streamLease);
  ParamLease lease;
  BatchView in, out;
  RaggedPlan plan;
  std::vector<EncodeClass> classes;
  const uint32_t* work_dev = nullptr;
  if (hp && asStrideViews(*hp, false, &in, &out)) {
    // (nothing to upload)
  } else if (hp) {
    const uint64_t *inP = nullptr, *outP = nullptr;
    const uint32_t* sz = nullptr;
    HostParams listed;
    const HostParams* up = hp;
    if ((histogram_dev == nullptr && planEncodeClassesCached(hp->sizes, 0u, &classes, &listed.work)) ||
        planEncode(hp->sizes, 0u, maxSize, histogram_dev == nullptr, &plan, &listed.work)) {
      listed.inPtrs = hp->inPtrs, listed.outPtrs = hp->outPtrs, listed.sizes = hp->sizes;
      up = &listed;
    }
    int rc = uploadParams(lease, stream, *up, &inP, &outP, &sz, nullptr, &work_dev);
    if (rc) return rc;
    in = viewPointers(inP, sz, 0);
    out = viewPointers(outP, nullptr, 0);
  } else {
    in = *strideIn;
    out = *strideOut;
  }
  if (!classes.empty()) {
    EncodeShared shared;
    int rc = DGPU_OK;
    for (const EncodeClass& c : classes) {
      rc = encodeCommon(arena, streamLease, stream, P, useChecksum != 0, B, in, out, 0, c.maxSize, nullptr, outSize_dev, 0xffffffffu, nullptr,
                        work_dev, &c, &shared);
      if (rc) break;
    }
    if (tempUsed) *tempUsed = arena.requested();
    return rc;
  }
  int rc = encodeCommon(arena, streamLease, stream, P, useChecksum != 0, B, in, out, 0, maxSize, histogram_dev, outSize_dev, 0xffffffffu,
                        &plan, work_dev);
  if (tempUsed) *tempUsed = arena.requested();
  return rc;
}

int floatCompressImpl(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t ft, int P, int useChecksum,
    uint32_t B, const HostParams& hp, uint32_t maxSize, uint32_t* outSize_dev,
    hipStream_t stream) {
  DGPU_REQUIRE(validProbBits(P), "probBits must be 9, 10 or 11");
  DGPU_REQUIRE(validFloatType(ft), "floatType must be float16, bfloat16 or float32");
  DGPU_REQUIRE(B <= 65535u, "numInBatch must be <= 65535");
  DGPU_REQUIRE(encodableSize(maxSize), "tensor larger than 1717538816 words: the maximum compressed size of its exponent plane exceeds INT32_MAX (GpuANSEncode.cu:22)");
  if (tempUsed) *tempUsed = 0;
  if (B == 0) return DGPU_OK;

  StreamLease streamLease(stream);
  TempArena arena(temp_dev, tempBytes, streamLease);
  ParamLease lease;
  BatchView in, out;
  int rc = DGPU_OK;
  RaggedPlan plan;
  const uint32_t* work_dev = nullptr;
  std::vector<EncodeClass> classes;
  if (!asStrideViews(hp, false, &in, &out)) {
    const uint64_t *inP = nullptr, *outP = nullptr;
    const uint32_t* sz = nullptr;
    HostParams listed;
    const HostParams* up = &hp;
    if (planEncodeClassesCached(hp.sizes, ft, &classes, &listed.work) || planEncode(hp.sizes, ft, maxSize, true, &plan, &listed.work)) {
      listed.inPtrs = hp.inPtrs, listed.outPtrs = hp.outPtrs, listed.sizes = hp.sizes;
      up = &listed;
    }
    rc = uploadParams(lease, stream, *up, &inP, &outP, &sz, nullptr, &work_dev);
    if (rc) return rc;
    in = viewPointers(inP, sz, 0);
    out = viewPointers(outP, nullptr, 0);
  }

  // No exponent plane in temp memory: the encoder splits the float words itself.
  if (!classes.empty()) {
    // every size class on the kernels of its own geometry, one after the other (EncodeClass)
    EncodeShared shared;
    for (const EncodeClass& c : classes) {
      rc = encodeCommon(arena, streamLease, stream, P, useChecksum != 0, B, in, out, ft, c.maxSize, nullptr, outSize_dev, 0xffffffffu, nullptr,
                        work_dev, &c, &shared);
      if (rc) break;
    }
    if (tempUsed) *tempUsed = arena.requested();
    return rc;
  }
  rc = encodeCommon(arena, streamLease, stream, P, useChecksum != 0, B, in, out, ft, maxSize, nullptr, outSize_dev, 0xffffffffu, &plan,
                    work_dev);
  if (tempUsed) *tempUsed = arena.requested();
  return rc;
}

// Order of k_ans_decode's workgroups (kernels_decode.h: decodeTileOf).  Measured on MI355X, cold round trip
// (profiles/r05_ab_decoder_order.txt): with the tiles of an element back to back, eight consecutive workgroups -- one per
// XCD -- read one archive and write one output row; letting every XCD walk its own elements spreads a moment's traffic
// over eight times as many rows: 256 x 512 Ki bf16 decode 107 -> 102 us (step -1.7 %), fp16 -3.5 %, 64 x 2 Mi -4 %,
// Zipf bytes -1 %.  It needs enough elements to keep the eight XCDs level (16 x 8 Mi: +17 % for the decoder alone, a
// batch of one: everything on one XCD), so small batches keep the element-major order.  -1 = this policy;
// dgpu_debug_set_decoder_order / DGPU_DEC_ORDER force an order (tests, A/B runs).
// blocks per decoder tile for a batch whose largest capacity has `maxBlocks` blocks: elements of up to 8 blocks:
// 4-block workgroups, of up to 2 blocks: one wavefront (see kDecBlocksPerSmallTile)
uint32_t decTileBlocksFor(uint32_t maxBlocks) {
  return maxBlocks <= 1u ? kDecBlocksPerSingleTile
      : maxBlocks <= 2u  ? kDecBlocksPerTinyTile
      : maxBlocks <= 8u  ? kDecBlocksPerSmallTile
                         : kDecBlocksPerTile;
}
std::atomic<int> g_decOrder{[] {
  const char* e = getenv("DGPU_DEC_ORDER");
  return e && *e ? atoi(e) : -1;
}()};
uint32_t decodeOrder(uint32_t B) {
  const int forced = g_decOrder.load();
  if (forced >= 0 && forced <= (int)kDecOrderXcd) return (uint32_t)forced;
  return B >= 64u ? kDecOrderXcd : kDecOrderElementMajor;
}

template <int P, uint32_t FT>
int launchDecodePF(const DecodeArgs& a, uint32_t tileBlocks, dim3 grid, hipStream_t stream) {
  if (tileBlocks == kDecBlocksPerSingleTile) {
    // every capacity <= 4096 symbols: two elements per wavefront (kernels_pairs.h)
    const uint32_t elems = (a.order == kDecOrderMap && a.workMap) ? a.numListed : a.numInBatch;  // (a size class: its elements)
    DGPU_LAUNCH("k_ans_decode_pair", stream, (k_ans_decode_pair<P, FT>), dim3((elems + 1u) / 2u), dim3(64), decPairLdsBytes(P, FT),
                stream, a);
  } else if (tileBlocks == kDecBlocksPerTinyTile) {
    DGPU_LAUNCH("k_ans_decode", stream, (k_ans_decode<P, FT, kDecBlocksPerTinyTile>), grid, dim3(kDecBlocksPerTinyTile * 32u),
                decLdsBytes(P, FT, kDecBlocksPerTinyTile), stream, a);
  } else if (tileBlocks == kDecBlocksPerSmallTile) {
    DGPU_LAUNCH("k_ans_decode", stream, (k_ans_decode<P, FT, kDecBlocksPerSmallTile>), grid, dim3(kDecBlocksPerSmallTile * 32u),
                decLdsBytes(P, FT, kDecBlocksPerSmallTile), stream, a);
  } else {
    DGPU_LAUNCH("k_ans_decode", stream, (k_ans_decode<P, FT, kDecBlocksPerTile>), grid, dim3(kDecBlocksPerTile * 32u),
                decLdsBytes(P, FT, kDecBlocksPerTile), stream, a);
  }
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}

template <uint32_t FT>
int launchDecodeF(int P, const DecodeArgs& a, uint32_t tileBlocks, dim3 grid, hipStream_t stream) {
  switch (P) {
    case 9: return launchDecodePF<9, FT>(a, tileBlocks, grid, stream);
    case 10: return launchDecodePF<10, FT>(a, tileBlocks, grid, stream);
    default: return launchDecodePF<11, FT>(a, tileBlocks, grid, stream);
  }
}

int decodeImpl(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t ft, int P, int useChecksum,
    uint32_t B, const HostParams* hp, const BatchView* strideIn, const BatchView* strideOut,
    uint32_t maxCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, hipStream_t stream,
    int32_t* errBatch, uint32_t uniformInBytes = 0 /* stride batches: bytes available per archive (0 = unknown) */) {
  DGPU_REQUIRE(validProbBits(P), "probBits must be 9, 10 or 11");
  DGPU_REQUIRE(ft == 0 || validFloatType(ft), "bad floatType");
  DGPU_REQUIRE(B <= 65535u, "numInBatch must be <= 65535");
  if (tempUsed) *tempUsed = 0;
  if (errBatch) *errBatch = -1;
  g_mismatches.clear();
  if (B == 0) return DGPU_OK;
  if (useChecksum && streamIsCapturing(stream)) {
    // the comparison is host work behind a stream synchronise (as upstream, GpuANSDecode.cuh:557-591): it would
    // invalidate the capture, and a replay could never repeat it
    return fail(DGPU_ERR_HIP, "HIP graph capture: checksum verification on decode copies to the host and synchronises the "
                              "stream; it cannot be captured into a HIP graph (decode with useChecksum = 0 under capture)");
  }

  StreamLease streamLease(stream);
  TempArena arena(temp_dev, tempBytes, streamLease);
  ParamLease lease;
  BatchView in, out;
  const uint32_t* inBytes_dev = nullptr;
  const uint32_t* work_dev = nullptr;
  uint32_t numListedTiles = 0;
  std::vector<DecodeClass> classes;
  if (hp && asStrideViews(*hp, true, &in, &out)) {
    // (nothing to upload)
  } else if (hp) {
    const uint64_t *inP = nullptr, *outP = nullptr;
    const uint32_t* cap = nullptr;
    // (capacities that differ widely: only the tiles inside each element's capacity are launched, see RaggedPlan)
    HostParams listed;
    const HostParams* up = hp;
    if (planDecodeClassesCached(hp->sizes, &classes, &listed.work)) {
      listed.inPtrs = hp->inPtrs, listed.outPtrs = hp->outPtrs, listed.sizes = hp->sizes, listed.inBytes = hp->inBytes;
      up = &listed;
    } else {
      const uint32_t blocks = divUp(maxCapacity, kBlockSize);
      const uint32_t tb = decTileBlocksFor(blocks);
      if (tb != kDecBlocksPerSingleTile && planTileList(hp->sizes, tb * kBlockSize, std::max(1u, divUp(blocks, tb)), 1u, &listed.work)) {
        listed.inPtrs = hp->inPtrs, listed.outPtrs = hp->outPtrs, listed.sizes = hp->sizes, listed.inBytes = hp->inBytes;
        up = &listed;
        numListedTiles = (uint32_t)listed.work.size();
      }
    }
    int rc = uploadParams(lease, stream, *up, &inP, &outP, &cap, &inBytes_dev, &work_dev);
    if (rc) return rc;
    in = viewPointers(inP, nullptr, 0);
    out = viewPointers(outP, cap, 0);
  } else {
    in = *strideIn;
    out = *strideOut;
  }

  uint32_t* sizesForChecksum = outSize_dev;
  uint8_t* successForChecksum = outSuccess_dev;
  if (useChecksum) {
    if (!sizesForChecksum) {
      DGPU_ALLOC(s, uint32_t, arena, B);
      sizesForChecksum = s;
    }
    if (!successForChecksum) {
      DGPU_ALLOC(s, uint8_t, arena, B);
      successForChecksum = s;
    }
  }

  const uint32_t maxBlocks = divUp(maxCapacity, kBlockSize);
  const uint32_t tileBlocks = decTileBlocksFor(maxBlocks);
  const uint32_t maxTiles = std::max(1u, divUp(maxBlocks, tileBlocks));
  // every size class of the batch on the decoder of its own geometry, one after the other (DecodeClass)
  for (const DecodeClass& c : classes) {
    DecodeArgs d;
    d.in = in;
    d.out = out;
    d.floatType = ft;
    d.outSuccess = useChecksum ? successForChecksum : outSuccess_dev;
    d.outSize = useChecksum ? sizesForChecksum : outSize_dev;
    d.inBytes = inBytes_dev;
    d.uniformInBytes = uniformInBytes;
    d.numInBatch = B;
    d.maxTiles = std::max(1u, divUp(c.maxBlocks, c.tileBlocks));
    d.order = kDecOrderMap;
    d.workMap = work_dev + (c.tileBlocks == 1u ? c.elemsAt : c.tilesAt);
    d.numListed = c.numElems;
    const dim3 grid(std::max(c.numTiles, 1u));
    int rc;
    if (ft == 0) rc = launchDecodeF<0>(P, d, c.tileBlocks, grid, stream);
    else if (ft == kFloat16) rc = launchDecodeF<kFloat16>(P, d, c.tileBlocks, grid, stream);
    else if (ft == kBFloat16) rc = launchDecodeF<kBFloat16>(P, d, c.tileBlocks, grid, stream);
    else rc = launchDecodeF<kFloat32>(P, d, c.tileBlocks, grid, stream);
    if (rc) return rc;
  }
  if (classes.empty()) {
    DecodeArgs d;
    d.in = in;
    d.out = out;
    d.floatType = ft;
    d.outSuccess = useChecksum ? successForChecksum : outSuccess_dev;
    d.outSize = useChecksum ? sizesForChecksum : outSize_dev;
    d.inBytes = inBytes_dev;
    d.uniformInBytes = uniformInBytes;
    d.numInBatch = B;
    d.maxTiles = maxTiles;
    d.order = decodeOrder(B);
    d.workMap = nullptr;
    d.numListed = 0;
    dim3 grid((d.order == kDecOrderXcd ? roundUp(B, 8u) : B) * maxTiles);
    if (work_dev && numListedTiles) {
      d.order = kDecOrderMap;
      d.workMap = work_dev;
      grid = dim3(numListedTiles);
    }
    int rc;
    if (ft == 0) rc = launchDecodeF<0>(P, d, tileBlocks, grid, stream);
    else if (ft == kFloat16) rc = launchDecodeF<kFloat16>(P, d, tileBlocks, grid, stream);
    else if (ft == kBFloat16) rc = launchDecodeF<kBFloat16>(P, d, tileBlocks, grid, stream);
    else rc = launchDecodeF<kFloat32>(P, d, tileBlocks, grid, stream);
    if (rc) return rc;
  }

  int status = DGPU_OK;
  if (useChecksum) {
    // checksum the decoded data, fetch the archived checksum, compare on the
    // host (GpuANSDecode.cuh:557-591, GpuFloatDecompress.cuh:699-733).  The
    // decoded size (bytes, or -- float quirk -- float words used as a byte
    // count) bounds the checksummed range.
    DGPU_ALLOC(sums, uint32_t, arena, 2 * (size_t)B);
    DGPU_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)B * 4, stream));
    dim3 grid(gridX(maxCapacity * (ft ? floatWordBytes(ft) : 1u), 64 * 1024, 64), B);
    hipLaunchKernelGGL(k_checksum, grid, dim3(256), 0, stream, out, (const uint32_t*)sizesForChecksum,
                       (const uint8_t*)successForChecksum, sums);
    DGPU_HIP(hipGetLastError());
    if (ft) {
      hipLaunchKernelGGL(k_float_info, dim3(divUp(B, 128)), dim3(128), 0, stream, in, B,
                         (uint32_t*)nullptr, (uint32_t*)nullptr, sums + B);
    } else {
      hipLaunchKernelGGL(k_ans_info, dim3(divUp(B, 128)), dim3(128), 0, stream, in, B,
                         (uint32_t*)nullptr, sums + B);
    }
    DGPU_HIP(hipGetLastError());
    std::vector<uint32_t> h(2 * (size_t)B);
    std::vector<uint8_t> ok(B);
    DGPU_HIP(hipMemcpyAsync(h.data(), sums, h.size() * 4, hipMemcpyDeviceToHost, stream));
    DGPU_HIP(hipMemcpyAsync(ok.data(), successForChecksum, B, hipMemcpyDeviceToHost, stream));
    DGPU_HIP(hipStreamSynchronize(stream));
    // EVERY mismatching member is reported, as upstream pushes every one into errorInfo; the message is the
    // reference's, one line per member (its stringstream is never reset, so the text accumulates)
    std::string msg;
    for (uint32_t i = 0; i < B; ++i) {
      if (ok[i] && h[i] != h[B + i]) {
        char buf[160];
        snprintf(buf, sizeof(buf),
                 "Checksum mismatch in batch member %u: expected checksum %x got %x\n", i,
                 h[B + i], h[i]);
        msg += buf;
        g_mismatches.push_back({(int32_t)i, h[B + i], h[i]});
        if (status == DGPU_OK && errBatch) *errBatch = (int32_t)i;
        status = DGPU_ERR_CHECKSUM_MISMATCH;
      }
    }
    if (status != DGPU_OK) g_lastError = msg;
  }
  if (tempUsed) *tempUsed = arena.requested();
  return status;
}

int splitSizesToPointers(
    const void* base, const uint32_t* splitSizes, uint32_t B, uint32_t wordBytes,
    std::vector<uint64_t>* ptrs, std::vector<uint32_t>* sizes, uint32_t* maxSize) {
  ptrs->resize(B);
  sizes->resize(B);
  uint64_t prefix = 0;
  uint32_t mx = 0;
  for (uint32_t i = 0; i < B; ++i) {
    (*ptrs)[i] = (uint64_t)(uintptr_t)base + prefix * wordBytes;
    (*sizes)[i] = splitSizes[i];
    prefix += splitSizes[i];
    mx = std::max(mx, splitSizes[i]);
  }
  *maxSize = mx;
  return DGPU_OK;
}

}  // namespace

// ===========================================================================
// extern "C" surface
// ===========================================================================
extern "C" {

const char* dgpu_version(void) { return "dietgpu_amd 0.1 (gfx950)"; }
uint32_t dgpu_abi_version(void) { return DGPU_ABI_VERSION; }
const char* dgpu_last_error(void) { return g_lastError.c_str(); }

uint32_t dgpu_last_checksum_mismatches(int32_t* batchIdx, uint32_t* expected, uint32_t* got, uint32_t cap) {
  const uint32_t n = (uint32_t)g_mismatches.size();
  for (uint32_t i = 0; i < n && i < cap; ++i) {
    if (batchIdx) batchIdx[i] = g_mismatches[i].batch;
    if (expected) expected[i] = g_mismatches[i].expected;
    if (got) got[i] = g_mismatches[i].got;
  }
  return n;
}

static std::atomic<uint32_t> g_absentModulo{0};
}  // extern "C"
namespace {
uint32_t absentWorkgroupModulo() { return g_absentModulo.load(); }
}  // namespace
extern "C" {
void dgpu_debug_set_absent_workgroups(uint32_t modulo) { g_absentModulo.store(modulo); }
void dgpu_debug_set_encoder_dispatch(int mode) { g_encDispatch.store(mode < 0 ? -1 : (mode != 0)); }
void dgpu_debug_set_decoder_order(int order) { g_decOrder.store(order); }
void dgpu_debug_set_param_cache(int on) { g_paramCacheEnabled.store(on != 0); }
void dgpu_debug_set_work_lists(int mode) { g_workLists.store(mode < 0 ? -1 : (mode ? 1 : 0)); }
void dgpu_debug_set_size_classes(int mode) { g_sizeClasses.store(mode < 0 ? -1 : (mode ? 1 : 0)); }
void dgpu_set_histogram_load_policy(int mode) { g_histLoadPolicy.store(mode < 0 ? -1 : (mode != 0)); }
int dgpu_release_graph_state(void) {
  const int n = paramCache().releaseGraphPins();
  return n + streamRegistry().release(nullptr, true, true);
}

int dgpu_release_stream_state(void* stream) { return streamRegistry().release((hipStream_t)stream, false); }
int dgpu_release_all_stream_state(void) { return streamRegistry().release(nullptr, true); }
uint32_t dgpu_debug_stream_state_count(void) { return (uint32_t)streamRegistry().size(); }

void dgpu_prof_enable(int on) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  p.enabled = on != 0;
}

void dgpu_prof_reset(void) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto& s : p.open) {
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  p.open.clear();
  p.acc.clear();
}

int dgpu_prof_summary(char* buf, size_t cap) {
  ProfState& p = prof();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto& s : p.open) {
    float ms = 0.f;
    if (hipEventSynchronize(s.stop) == hipSuccess && hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
      auto& a = p.acc[s.name];
      a.first += 1;
      a.second += ms;
    }
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  p.open.clear();
  std::string out = "{";
  bool first = true;
  for (auto& kv : p.acc) {
    char line[256];
    snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %llu, \"total_ms\": %.6f}", first ? "" : ", ",
             kv.first.c_str(), (unsigned long long)kv.second.first, kv.second.second);
    out += line;
    first = false;
  }
  out += "}";
  if (out.size() + 1 > cap) return -1;
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

uint32_t dgpu_ans_max_compressed_size(uint32_t bytes) { return maxCompressedSizeHost(bytes); }

uint32_t dgpu_float_max_compressed_size(uint32_t ft, uint32_t n) {
  const uint32_t ans = maxCompressedSizeHost(n);
  if (ans == 0u) return 0u;  // beyond the reference's INT32_MAX guard
  const uint64_t total = 16ull + ans + (ft == kFloat32 ? 2ull * roundUp(n, 8u) + roundUp(n, 16u) : (uint64_t)roundUp(n, 16u));
  return total > 0xffffffffull ? 0u : (uint32_t)total;
}

static size_t encodeTempBytes(uint32_t B, uint32_t maxBytes, uint32_t wordBytes, bool spills) {
  size_t tiles = std::max(tilesFor(maxBytes), 1u);
  size_t parts = histPartsFor(B, maxBytes * wordBytes, true);
  size_t t = 0;
  t += alignUp((size_t)B * 4, kTempAlign);                                        // checksums
  // partial histograms: the rectangle, or the list of a batch whose elements differ widely in size (planHistList)
  t += alignUp(std::max((size_t)B * parts, (size_t)B + kHistTargetWgsForListsRaw + 16u) * kNumSymbols * 4, kTempAlign);
  t += alignUp((size_t)B * kNumSymbols * 16, kTempAlign);                         // encoder table
  t += alignUp((size_t)B * tiles * 8, kTempAlign);                                // tile descriptors
  t += alignUp((size_t)B * tiles * 4, kTempAlign);                                // tile claim words
  if (tiles > kLookbackGroup) t += alignUp((size_t)B * divUp((uint32_t)tiles, kLookbackGroup) * (kGroupArriveStride + 1u) * 8, kTempAlign);  // look-back groups
  if (spills) {
    // spill slots of the persistent encoder workgroups (bounded by what fits on the chip)
    size_t perCu = (160u * 1024u) / encLdsBytes(9, true, kBFloat16, kBlocksPerTile);
    size_t grid = std::min((size_t)B * tiles, perCu * numComputeUnits());
    t += alignUp(grid * kBlocksPerTile * encSpillSlotWords(11) * 2, kTempAlign);
  }
  return t + kTempAlign;
}

size_t dgpu_ans_encode_temp_bytes(uint32_t B, uint32_t maxBytes) {
  return encodeTempBytes(B, maxBytes, 1, encodeSpills(0));
}

size_t dgpu_ans_decode_temp_bytes(uint32_t B, uint32_t maxBytes, int probBits) {
  (void)maxBytes;
  (void)probBits;  // the decode LUT is built in LDS by each workgroup
  size_t t = 0;
  t += 3 * alignUp((size_t)B * 8, kTempAlign);  // checksum verification scratch
  return t + kTempAlign;
}

size_t dgpu_float_compress_temp_bytes(uint32_t ft, uint32_t B, uint32_t maxFloats) {
  // no exponent plane: the split is fused into the encoder
  return encodeTempBytes(B, maxFloats, validFloatType(ft) ? floatWordBytes(ft) : 4u, true);
}

size_t dgpu_float_decompress_temp_bytes(uint32_t ft, uint32_t B, uint32_t maxFloats, int probBits) {
  (void)ft;
  return dgpu_ans_decode_temp_bytes(B, maxFloats, probBits);
}

// ---- encode ----------------------------------------------------------------
int dgpu_ans_encode_batch_stride(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inPerBatchSize, uint32_t inPerBatchStride,
    const uint32_t* histogram_dev, void* out_dev, uint32_t outPerBatchStride,
    uint32_t* outSize_dev, void* stream) {
  DGPU_REQUIRE(((uintptr_t)in_dev % DGPU_ANS_REQUIRED_ALIGNMENT) == 0 &&
                   (numInBatch <= 1 || inPerBatchStride % DGPU_ANS_REQUIRED_ALIGNMENT == 0),
               "ANS input must be 4-byte aligned");
  DGPU_REQUIRE(((uintptr_t)out_dev % 16) == 0 && (numInBatch <= 1 || outPerBatchStride % 16 == 0),
               "compressed output must be 16-byte aligned");
  BatchView in = viewStride(in_dev, inPerBatchStride, inPerBatchSize);
  BatchView out = viewStride(out_dev, outPerBatchStride, 0);
  return ansEncodeImpl(temp_dev, tempBytes, tempUsed, probBits, useChecksum, numInBatch, nullptr,
                       &in, &out, inPerBatchSize, histogram_dev, outSize_dev, (hipStream_t)stream);
}

int dgpu_ans_encode_batch_pointer(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inSize,
    const uint32_t* histogram_dev, void* const* out, uint32_t* outSize_dev, void* stream) {
  HostParams hp;
  hp.inPtrs.resize(numInBatch);
  hp.outPtrs.resize(numInBatch);
  hp.sizes.resize(numInBatch);
  uint32_t maxSize = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) {
    DGPU_REQUIRE(((uintptr_t)in[i] % DGPU_ANS_REQUIRED_ALIGNMENT) == 0, "ANS input must be 4-byte aligned");
    DGPU_REQUIRE(((uintptr_t)out[i] % 16) == 0, "compressed output must be 16-byte aligned");
    hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
    hp.outPtrs[i] = (uint64_t)(uintptr_t)out[i];
    hp.sizes[i] = inSize[i];
    maxSize = std::max(maxSize, inSize[i]);
  }
  return ansEncodeImpl(temp_dev, tempBytes, tempUsed, probBits, useChecksum, numInBatch, &hp,
                       nullptr, nullptr, maxSize, histogram_dev, outSize_dev, (hipStream_t)stream);
}

int dgpu_ans_encode_batch_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, const uint32_t* inSplitSizes,
    const uint32_t* histogram_dev, void* out_dev, uint32_t outStride, uint32_t* outSize_dev,
    void* stream) {
  // alignment rules of GpuANSEncode.cu:132-140
  DGPU_REQUIRE(((uintptr_t)in_dev % DGPU_ANS_REQUIRED_ALIGNMENT) == 0, "ANS input must be 4-byte aligned");
  DGPU_REQUIRE(((uintptr_t)out_dev % 16) == 0 && (numInBatch <= 1 || outStride % 16 == 0),
               "compressed output must be 16-byte aligned");
  for (uint32_t i = 0; i + 1 < numInBatch; ++i) {
    DGPU_REQUIRE(inSplitSizes[i] % DGPU_ANS_REQUIRED_ALIGNMENT == 0,
                 "interior split sizes must be multiples of 4 bytes");
  }
  HostParams hp;
  uint32_t maxSize = 0;
  splitSizesToPointers(in_dev, inSplitSizes, numInBatch, 1, &hp.inPtrs, &hp.sizes, &maxSize);
  hp.outPtrs.resize(numInBatch);
  for (uint32_t i = 0; i < numInBatch; ++i) {
    hp.outPtrs[i] = (uint64_t)(uintptr_t)out_dev + (uint64_t)i * outStride;
  }
  return ansEncodeImpl(temp_dev, tempBytes, tempUsed, probBits, useChecksum, numInBatch, &hp,
                       nullptr, nullptr, maxSize, histogram_dev, outSize_dev, (hipStream_t)stream);
}

// ---- decode ----------------------------------------------------------------
int dgpu_ans_decode_batch_stride(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inPerBatchStride, void* out_dev,
    uint32_t outPerBatchStride, uint32_t outPerBatchCapacity, uint8_t* outSuccess_dev,
    uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  DGPU_REQUIRE(((uintptr_t)in_dev % 16) == 0 && (numInBatch <= 1 || inPerBatchStride % 16 == 0),
               "compressed input must be 16-byte aligned");
  BatchView in = viewStride(in_dev, inPerBatchStride, 0);
  BatchView out = viewStride(out_dev, outPerBatchStride, outPerBatchCapacity);
  return decodeImpl(temp_dev, tempBytes, tempUsed, 0, probBits, useChecksum, numInBatch, nullptr,
                    &in, &out, outPerBatchCapacity, outSuccess_dev, outSize_dev,
                    (hipStream_t)stream, errBatch);
}

static int decodePointerCommon(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t ft, int probBits,
    int useChecksum, uint32_t numInBatch, const void* const* in, void* const* out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream,
    int32_t* errBatch, const uint32_t* inBytes = nullptr) {
  HostParams hp;
  hp.inPtrs.resize(numInBatch);
  hp.outPtrs.resize(numInBatch);
  hp.sizes.resize(numInBatch);
  if (inBytes) hp.inBytes.assign(inBytes, inBytes + numInBatch);
  uint32_t maxCap = 0;
  for (uint32_t i = 0; i < numInBatch; ++i) {
    DGPU_REQUIRE(((uintptr_t)in[i] % 16) == 0, "compressed input must be 16-byte aligned");
    hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
    hp.outPtrs[i] = (uint64_t)(uintptr_t)out[i];
    hp.sizes[i] = outCapacity[i];
    maxCap = std::max(maxCap, outCapacity[i]);
  }
  return decodeImpl(temp_dev, tempBytes, tempUsed, ft, probBits, useChecksum, numInBatch, &hp,
                    nullptr, nullptr, maxCap, outSuccess_dev, outSize_dev, (hipStream_t)stream,
                    errBatch);
}

int dgpu_ans_decode_batch_pointer(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, void* const* out, const uint32_t* outCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  return decodePointerCommon(temp_dev, tempBytes, tempUsed, 0, probBits, useChecksum, numInBatch,
                             in, out, outCapacity, outSuccess_dev, outSize_dev, stream, errBatch);
}

static int decodeSplitCommon(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t ft, int probBits,
    int useChecksum, uint32_t numInBatch, const void* const* in, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream,
    int32_t* errBatch, const uint32_t* inBytes = nullptr) {
  HostParams hp;
  if (inBytes) hp.inBytes.assign(inBytes, inBytes + numInBatch);
  uint32_t maxCap = 0;
  splitSizesToPointers(out_dev, outSplitSizes, numInBatch, ft ? floatWordBytes(ft) : 1u,
                       &hp.outPtrs, &hp.sizes, &maxCap);
  hp.inPtrs.resize(numInBatch);
  for (uint32_t i = 0; i < numInBatch; ++i) {
    DGPU_REQUIRE(((uintptr_t)in[i] % 16) == 0, "compressed input must be 16-byte aligned");
    hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
  }
  return decodeImpl(temp_dev, tempBytes, tempUsed, ft, probBits, useChecksum, numInBatch, &hp,
                    nullptr, nullptr, maxCap, outSuccess_dev, outSize_dev, (hipStream_t)stream,
                    errBatch);
}

int dgpu_ans_decode_batch_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, void* out_dev, const uint32_t* outSplitSizes,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  DGPU_REQUIRE(((uintptr_t)out_dev % DGPU_ANS_REQUIRED_ALIGNMENT) == 0, "output must be 4-byte aligned");
  for (uint32_t i = 0; i + 1 < numInBatch; ++i) {
    DGPU_REQUIRE(outSplitSizes[i] % DGPU_ANS_REQUIRED_ALIGNMENT == 0,
                 "interior split sizes must be multiples of 4 bytes");
  }
  return decodeSplitCommon(temp_dev, tempBytes, tempUsed, 0, probBits, useChecksum, numInBatch, in,
                           out_dev, outSplitSizes, outSuccess_dev, outSize_dev, stream, errBatch);
}

// ---- decode with known input sizes ("bounded") ---------------------------------
// Same as the four pointer / split-size decode entry points, plus `inBytes` (HOST array): the bytes available at
// in[i].  The reference API carries no compressed sizes, so a truncated archive is followed past its buffer
// there; the tensor API knows every tensor's size and uses these.
int dgpu_ans_decode_batch_pointer_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes, void* const* out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  return decodePointerCommon(temp_dev, tempBytes, tempUsed, 0, probBits, useChecksum, numInBatch, in, out, outCapacity,
                             outSuccess_dev, outSize_dev, stream, errBatch, inBytes);
}
int dgpu_ans_decode_batch_split_size_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  DGPU_REQUIRE(((uintptr_t)out_dev % DGPU_ANS_REQUIRED_ALIGNMENT) == 0, "output must be 4-byte aligned");
  for (uint32_t i = 0; i + 1 < numInBatch; ++i) {
    DGPU_REQUIRE(outSplitSizes[i] % DGPU_ANS_REQUIRED_ALIGNMENT == 0, "interior split sizes must be multiples of 4 bytes");
  }
  return decodeSplitCommon(temp_dev, tempBytes, tempUsed, 0, probBits, useChecksum, numInBatch, in, out_dev,
                           outSplitSizes, outSuccess_dev, outSize_dev, stream, errBatch, inBytes);
}
int dgpu_float_decompress_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes, void* const* out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  return decodePointerCommon(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum, numInBatch, in, out,
                             outCapacity, outSuccess_dev, outSize_dev, stream, errBatch, inBytes);
}
int dgpu_float_decompress_split_size_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  return decodeSplitCommon(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum, numInBatch, in, out_dev,
                           outSplitSizes, outSuccess_dev, outSize_dev, stream, errBatch, inBytes);
}

// ---- float stride batches with capacities on both sides (the compressed collectives) --------------------
int dgpu_float_compress_stride_capped(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inWords, uint32_t inStrideBytes, void* out_dev,
    uint32_t outStrideBytes, uint32_t outCapacityBytes, uint32_t* outSize_dev, void* stream) {
  DGPU_REQUIRE(validProbBits(probBits), "probBits must be 9, 10 or 11");
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  DGPU_REQUIRE(numInBatch <= 65535u, "numInBatch must be <= 65535");
  DGPU_REQUIRE(encodableSize(inWords), "tensor larger than 1717538816 words (GpuANSEncode.cu:22)");
  const uint32_t wb = floatWordBytes(floatType);
  DGPU_REQUIRE(((uintptr_t)in_dev % wb) == 0 && (numInBatch <= 1 || inStrideBytes % wb == 0), "float input must be float-word aligned");
  DGPU_REQUIRE(((uintptr_t)out_dev % 16) == 0 && (numInBatch <= 1 || outStrideBytes % 16 == 0) && outCapacityBytes % 16 == 0,
               "compressed output rows, their stride and their capacity must be 16-byte aligned");
  DGPU_REQUIRE(numInBatch <= 1 || outCapacityBytes <= outStrideBytes, "outCapacityBytes must not exceed outStrideBytes");
  DGPU_REQUIRE(inWords > kBlockSize, "capped compression needs rows of more than one 4096-word block");
  // everything except the block data is stored unconditionally: it must fit
  const uint32_t nb = divUp(inWords, kBlockSize);
  const uint64_t fixed = (uint64_t)ansOffsetInArchive(floatType, inWords) + ansOverhead(nb);
  DGPU_REQUIRE(fixed <= outCapacityBytes, "outCapacityBytes is smaller than the archive's header, tables and non-compressed planes");
  if (tempUsed) *tempUsed = 0;
  if (numInBatch == 0) return DGPU_OK;
  hipStream_t st = (hipStream_t)stream;
  StreamLease streamLease(st);
  TempArena arena(temp_dev, tempBytes, streamLease);
  const BatchView in = viewStride(in_dev, inStrideBytes, inWords);
  const BatchView out = viewStride(out_dev, outStrideBytes, 0);
  int rc = encodeCommon(arena, streamLease, st, probBits, useChecksum != 0, numInBatch, in, out, floatType, inWords, nullptr,
                        outSize_dev, outCapacityBytes);
  if (tempUsed) *tempUsed = arena.requested();
  return rc;
}

int dgpu_float_decompress_stride_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inStrideBytes, uint32_t inBytes, void* out_dev,
    uint32_t outStrideBytes, uint32_t outCapacityWords, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream,
    int32_t* errBatch) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  DGPU_REQUIRE(((uintptr_t)in_dev % 16) == 0 && (numInBatch <= 1 || inStrideBytes % 16 == 0),
               "compressed input must be 16-byte aligned");
  const uint32_t wb = floatWordBytes(floatType);
  DGPU_REQUIRE(((uintptr_t)out_dev % wb) == 0 && (numInBatch <= 1 || outStrideBytes % wb == 0), "float output must be float-word aligned");
  DGPU_REQUIRE(inBytes != 0, "inBytes must be the bytes available per compressed row");
  BatchView in = viewStride(in_dev, inStrideBytes, 0);
  BatchView out = viewStride(out_dev, outStrideBytes, outCapacityWords);
  return decodeImpl(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum, numInBatch, nullptr, &in, &out,
                    outCapacityWords, outSuccess_dev, outSize_dev, (hipStream_t)stream, errBatch, inBytes);
}

// ---- info ------------------------------------------------------------------
int dgpu_ans_get_compressed_info_device(
    const void* const* in_dev, uint32_t numInBatch, uint32_t* outSizes_dev,
    uint32_t* outChecksum_dev, void* stream) {
  if (numInBatch == 0 || (!outSizes_dev && !outChecksum_dev)) return DGPU_OK;
  BatchView in = viewPointers((const uint64_t*)in_dev, nullptr, 0);
  hipLaunchKernelGGL(k_ans_info, dim3(divUp(numInBatch, 128)), dim3(128), 0, (hipStream_t)stream,
                     in, numInBatch, outSizes_dev, outChecksum_dev);
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}

int dgpu_ans_get_compressed_info(
    void* temp_dev, size_t tempBytes, const void* const* in, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outChecksum_dev, void* stream) {
  if (numInBatch == 0 || (!outSizes_dev && !outChecksum_dev)) return DGPU_OK;
  (void)temp_dev;
  (void)tempBytes;
  ParamLease lease;
  HostParams hp;
  hp.inPtrs.resize(numInBatch);
  for (uint32_t i = 0; i < numInBatch; ++i) hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
  const uint64_t *inP = nullptr, *outP = nullptr;
  const uint32_t* sz = nullptr;
  int rc = uploadParams(lease, (hipStream_t)stream, hp, &inP, &outP, &sz);
  if (rc) return rc;
  return dgpu_ans_get_compressed_info_device((const void* const*)inP, numInBatch, outSizes_dev,
                                             outChecksum_dev, stream);
}

int dgpu_float_get_compressed_info_device(
    const void* const* in_dev, uint32_t numInBatch, uint32_t* outSizes_dev,
    uint32_t* outTypes_dev, uint32_t* outChecksum_dev, void* stream) {
  if (numInBatch == 0 || (!outSizes_dev && !outTypes_dev && !outChecksum_dev)) return DGPU_OK;
  BatchView in = viewPointers((const uint64_t*)in_dev, nullptr, 0);
  hipLaunchKernelGGL(k_float_info, dim3(divUp(numInBatch, 128)), dim3(128), 0, (hipStream_t)stream,
                     in, numInBatch, outSizes_dev, outTypes_dev, outChecksum_dev);
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}

int dgpu_float_get_compressed_info(
    void* temp_dev, size_t tempBytes, const void* const* in, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outTypes_dev, uint32_t* outChecksum_dev, void* stream) {
  if (numInBatch == 0 || (!outSizes_dev && !outTypes_dev && !outChecksum_dev)) return DGPU_OK;
  (void)temp_dev;
  (void)tempBytes;
  ParamLease lease;
  HostParams hp;
  hp.inPtrs.resize(numInBatch);
  for (uint32_t i = 0; i < numInBatch; ++i) hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
  const uint64_t *inP = nullptr, *outP = nullptr;
  const uint32_t* sz = nullptr;
  int rc = uploadParams(lease, (hipStream_t)stream, hp, &inP, &outP, &sz);
  if (rc) return rc;
  return dgpu_float_get_compressed_info_device((const void* const*)inP, numInBatch, outSizes_dev,
                                               outTypes_dev, outChecksum_dev, stream);
}

// ---- float codec -------------------------------------------------------------
int dgpu_float_compress(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits,
    int useChecksum, uint32_t numInBatch, const void* const* in, const uint32_t* inSize,
    void* const* out, uint32_t* outSize_dev, void* stream) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  HostParams hp;
  hp.inPtrs.resize(numInBatch);
  hp.outPtrs.resize(numInBatch);
  hp.sizes.resize(numInBatch);
  uint32_t maxSize = 0;
  const uint32_t wb = floatWordBytes(floatType);
  for (uint32_t i = 0; i < numInBatch; ++i) {
    DGPU_REQUIRE(((uintptr_t)in[i] % wb) == 0, "float input must be float-word aligned");
    DGPU_REQUIRE(((uintptr_t)out[i] % 16) == 0, "compressed output must be 16-byte aligned");
    hp.inPtrs[i] = (uint64_t)(uintptr_t)in[i];
    hp.outPtrs[i] = (uint64_t)(uintptr_t)out[i];
    hp.sizes[i] = inSize[i];
    maxSize = std::max(maxSize, inSize[i]);
  }
  return floatCompressImpl(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum,
                           numInBatch, hp, maxSize, outSize_dev, (hipStream_t)stream);
}

int dgpu_float_compress_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits,
    int useChecksum, uint32_t numInBatch, const void* in_dev, const uint32_t* inSplitSizes,
    void* out_dev, uint32_t outStride, uint32_t* outSize_dev, void* stream) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  DGPU_REQUIRE(((uintptr_t)out_dev % 16) == 0 && (numInBatch <= 1 || outStride % 16 == 0),
               "compressed output must be 16-byte aligned");
  HostParams hp;
  uint32_t maxSize = 0;
  splitSizesToPointers(in_dev, inSplitSizes, numInBatch, floatWordBytes(floatType), &hp.inPtrs,
                       &hp.sizes, &maxSize);
  hp.outPtrs.resize(numInBatch);
  for (uint32_t i = 0; i < numInBatch; ++i) {
    hp.outPtrs[i] = (uint64_t)(uintptr_t)out_dev + (uint64_t)i * outStride;
  }
  return floatCompressImpl(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum,
                           numInBatch, hp, maxSize, outSize_dev, (hipStream_t)stream);
}

int dgpu_float_decompress(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits,
    int useChecksum, uint32_t numInBatch, const void* const* in, void* const* out,
    const uint32_t* outCapacity, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream,
    int32_t* errBatch) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  return decodePointerCommon(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum,
                             numInBatch, in, out, outCapacity, outSuccess_dev, outSize_dev, stream,
                             errBatch);
}

int dgpu_float_decompress_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits,
    int useChecksum, uint32_t numInBatch, const void* const* in, void* out_dev,
    const uint32_t* outSplitSizes, uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream,
    int32_t* errBatch) {
  DGPU_REQUIRE(validFloatType(floatType), "floatType must be float16, bfloat16 or float32");
  return decodeSplitCommon(temp_dev, tempBytes, tempUsed, floatType, probBits, useChecksum,
                           numInBatch, in, out_dev, outSplitSizes, outSuccess_dev, outSize_dev,
                           stream, errBatch);
}

// ---- building blocks for parity tests ----------------------------------------
int dgpu_ans_histogram_batch_stride(
    uint32_t numInBatch, const void* in_dev, uint32_t inPerBatchSize, uint32_t inPerBatchStride,
    uint32_t* histogram_dev, void* stream) {
  DGPU_REQUIRE(numInBatch <= 65535u, "numInBatch must be <= 65535");
  if (numInBatch == 0) return DGPU_OK;
  DGPU_HIP(hipMemsetAsync(histogram_dev, 0, (size_t)numInBatch * kNumSymbols * 4, (hipStream_t)stream));
  BatchView in = viewStride(in_dev, inPerBatchStride, inPerBatchSize);
  dim3 grid(gridX(inPerBatchSize, 32 * 1024, 64), numInBatch);
  HistFuse noFuse;
  noFuse.workMap = nullptr;
  noFuse.partBytes = 0;
  noFuse.arrive = nullptr;
  noFuse.acc = nullptr;
  noFuse.norm = NormalizeArgs{};
  hipLaunchKernelGGL((k_histogram<kHistSlotsLarge, true>), grid, dim3(256), 0, (hipStream_t)stream, in, histogram_dev, 0u, noFuse);
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}

int dgpu_ans_calc_weights(
    uint32_t numInBatch, int probBits, const uint32_t* sizes_dev, uint32_t uniformSize,
    const uint32_t* histogram_dev, uint32_t* table_dev, void* stream) {
  DGPU_REQUIRE(validProbBits(probBits), "probBits must be 9, 10 or 11");
  if (numInBatch == 0) return DGPU_OK;
  NormalizeArgs n;
  n.sizes = viewPointers(nullptr, sizes_dev, uniformSize);
  n.hist = histogram_dev;
  n.histAcc = nullptr;
  n.histParts = 1;
  n.probBits = probBits;
  n.encTable = nullptr;
  n.refTable = (uint4*)table_dev;
  n.out = viewStride(nullptr, 0, 0);
  n.writeHeader = 0;
  n.floatType = 0;
  n.useChecksum = 0;
  n.checksum = nullptr;
  n.outSize = nullptr;
  n.floatUseChecksum = 0;
  n.tileDesc = nullptr;
  n.maxTiles = 0;
  n.claims = nullptr;
  n.numInBatch = numInBatch;
  n.tileBase = nullptr;
  n.tileSymbols = 0;
  n.groupWords = nullptr;
  n.groupWordsPerElement = 0;
  hipLaunchKernelGGL(k_normalize, dim3(numInBatch), dim3(256), 0, (hipStream_t)stream, n);
  DGPU_HIP(hipGetLastError());
  return DGPU_OK;
}

}  // extern "C"
