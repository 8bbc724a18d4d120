// Fibre scheduler + CUDA runtime stand-ins for dgemu.h (TEST INFRASTRUCTURE ONLY).
//
// One OS thread.  A launch runs the blocks of the grid one after the other; the
// threads of a block are ucontext fibres scheduled round-robin.  A fibre leaves
// the CPU only inside __syncthreads() or a warp collective, where it waits for
// a condition the scheduler can test without switching to it.
#include <signal.h>
#include <ucontext.h>

#include <cstdarg>
#include <string>
#include <vector>

#include "include/dgemu.h"

namespace dgemu {

ThreadCtx* cur = nullptr;

namespace {

constexpr size_t kStackBytes = 256 * 1024;

struct Warp {
  uint32_t live = 0;      // lanes that have not returned from the kernel
  uint32_t arrived = 0;   // lanes that deposited a value in the current exchange
  uint32_t consumed = 0;  // lanes that picked the result up
  uint32_t gen = 0;       // exchanges completed
  uint64_t vals[32];
};

enum Wait { kRunnable, kBarrier, kWarpArrive, kWarpGen, kDone };

struct Fibre {
  ucontext_t ctx;
  ThreadCtx tc;
  Wait wait = kRunnable;
  uint32_t warp = 0, lane = 0;
  uint32_t need = 0;  // kWarpArrive: mask that must have arrived
  uint32_t gen = 0;   // kWarpGen: generation to leave behind
};

struct Block {
  std::vector<Fibre> fibres;
  std::vector<Warp> warps;
  uint32_t live = 0;
  uint32_t atBarrier = 0;
  ucontext_t sched;
  Fibre* running = nullptr;
  void (*thunk)(void*) = nullptr;
  void* closure = nullptr;
};

Block* g_block = nullptr;
std::vector<char*> g_stacks;  // reused across launches

void yieldToScheduler() {
  Fibre* f = g_block->running;
  swapcontext(&f->ctx, &g_block->sched);
  cur = &f->tc;
}

void fibreMain() {
  Block* b = g_block;
  Fibre* f = b->running;
  b->thunk(b->closure);
  // thread exit
  f->wait = kDone;
  b->live--;
  b->warps[f->warp].live &= ~(1u << f->lane);
  swapcontext(&f->ctx, &b->sched);
  abort();  // never resumed
}

bool runnable(Block& b, Fibre& f) {
  switch (f.wait) {
    case kRunnable:
      return true;
    case kBarrier:
      return false;  // released in bulk by the scheduler
    case kWarpArrive: {
      Warp& w = b.warps[f.warp];
      const uint32_t need = f.need & w.live;
      return (w.arrived & need) == need;
    }
    case kWarpGen:
      return b.warps[f.warp].gen != f.gen;
    case kDone:
      return false;
  }
  return false;
}

}  // namespace

void syncthreads() {
  Block* b = g_block;
  Fibre* f = b->running;
  f->wait = kBarrier;
  b->atBarrier++;
  yieldToScheduler();
}

void warpExchange(uint32_t mask, uint64_t v, uint64_t out[32], uint32_t* participants) {
  Block* b = g_block;
  Fibre* f = b->running;
  Warp& w = b->warps[f->warp];
  // a previous exchange of this warp must have been fully consumed before values are overwritten
  while (w.consumed != 0 && ((w.consumed >> f->lane) & 1u)) {
    f->wait = kWarpGen;
    f->gen = w.gen;
    yieldToScheduler();
  }
  w.vals[f->lane] = v;
  w.arrived |= 1u << f->lane;
  for (;;) {
    const uint32_t need = mask & w.live;
    if ((w.arrived & need) == need) break;
    f->wait = kWarpArrive;
    f->need = mask;
    yieldToScheduler();
  }
  const uint32_t part = mask & w.live & w.arrived;
  for (int i = 0; i < 32; ++i) out[i] = w.vals[i];
  *participants = part;
  w.consumed |= 1u << f->lane;
  if ((w.consumed & part) == part) {
    // last one out resets the exchange
    w.arrived &= ~part;
    w.consumed = 0;
    w.gen++;
  } else {
    f->wait = kWarpGen;
    f->gen = w.gen;
    while (w.gen == f->gen) yieldToScheduler();
  }
  f->wait = kRunnable;
}

// On the GPU an integer division by zero does not trap (the quotient is all ones); the reference
// relies on that for the division magic of symbols with pdf 0, which is computed and never looked up
// (GpuANSStatistics.cuh:349-358).  x86 raises SIGFPE: skip the div/idiv and deliver the GPU's result.
void onSigFpe(int, siginfo_t* si, void* ucv) {
  ucontext_t* uc = (ucontext_t*)ucv;
  const uint8_t* ip = (const uint8_t*)uc->uc_mcontext.gregs[REG_RIP];
  const uint8_t* p = ip;
  while (*p == 0x66 || *p == 0x67 || *p == 0xf2 || *p == 0xf3) ++p;
  if ((*p & 0xf0) == 0x40) ++p;  // REX
  if ((si->si_code != FPE_INTDIV && si->si_code != FPE_INTOVF) || (*p != 0xf6 && *p != 0xf7)) {
    signal(SIGFPE, SIG_DFL);
    return;  // not an integer divide: re-raise with the default action
  }
  ++p;
  const uint8_t modrm = *p++;
  const uint8_t mod = modrm >> 6, rm = modrm & 7;
  if (mod != 3) {
    uint8_t base = rm;
    if (rm == 4) base = *p++ & 7;  // SIB
    if (mod == 1) p += 1;
    else if (mod == 2 || (mod == 0 && base == 5)) p += 4;
  }
  uc->uc_mcontext.gregs[REG_RAX] = -1;
  uc->uc_mcontext.gregs[REG_RDX] = 0;
  uc->uc_mcontext.gregs[REG_RIP] = (greg_t)(uintptr_t)p;
}

void runGrid(const LaunchCfg& cfg, void (*thunk)(void*), void* closure) {
  static bool fpeInstalled = false;
  if (!fpeInstalled) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = onSigFpe;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGFPE, &sa, nullptr);
    fpeInstalled = true;
  }
  const uint32_t nthreads = cfg.block.x * cfg.block.y * cfg.block.z;
  if (nthreads == 0 || cfg.grid.x * cfg.grid.y * cfg.grid.z == 0) return;
  if (g_block) {
    fprintf(stderr, "dgemu: nested launch\n");
    abort();
  }
  while (g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(kStackBytes));
  const uint32_t nwarps = (nthreads + 31) / 32;

  Block b;
  b.thunk = thunk;
  b.closure = closure;
  g_block = &b;
  for (uint32_t bz = 0; bz < cfg.grid.z; ++bz)
    for (uint32_t by = 0; by < cfg.grid.y; ++by)
      for (uint32_t bx = 0; bx < cfg.grid.x; ++bx) {
        b.fibres.assign(nthreads, Fibre());
        b.warps.assign(nwarps, Warp());
        b.live = nthreads;
        b.atBarrier = 0;
        for (uint32_t t = 0; t < nthreads; ++t) {
          Fibre& f = b.fibres[t];
          f.tc.tid = uint3{t % cfg.block.x, (t / cfg.block.x) % cfg.block.y, t / (cfg.block.x * cfg.block.y)};
          f.tc.bid = uint3{bx, by, bz};
          f.tc.bdim = cfg.block;
          f.tc.gdim = cfg.grid;
          f.tc.linear = t;
          f.warp = t / 32;
          f.lane = t % 32;
          b.warps[f.warp].live |= 1u << f.lane;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = g_stacks[t];
          f.ctx.uc_stack.ss_size = kStackBytes;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fibreMain, 0);
        }
        // round-robin until every fibre has returned
        while (b.live > 0) {
          bool progressed = false;
          for (uint32_t t = 0; t < nthreads; ++t) {
            Fibre& f = b.fibres[t];
            if (!runnable(b, f)) continue;
            progressed = true;
            f.wait = (f.wait == kWarpArrive || f.wait == kWarpGen) ? f.wait : kRunnable;
            b.running = &f;
            cur = &f.tc;
            swapcontext(&b.sched, &f.ctx);
          }
          // barrier release: every live thread is waiting at it
          if (b.live > 0 && b.atBarrier == b.live) {
            for (Fibre& f : b.fibres)
              if (f.wait == kBarrier) f.wait = kRunnable;
            b.atBarrier = 0;
            progressed = true;
          }
          if (!progressed) {
            fprintf(stderr, "dgemu: deadlock in block (%u,%u,%u): %u live, %u at barrier\n", bx, by, bz, b.live, b.atBarrier);
            abort();
          }
        }
      }
  g_block = nullptr;
  cur = nullptr;
}

// PTX of dietgpu/utils/PtxUtils.cuh:14-100 and dietgpu/float/GpuFloatUtils.cuh:155-157
uint64_t ptx_eval(const char* tmpl, const uint64_t* in, int n) {
  const std::string t(tmpl);
  auto starts = [&](const char* p) { return t.rfind(p, 0) == 0; };
  const uint32_t lane = cur ? (cur->linear & 31u) : 0u;
  if (starts("bfe.u32")) {  // bit field extract: val, pos, len
    const uint32_t val = (uint32_t)in[0], pos = (uint32_t)in[1] & 0xff, len = (uint32_t)in[2] & 0xff;
    if (len == 0 || pos >= 32) return 0;
    const uint64_t m = len >= 32 ? 0xffffffffull : ((1ull << len) - 1);
    return ((uint64_t)val >> pos) & m;
  }
  if (starts("bfe.u64")) {
    const uint64_t val = in[0];
    const uint32_t pos = (uint32_t)in[1] & 0xff, len = (uint32_t)in[2] & 0xff;
    if (len == 0 || pos >= 64) return 0;
    const uint64_t m = len >= 64 ? ~0ull : ((1ull << len) - 1);
    return (val >> pos) & m;
  }
  if (starts("bfi.b32")) {  // insert `a` into `b` at pos, len: operands a, b, pos, len
    const uint32_t a = (uint32_t)in[0], b = (uint32_t)in[1], pos = (uint32_t)in[2] & 0xff, len = (uint32_t)in[3] & 0xff;
    if (len == 0 || pos >= 32) return b;
    const uint32_t m = (uint32_t)((len >= 32 ? 0xffffffffull : ((1ull << len) - 1)) << pos);
    return (b & ~m) | ((a << pos) & m);
  }
  if (starts("shf.l.clamp.b32")) {  // funnel shift left of {b, a} by min(c, 32); result = high word: operands a(lo), b(hi), c
    const uint64_t lo = (uint32_t)in[0], hi = (uint32_t)in[1];
    const uint32_t c = std::min<uint32_t>((uint32_t)in[2], 32u);
    const uint64_t v = (hi << 32) | lo;
    return c == 0 ? hi : (uint32_t)((v << c) >> 32);
  }
  if (starts("shf.r.clamp.b32")) {  // funnel shift right of {b, a} by min(c, 32); result = low word
    const uint64_t lo = (uint32_t)in[0], hi = (uint32_t)in[1];
    const uint32_t c = std::min<uint32_t>((uint32_t)in[2], 32u);
    const uint64_t v = (hi << 32) | lo;
    return (uint32_t)(v >> c);
  }
  if (t.find("%%laneid") != std::string::npos || t.find("%laneid") != std::string::npos) return lane;
  if (t.find("lanemask_lt") != std::string::npos) return (1u << lane) - 1u;
  if (t.find("lanemask_le") != std::string::npos) return lane == 31 ? 0xffffffffu : ((1u << (lane + 1)) - 1u);
  if (t.find("lanemask_gt") != std::string::npos) return lane == 31 ? 0u : ~((1u << (lane + 1)) - 1u);
  if (t.find("lanemask_ge") != std::string::npos) return ~((1u << lane) - 1u);
  fprintf(stderr, "dgemu: PTX not emulated: %s (%d operands)\n", tmpl, n);
  abort();
}

}  // namespace dgemu

// ---- runtime on host memory ----------------------------------------------------------
cudaError_t cudaMalloc(void** p, size_t n) {
  *p = calloc(n ? n : 1, 1);  // zero-filled: bytes the reference never writes read as zero
  return *p ? cudaSuccess : cudaErrorInvalidValue;
}
cudaError_t cudaFree(void* p) {
  free(p);
  return cudaSuccess;
}
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) {
  memmove(d, s, n);
  return cudaSuccess;
}
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t) {
  memmove(d, s, n);
  return cudaSuccess;
}
cudaError_t cudaMemset(void* d, int v, size_t n) {
  memset(d, v, n);
  return cudaSuccess;
}
cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t) {
  memset(d, v, n);
  return cudaSuccess;
}
cudaError_t cudaGetDevice(int* d) {
  *d = 0;
  return cudaSuccess;
}
cudaError_t cudaSetDevice(int) { return cudaSuccess; }
cudaError_t cudaGetDeviceCount(int* n) {
  *n = 1;
  return cudaSuccess;
}
cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "dgemu CPU SIMT emulation");
  p->major = 8;
  p->minor = 0;
  p->multiProcessorCount = 2;
  p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 48 * 1024;
  p->unifiedAddressing = 1;
  return cudaSuccess;
}
cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
cudaError_t cudaGetLastError() { return cudaSuccess; }
const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "error"; }
const char* cudaGetErrorName(cudaError_t e) { return e == cudaSuccess ? "cudaSuccess" : "cudaError"; }
cudaError_t cudaStreamCreateWithFlags(cudaStream_t* s, unsigned) {
  *s = (cudaStream_t)malloc(8);
  return cudaSuccess;
}
cudaError_t cudaStreamDestroy(cudaStream_t s) {
  free(s);
  return cudaSuccess;
}
cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned) { return cudaSuccess; }
cudaError_t cudaEventCreateWithFlags(cudaEvent_t* e, unsigned) {
  *e = (cudaEvent_t)malloc(8);
  return cudaSuccess;
}
cudaError_t cudaEventCreate(cudaEvent_t* e) { return cudaEventCreateWithFlags(e, 0); }
cudaError_t cudaEventDestroy(cudaEvent_t e) {
  free(e);
  return cudaSuccess;
}
cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t) { return cudaSuccess; }
cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t, cudaEvent_t) {
  *ms = 0.f;
  return cudaSuccess;
}
cudaError_t cudaPointerGetAttributes(cudaPointerAttributes* a, const void* p) {
  a->type = a->memoryType = cudaMemoryTypeDevice;
  a->device = 0;
  a->devicePointer = a->hostPointer = (void*)p;
  return cudaSuccess;
}
cudaError_t cudaProfilerStart() { return cudaSuccess; }
cudaError_t cudaProfilerStop() { return cudaSuccess; }
