// Histogram variants on bf16 normal data (exponent byte), 256 x 524288 words.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hist_variants tools/microbench/hist_variants.hip && /tmp/hist_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int kSlots, int kThreads, int kInFlight, bool kAtomic>
__global__ __launch_bounds__(kThreads) void k_hist(const uint4* __restrict__ in, uint32_t vecPerElem, uint32_t* __restrict__ out) {
  __shared__ uint32_t bins[256 * kSlots];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 256 * kSlots / 4; i += kThreads) ((uint4*)bins)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  uint32_t* mine = bins + (tid & (kSlots - 1));
  const uint4* pv = in + (size_t)blockIdx.y * vecPerElem;
  const uint32_t stride = gridDim.x * kThreads;
  uint32_t acc = 0;
  auto add = [&](uint32_t c) {
    if (kAtomic) atomicAdd(&mine[c * kSlots], 1u);
    else acc += c;
  };
  auto addVec = [&](const uint4& x) {
    const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      add((xw[j] >> 7) & 0xffu);
      add((xw[j] >> 23) & 0xffu);
    }
  };
  uint32_t v = blockIdx.x * kThreads + tid;
  for (; v + (kInFlight - 1) * stride < vecPerElem; v += kInFlight * stride) {
    uint4 x[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) x[k] = pv[v + k * stride];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) addVec(x[k]);
  }
  for (; v < vecPerElem; v += stride) addVec(pv[v]);
  __syncthreads();
  if (!kAtomic) { if (acc == 0x12345u) out[0] = acc; return; }
  for (uint32_t bin = tid; bin < 256; bin += kThreads) {
    uint32_t sum = 0;
    for (int k = 0; k < kSlots; ++k) sum += bins[bin * kSlots + ((k + tid) & (kSlots - 1))];
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + bin] = sum;
  }
}

// contiguous chunk per workgroup instead of grid-strided
template <int kSlots, int kThreads, int kInFlight>
__global__ __launch_bounds__(kThreads) void k_hist_chunk(const uint4* __restrict__ in, uint32_t vecPerElem, uint32_t* __restrict__ out) {
  __shared__ uint32_t bins[256 * kSlots];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 256 * kSlots / 4; i += kThreads) ((uint4*)bins)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  uint32_t* mine = bins + (tid & (kSlots - 1));
  const uint32_t per = vecPerElem / gridDim.x;
  const uint4* pv = in + (size_t)blockIdx.y * vecPerElem + (size_t)blockIdx.x * per;
  auto addVec = [&](const uint4& x) {
    const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      atomicAdd(&mine[((xw[j] >> 7) & 0xffu) * kSlots], 1u);
      atomicAdd(&mine[((xw[j] >> 23) & 0xffu) * kSlots], 1u);
    }
  };
  for (uint32_t v = tid; v < per; v += kInFlight * kThreads) {
    uint4 x[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) x[k] = pv[v + k * kThreads];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) addVec(x[k]);
  }
  __syncthreads();
  for (uint32_t bin = tid; bin < 256; bin += kThreads) {
    uint32_t sum = 0;
    for (int k = 0; k < kSlots; ++k) sum += bins[bin * kSlots + ((k + tid) & (kSlots - 1))];
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + bin] = sum;
  }
}

int main() {
  const uint32_t B = 256, N = 524288;
  const size_t bytes = (size_t)B * N * 2;
  std::vector<uint16_t> h((size_t)B * N);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < (size_t)N * 8; ++i) { float f = nd(rng); uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  for (size_t i = (size_t)N * 8; i < h.size(); ++i) h[i] = h[i - (size_t)N * 8 + (i % 977)];
  // kWindows copies of the data, visited in rotation, so that the 256 MiB infinity cache
  // cannot serve the reads (as in the codec's step, whose working set is ~700 MB)
  const int kWindows = 6;
  uint4* dAll; uint32_t* out;
  CK(hipMalloc(&dAll, bytes * kWindows)); CK(hipMalloc(&out, (size_t)B * 64 * 256 * 4));
  for (int w = 0; w < kWindows; ++w) CK(hipMemcpy((char*)dAll + bytes * w, h.data(), bytes, hipMemcpyHostToDevice));
  int rot = 0;
  uint4* d = dAll;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) { d = (uint4*)((char*)dAll + bytes * (rot++ % kWindows)); launch(); }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3 / reps, bytes / 1e9 * reps / (ms * 1e-3));
  };
  const uint32_t V = N / 8;
#define RUN(S, T, F, A, GX) timeit("slots " #S " thr " #T " inflight " #F " atomic " #A " gx " #GX, [&] { hipLaunchKernelGGL((k_hist<S, T, F, A>), dim3(GX, B), dim3(T), 0, 0, d, V, out); })
  RUN(16, 256, 4, true, 32);
  RUN(32, 256, 4, true, 32);
  RUN(32, 256, 4, true, 16);
  RUN(32, 256, 4, true, 8);
  RUN(32, 256, 8, true, 16);
  RUN(32, 512, 4, true, 16);
  RUN(32, 1024, 4, true, 8);
  RUN(16, 256, 4, false, 32);
  RUN(16, 256, 8, false, 16);
#define RUNC(S, T, F, GX) timeit("chunk slots " #S " thr " #T " inflight " #F " gx " #GX, [&] { hipLaunchKernelGGL((k_hist_chunk<S, T, F>), dim3(GX, B), dim3(T), 0, 0, d, V, out); })
  RUNC(32, 256, 4, 16);
  RUNC(32, 256, 8, 8);
  RUNC(16, 256, 4, 16);
  return 0;
}
