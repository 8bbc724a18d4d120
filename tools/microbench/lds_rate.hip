// Microbenchmark: LDS latency of a dependent read chain and throughput of the decoder's LUT access pattern
// (64 random 8-byte reads per wave-instruction over an 8 KiB table) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// dependent chain: x = lut[x & 1023] (b64: next index in .x), kIters times; one result per wave
template <int kBytes>
__global__ __launch_bounds__(256) void k_chain(uint64_t* out, uint32_t iters, uint32_t seed) {
  __shared__ uint2 lut[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lut[i] = make_uint2((uint32_t)((i * 2654435761u + seed) >> 7), i);
  __syncthreads();
  uint32_t x = threadIdx.x * 37u + blockIdx.x;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (uint32_t i = 0; i < iters; ++i) {
    if (kBytes == 8) { const uint2 e = lut[x & 1023u]; x = e.x + e.y; }
    else { x = ((const uint32_t*)lut)[x & 2047u] + 1u; }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (t1 - t0) | ((uint64_t)(x & 1) << 63);
}

// throughput: 8 independent random reads per iteration
template <int kMode>  // 0: b64 random, 1: b64 linear (conflict-free), 2: b32 random
__global__ __launch_bounds__(256) void k_tput(uint32_t* out, uint32_t iters, uint32_t seed) {
  __shared__ uint2 lut[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lut[i] = make_uint2(i * 3u + seed, i);
  __syncthreads();
  uint32_t a[8], acc = 0;
  for (int j = 0; j < 8; ++j) a[j] = (threadIdx.x * 2654435761u + j * 40503u + seed) >> 5;
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (kMode == 0) { const uint2 e = lut[a[j] & 1023u]; acc += e.x ^ e.y; }
      if (kMode == 1) { const uint2 e = lut[(threadIdx.x + j * 64u + i) & 1023u]; acc += e.x ^ e.y; }
      if (kMode == 2) { acc += ((const uint32_t*)lut)[a[j] & 2047u]; }
      a[j] = a[j] * 1664525u + 1013904223u;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  uint64_t* d64; uint32_t* d32;
  hipMalloc(&d64, 256 * 16 * 4 * 8); hipMalloc(&d32, 256 * 16 * 256 * 4);
  const uint32_t iters = 4096;
  for (int wgPerCu : {1, 2, 4, 6, 8}) {
    for (int b : {8, 4}) {
      const int grid = 256 * wgPerCu;
      if (b == 8) hipLaunchKernelGGL(k_chain<8>, dim3(grid), dim3(256), 0, 0, d64, iters, 7u);
      else hipLaunchKernelGGL(k_chain<4>, dim3(grid), dim3(256), 0, 0, d64, iters, 7u);
      hipDeviceSynchronize();
      std::vector<uint64_t> h(grid * 4);
      hipMemcpy(h.data(), d64, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += (double)(v & ~(1ull << 63));
      printf("dependent ds_read_b%-2d chain, %d waves/SIMD: %7.1f cycles per read (s_memtime/readcyclecounter ticks)\n", b * 8, wgPerCu, s / h.size() / iters);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"ds_read_b64 random (8 KiB table)", "ds_read_b64 linear", "ds_read_b32 random"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int wgPerCu : {2, 8}) {
      const int grid = 256 * wgPerCu;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_tput<0>, dim3(grid), dim3(256), 0, 0, d32, iters, 3u);
        if (mode == 1) hipLaunchKernelGGL(k_tput<1>, dim3(grid), dim3(256), 0, 0, d32, iters, 3u);
        if (mode == 2) hipLaunchKernelGGL(k_tput<2>, dim3(grid), dim3(256), 0, 0, d32, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double waveInstrPerCu = (double)wgPerCu * 4 * iters * 8;
      printf("%-34s %d waves/SIMD: %6.3f ms = %5.2f cycles per wave-instruction per CU @2.4 GHz\n", names[mode], wgPerCu, ms,
             ms * 1e-3 * 2.4e9 / waveInstrPerCu);
    }
  }
  return 0;
}
