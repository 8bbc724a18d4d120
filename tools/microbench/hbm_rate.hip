// Achievable HBM rates on this chip for the access shapes the codec uses:
// read-only, write-only, copy, and "read 2 bytes, write 1.35 bytes" (the bf16
// encoder's ratio).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_rate tools/microbench/hbm_rate.hip && /tmp/hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n; i += stride) { uint4 a = in[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// contiguous chunk per workgroup (the codec's shape: each WG streams its own 32 KiB+ region)
__global__ __launch_bounds__(256) void k_read_chunk(const uint4* __restrict__ in, size_t n, uint32_t* out, uint32_t vecPerWg) {
  uint32_t acc = 0;
  const uint4* p = in + (size_t)blockIdx.x * vecPerWg;
  for (uint32_t i = threadIdx.x; i < vecPerWg; i += 1024) {
    uint4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ out, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
    out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) out[i] = in[i];
}
// read 2 vectors, write ~1.35: writes 27 of every 40 output vectors
__global__ __launch_bounds__(256) void k_mix(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n; i += 2 * stride) {
    uint4 a = in[i], b = in[i + stride];
    out[i >> 1] = make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
    if ((i % 40) < 14) out[(n >> 1) + (i >> 2)] = b;
  }
}

int main() {
  const size_t bytes = 256ull << 20;
  const size_t n = bytes / 16;
  uint4 *a, *b; uint32_t* o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double gb, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms * 1e3 / reps, gb * reps / (ms * 1e-3));
  };
  const double GB = bytes / 1e9;
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "read  grid %d", g); timeit(nm, GB, [&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, o); });
  }
  for (uint32_t v : {2048u, 4096u, 16384u}) {
    char nm[64];
    snprintf(nm, 64, "read chunk %u KiB/WG", v * 16 / 1024);
    timeit(nm, GB, [&] { hipLaunchKernelGGL(k_read_chunk, dim3(n / v), dim3(256), 0, 0, a, n, o, v); });
  }
  for (int g : {2048, 8192}) {
    char nm[64];
    snprintf(nm, 64, "write grid %d", g); timeit(nm, GB, [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
    snprintf(nm, 64, "copy  grid %d (r+w bytes)", g); timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); });
    snprintf(nm, 64, "mix   grid %d (r + .675w)", g); timeit(nm, 1.675 * GB, [&] { hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, a, b, n); });
  }
  // small buffer that fits the 256 MiB infinity cache / L2: 32 MiB read
  timeit("read 32 MiB (cache) g4096", 32.0 * (1 << 20) / 1e9, [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, (32u << 20) / 16, o); });
  return 0;
}
