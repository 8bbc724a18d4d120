// Calibrates rocprofv3's WRITE_SIZE (and the time) for the store shapes the decoder uses:
// every kernel writes exactly 256 MiB, so WRITE_SIZE (KiB) / 262144 is the counter's scale for that
// shape.  The decoder's float join stores 2 bytes per lane per row; a half-wave row is one
// contiguous 64-byte piece (bf16/fp16), 128 bytes (fp32) or 32 bytes (raw bytes).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/write_calib tools/microbench/write_calib.hip
//   rocprofv3 --pmc WRITE_SIZE -d /tmp/wc -o wc -- /tmp/write_calib      (tools/gpu_write_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t kBytes = 256u << 20;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte stores, consecutive lanes consecutive vectors
template <bool kNt>
__global__ __launch_bounds__(256) void wc_store16(uint8_t* out) {
  u32x4* p = (u32x4*)out;
  const size_t n = kBytes / 16, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    u32x4 v = {(uint32_t)i, 1, 2, 3};
    if (kNt) __builtin_nontemporal_store(v, &p[i]); else p[i] = v;
  }
}

// The decoder's shape: a half-wave (32 lanes) owns a block of 4096 elements of W bytes and writes it
// row by row, W bytes per lane per row (row r = elements [32 r, 32 r + 32)).  16 blocks per workgroup.
template <typename T, bool kNt>
__global__ __launch_bounds__(512) void wc_rows(uint8_t* out) {
  const uint32_t hw = threadIdx.x >> 5, hl = threadIdx.x & 31u;
  const size_t numBlocks = kBytes / (4096 * sizeof(T));
  for (size_t blk = (size_t)blockIdx.x * 16 + hw; blk < numBlocks; blk += (size_t)gridDim.x * 16) {
    T* p = (T*)out + blk * 4096 + hl;
#pragma unroll 8
    for (int r = 127; r >= 0; --r) {
      const T v = (T)(r * 3 + hl);
      if (kNt) __builtin_nontemporal_store(v, &p[r * 32]); else p[r * 32] = v;
    }
  }
}

int main() {
  uint8_t* d;
  CK(hipMalloc(&d, kBytes));
  hipEvent_t evA, evB;
  CK(hipEventCreate(&evA));
  CK(hipEventCreate(&evB));
  auto timeit = [&](const char* name, auto launch) {
    launch();
    (void)hipEventRecord(evA);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(evB);
    (void)hipEventSynchronize(evB);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, evA, evB);
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms / 5 * 1e3, kBytes / (ms / 5 * 1e-3) / 1e9);
    return 0;
  };
  timeit("store16 plain", [&] { hipLaunchKernelGGL(wc_store16<false>, dim3(2048), dim3(256), 0, 0, d); });
  timeit("store16 nt", [&] { hipLaunchKernelGGL(wc_store16<true>, dim3(2048), dim3(256), 0, 0, d); });
  timeit("rows u16 plain (64 B pieces)", [&] { hipLaunchKernelGGL((wc_rows<uint16_t, false>), dim3(2048), dim3(512), 0, 0, d); });
  timeit("rows u16 nt", [&] { hipLaunchKernelGGL((wc_rows<uint16_t, true>), dim3(2048), dim3(512), 0, 0, d); });
  timeit("rows u32 plain (128 B)", [&] { hipLaunchKernelGGL((wc_rows<uint32_t, false>), dim3(2048), dim3(512), 0, 0, d); });
  timeit("rows u32 nt", [&] { hipLaunchKernelGGL((wc_rows<uint32_t, true>), dim3(2048), dim3(512), 0, 0, d); });
  timeit("rows u8 plain (32 B)", [&] { hipLaunchKernelGGL((wc_rows<uint8_t, false>), dim3(2048), dim3(512), 0, 0, d); });
  timeit("rows u8 nt", [&] { hipLaunchKernelGGL((wc_rows<uint8_t, true>), dim3(2048), dim3(512), 0, 0, d); });
  CK(hipDeviceSynchronize());
  return 0;
}
