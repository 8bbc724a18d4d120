// Ablation microbenchmark of the rANS decode row loop on gfx950 (one wave64 = two
// 32-lane blocks; 64-bit LUT + word ring in LDS; per-row global byte load and
// short store), mirroring dietgpu_amd/csrc/kernels_decode.h.
//   bit0: no ring word read     bit1: no ballot bookkeeping
//   bit2: no LUT lookup         bit3: no global load/store (join kept live)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) uint16_t LdsU16;
constexpr int kRows = 128 * 16;
constexpr int P = 10;

template <int VARIANT>
__global__ __launch_bounds__(256) void k(uint16_t* out, const uint8_t* nc, const uint2* glut, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint2* lut = (uint2*)(smem + 16384);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane & 31;
  const bool upper = lane >= 32;
  const uint32_t hw = wave * 2 + (upper ? 1 : 0);
  for (int i = tid; i < 1024; i += 256) lut[i] = glut[i];
  for (int i = tid; i < 16384 / 4; i += 256) ((uint32_t*)smem)[i] = (i * 2654435761u + seed);
  __syncthreads();
  const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t ringBase = ldsBase + hw * 2048;
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  uint32_t state = (1u << 15) + tid * 977u, posw = 1000000;
  const size_t blk = ((size_t)blockIdx.x * 8 + hw) * 4096;
  const uint8_t* myNc = nc + blk % (32u << 20) + hl;
  uint16_t* myOut = out + blk % (16u << 20) + hl;
  uint32_t acc = 0;

  for (int g = 0; g < kRows / 8; ++g) {
    uint32_t pre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = (VARIANT & 8) ? (g + j) : myNc[((g & 15) * 8 + j) * 32];
#pragma unroll
    for (int j = 7; j >= 0; --j) {
      uint2 e;
      if (VARIANT & 4) e = make_uint2((state & 63) + 1 | (state << 24), state & 1023);
      else e = lut[state & 1023];
      state = __umul24(e.x, state >> P) + e.y;
      const bool read = state < (1u << 15);
      uint32_t idx;
      if (VARIANT & 2) { idx = posw; posw -= 1; }
      else {
        const uint64_t vote = __ballot(read);
        const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
        posw -= __popc(vh);
        idx = posw + __popc(vh & laneMaskLt);
      }
      uint32_t w = idx;
      if (!(VARIANT & 1)) w = *(const LdsU16*)(uintptr_t)(((idx << 1) & 2047u) | ringBase);
      state = read ? ((state << 16) | (w & 0xffffu)) : state;
      state = (state & 0x7fffffffu) | 0x10000u;
      const uint32_t lo = (pre[j] << 16) | e.x;
      const uint32_t v = __builtin_amdgcn_alignbit(pre[j], lo, 1);
      if (VARIANT & 8) acc ^= v; else myOut[((g & 15) * 8 + j) * 32] = (uint16_t)(v >> 16);
    }
  }
  if (acc == 0x12345) out[tid] = 1;
  if (state == 77) out[tid] = 2;
}

template <int V>
void run(const char* name, uint16_t* d, uint8_t* nc, uint2* lut) {
  for (int wgPerCU : {1, 2, 4, 6, 8}) {
    int grid = 256 * wgPerCU;
    size_t lds = 16384 + 8192;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, nc, lut, 1u);
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, nc, lut, 2u + r);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wgPerCU * kRows);
    printf("%-34s %d waves/SIMD: %7.3f ms  %6.1f cycles/row/SIMD @2.4GHz (per-wave row time %6.1f)\n", name, wgPerCU, ms, cyc, cyc * wgPerCU);
  }
}

int main() {
  uint16_t* d; hipMalloc(&d, (size_t)96 << 20);
  uint8_t* nc; hipMalloc(&nc, (size_t)96 << 20); hipMemset(nc, 1, (size_t)96 << 20);
  std::vector<uint2> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = make_uint2((1 + (i % 60)) | ((i & 255u) << 24), i % 50);
  uint2* lut; hipMalloc(&lut, 8192); hipMemcpy(lut, h.data(), 8192, hipMemcpyHostToDevice);
  run<0>("full row", d, nc, lut);
  run<1>("no ring word read", d, nc, lut);
  run<2>("no ballot bookkeeping", d, nc, lut);
  run<4>("no LUT lookup", d, nc, lut);
  run<8>("no global load/store", d, nc, lut);
  run<15>("skeleton (all off)", d, nc, lut);
  return 0;
}
