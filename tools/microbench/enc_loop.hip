// Ablation microbenchmark of the rANS encode row loop on gfx950.
// One wave64 = two 32-lane blocks, table + symbol ring + word stage in LDS, as in
// dietgpu_amd/csrc/kernels_encode.h.  VARIANT bits switch parts of the row off
// (results are kept live with asm volatile so nothing is dead-code-eliminated).
//   bit0: no stage write      bit1: no ballot bookkeeping (fixed prefix)
//   bit2: no division chain   bit3: no table lookup (constant entry)
// Prints cycles per row per SIMD for several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) uint16_t LdsU16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 LdsU4;

constexpr int kRows = 128 * 16;  // rows per wave

template <int VARIANT>
__global__ __launch_bounds__(256) void k(uint32_t* out, const uint4* gtable, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint4* table = (uint4*)smem;                  // 4 KiB
  uint8_t* ring = smem + 4096;                  // 8 x 512
  uint8_t* stage = smem + 4096 + 4096;          // 8 x 5184
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane & 31;
  const bool upper = lane >= 32;
  const uint32_t hw = wave * 2 + (upper ? 1 : 0);
  table[tid] = gtable[tid];
  for (int i = tid; i < 4096 / 4; i += 256) ((uint32_t*)ring)[i] = (i * 2654435761u + seed) >> 3;
  __syncthreads();
  const uint32_t tableLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)table;
  const uint32_t stageBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(stage + hw * 5184);
  const uint32_t dummyAddr = stageBase + 5100;
  const uint8_t* myRing = ring + hw * 512;
  const uint32_t laneMaskLt = (1u << hl) - 1u;
  uint32_t state = 1u << 15, outOff = 0;
  const uint4 constE = table[7];

  for (int c = 0; c < kRows / 16; ++c) {
    uint32_t toff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      toff[r] = tableLds + (((uint32_t)myRing[r * 32 + hl] & 63u) << 4);  // 64 distinct symbols
      asm volatile("" : "+v"(toff[r]));
    }
    uint4 e[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { u32x4 v = *(const LdsU4*)(uintptr_t)toff[r]; e[r] = make_uint4(v.x, v.y, v.z, v.w); }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      uint4 ce = (VARIANT & 8) ? constE : e[r % 4];
      if (r + 4 < 16 && !(VARIANT & 8)) { u32x4 v = *(const LdsU4*)(uintptr_t)toff[r + 4]; e[r % 4] = make_uint4(v.x, v.y, v.z, v.w); }
      const bool write = state >= ce.x;
      uint32_t idx, cnt;
      if (VARIANT & 2) { idx = outOff; cnt = 1; }
      else {
        const uint64_t vote = __ballot(write);
        const uint32_t vh = upper ? (uint32_t)(vote >> 32) : (uint32_t)vote;
        idx = outOff + __popc(vh & laneMaskLt);
        cnt = __popc(vh);
      }
      if (!(VARIANT & 1)) {
        const uint32_t addr = write ? stageBase + 2u * (idx & 2047u) : dummyAddr;
        *(LdsU16*)(uintptr_t)addr = (uint16_t)state;
      }
      state = write ? (state >> 16) : state;
      if (VARIANT & 4) { state = state + ce.z + (ce.w & 1023); }
      else {
        const uint32_t t = __umulhi(state, ce.y);
        const uint32_t div = (t + state) >> (ce.w >> 24);
        state = __umul24(div, ce.w) + state + ce.z;
      }
      state &= 0x7fffffffu;
      outOff += cnt;
    }
    outOff &= 1023u;
  }
  out[blockIdx.x * 256 + tid] = state + outOff;
}

template <int V>
void run(const char* name, uint32_t* d, uint4* tab) {
  for (int wgPerCU : {1, 2, 3, 4, 6}) {
    int grid = 256 * wgPerCU;
    size_t lds = 4096 + 4096 + 8 * 5184 + 64;
    if (wgPerCU > 3) lds = 4096 + 4096 + 8 * 5184 / 2;  // fits more WGs (stage wraps at 2048 words anyway)
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, tab, 1u);
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, tab, 2u + r);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    // rows per SIMD = wgPerCU waves per SIMD * kRows
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wgPerCU * kRows);
    printf("%-34s %d waves/SIMD: %7.3f ms  %6.1f cycles/row/SIMD @2.4GHz (per-wave row time %6.1f)\n", name, wgPerCU, ms, cyc, cyc * wgPerCU);
  }
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  std::vector<uint4> h(256);
  for (int i = 0; i < 256; ++i) {
    uint32_t pdf = 1 + (i * 37) % 60;  // small pdfs, P=10
    uint32_t shift = 32 - __builtin_clz(pdf - 1 ? pdf - 1 : 1); if (pdf == 1) shift = 0;
    uint64_t magic = ((1ull << 32) * ((1ull << shift) - pdf)) / pdf + 1;
    h[i] = make_uint4(pdf << 21, (uint32_t)magic, (i * 13) & 1023, ((1024 - pdf) & 0xffffff) | (shift << 24));
  }
  uint4* tab; hipMalloc(&tab, 4096); hipMemcpy(tab, h.data(), 4096, hipMemcpyHostToDevice);
  run<0>("full row", d, tab);
  run<1>("no stage write", d, tab);
  run<2>("no ballot bookkeeping", d, tab);
  run<4>("no division chain", d, tab);
  run<8>("no table lookup", d, tab);
  run<15>("skeleton (all off)", d, tab);
  return 0;
}
