// Microbenchmark: issue rate of the integer VALU ops the rANS loops are made of,
// on gfx950, at full occupancy.  Prints wave-instructions per cycle per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 256
#define UNROLL 8

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
  uint32_t a4 = a0 ^ 0x1234, a5 = a1 ^ 0x777, a6 = a2 ^ 0x999, a7 = a3 ^ 0x555;
  uint32_t m = seed | 1;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (OP == 0) { a0 += m; a1 += m; a2 += m; a3 += m; a4 += m; a5 += m; a6 += m; a7 += m; }
      if (OP == 1) { a0 = __umulhi(a0, m); a1 = __umulhi(a1, m); a2 = __umulhi(a2, m); a3 = __umulhi(a3, m);
                     a4 = __umulhi(a4, m); a5 = __umulhi(a5, m); a6 = __umulhi(a6, m); a7 = __umulhi(a7, m); }
      if (OP == 2) { a0 = __umul24(a0, m) + a1; a1 = __umul24(a1, m) + a2; a2 = __umul24(a2, m) + a3; a3 = __umul24(a3, m) + a4;
                     a4 = __umul24(a4, m) + a5; a5 = __umul24(a5, m) + a6; a6 = __umul24(a6, m) + a7; a7 = __umul24(a7, m) + a0; }
      if (OP == 3) { a0 = __popc(a0) + a1; a1 = __popc(a1) + a2; a2 = __popc(a2) + a3; a3 = __popc(a3) + a4;
                     a4 = __popc(a4) + a5; a5 = __popc(a5) + a6; a6 = __popc(a6) + a7; a7 = __popc(a7) + a0; }
      if (OP == 4) { uint64_t x = ((uint64_t)a1 << 32) | a0; x >>= (a2 & 32); a0 = (uint32_t)x + a3;
                     uint64_t y = ((uint64_t)a4 << 32) | a3; y >>= (a5 & 32); a3 = (uint32_t)y + a6;
                     uint64_t z = ((uint64_t)a7 << 32) | a6; z >>= (a1 & 32); a6 = (uint32_t)z + a0;
                     uint64_t w = ((uint64_t)a2 << 32) | a1; w >>= (a4 & 32); a1 = (uint32_t)w + a7; }
      if (OP == 5) { a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m; }
      if (OP == 6) { a0 = (a0 >> 3) & m; a1 = (a1 >> 3) & m; a2 = (a2 >> 3) & m; a3 = (a3 >> 3) & m;
                     a4 = (a4 >> 3) & m; a5 = (a5 >> 3) & m; a6 = (a6 >> 3) & m; a7 = (a7 >> 3) & m; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
double run(const char* name, int opsPerUnroll, uint32_t* d, int blocksPerCU) {
  int grid = 256 * blocksPerCU;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(s);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 12345u + r);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
  double waveInstr = (double)grid * 4 * ITERS * UNROLL * opsPerUnroll;  // per launch
  double perSimdPerSec = waveInstr / (ms * 1e-3) / 1024.0;
  printf("%-26s blocks/CU %d  %8.3f ms  %6.3f G wave-instr/s/SIMD  (= %.2f cycles/instr @2.4GHz, %.2f @2.0GHz)\n",
         name, blocksPerCU, ms, perSimdPerSec / 1e9, 2.4e9 / perSimdPerSec, 2.0e9 / perSimdPerSec);
  return perSimdPerSec;
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int bpc : {1, 2, 4, 8}) {
    run<0>("v_add_u32", 8, d, bpc);
    run<1>("v_mul_hi_u32", 8, d, bpc);
    run<2>("v_mad_u32_u24", 8, d, bpc);
    run<3>("v_bcnt+add (2 ops?)", 8, d, bpc);
    run<4>("v_lshrrev_b64+and+add", 4 * 3, d, bpc);
    run<5>("v_mul_lo_u32", 8, d, bpc);
    run<6>("v_lshr + v_and (2 ops)", 16, d, bpc);
    printf("\n");
  }
  return 0;
}
