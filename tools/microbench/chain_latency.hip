// Microbenchmark: issue-to-issue latency of DEPENDENT instruction chains on gfx950 (what a row of the rANS loops is
// made of): VALU -> VALU, SALU -> SALU, and the hop VALU -> SALU -> VALU (v_cmp, s_bcnt1, v_add with the SGPR).
// Build: hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
template <int kMode>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters, uint32_t seed) {
  uint32_t v = threadIdx.x + seed, w = seed | 1u;
  uint32_t s = seed;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (uint32_t i = 0; i < iters; ++i) {
    if (kMode == 0) { REP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));) }
    if (kMode == 1) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v) : "v"(w));) }
    if (kMode == 2) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v) : "v"(w));) }
    if (kMode == 3) { REP8(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");) }
    if (kMode == 4) {  // VALU -> SGPR (v_cmp) -> SALU (s_bcnt1) -> VALU
      REP8(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\ts_bcnt1_i32_b64 %2, vcc\n\tv_add_u32 %0, %0, %2" : "+v"(v), "+v"(w), "+s"(s) : : "vcc", "scc");)
    }
    if (kMode == 5) {  // VALU -> SGPR (v_cmp) -> VALU reading the SGPR pair (v_mbcnt_lo / v_mbcnt_hi)
      REP8(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_mbcnt_lo_u32_b32 %0, vcc_lo, %0\n\tv_mbcnt_hi_u32_b32 %0, vcc_hi, %0" : "+v"(v), "+v"(w) : : "vcc");)
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (t1 - t0) + ((v + s) & 1u);
}

template <int kMode>
void run(const char* name, int opsPerRep, uint64_t* d) {
  for (int wgPerCu : {1, 8}) {
    const int grid = 256 * wgPerCu;
    const uint32_t iters = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<kMode>, dim3(grid), dim3(256), 0, 0, d, iters, 12345u);  // warm
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<kMode>, dim3(grid), dim3(256), 0, 0, d, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(grid * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double ticksPerWave = s / h.size();
    printf("%-52s %d waves/SIMD: %6.2f ticks per instruction  (kernel %.1f us, %.0f ticks per wave => counter at %.2f GHz if a wave spans the kernel)\n",
           name, wgPerCu, ticksPerWave / iters / 8 / opsPerRep, ms * 1e3, ticksPerWave, ticksPerWave / (ms * 1e6));
  }
}

int main() {
  uint64_t* d; (void)hipMalloc(&d, 256 * 8 * 4 * 8);
  run<0>("dependent v_add_u32", 1, d);
  run<1>("dependent v_mad_u32_u24", 1, d);
  run<2>("dependent v_mul_hi_u32", 1, d);
  run<3>("dependent s_add_u32", 1, d);
  run<4>("v_cmp -> s_bcnt1 -> v_add (3 instructions per link)", 3, d);
  run<5>("v_cmp -> v_mbcnt_lo -> v_mbcnt_hi (3 per link)", 3, d);
  return 0;
}
