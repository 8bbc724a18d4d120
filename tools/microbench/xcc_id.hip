// Which XCD does workgroup i run on?  Prints HW_REG_XCC_ID for the first workgroups of a 1-D grid.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t* out) {
  if (threadIdx.x == 0) {
    uint32_t a = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    uint32_t full = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
    uint32_t hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    out[blockIdx.x * 3] = a; out[blockIdx.x * 3 + 1] = full; out[blockIdx.x * 3 + 2] = hwid;
  }
}
int main() {
  uint32_t* d; hipMalloc(&d, 4096 * 12);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d);
  uint32_t h[4096 * 3]; hipMemcpy(h, d, 2048 * 12, hipMemcpyDeviceToHost);
  for (int i = 0; i < 24; ++i) printf("wg %d xcc %u full 0x%x hwid 0x%x\n", i, h[i * 3], h[i * 3 + 1], h[i * 3 + 2]);
  int match = 0; for (int i = 0; i < 2048; ++i) match += (h[i * 3] == (uint32_t)(i % 8));
  printf("xcc == blockIdx %% 8 for %d of 2048 workgroups\n", match);
  return 0;
}
