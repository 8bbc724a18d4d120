// What can a COPY reach on this chip?  (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; hbm_rate.hip's copy
// reads 4.3-4.9.)  Sweep: plain / non-temporal loads x plain / non-temporal stores x grid x vectors in flight per lane,
// on 256 MiB -> 256 MiB and 1 GiB -> 1 GiB; and the decoder's mix (read 0.4 N, write 0.6 N).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_rate tools/microbench/copy_rate.hip && /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
      else out[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
// contiguous 64 KiB per workgroup visit (the codec's shape), persistent workgroups
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_chunks(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
  const size_t chunks = n / 4096;  // 64 KiB chunks
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const u32x4* p = in + c * 4096;
    u32x4* q = out + c * 4096;
    for (uint32_t i = threadIdx.x; i < 4096; i += 1024) {
      u32x4 a = NTL ? __builtin_nontemporal_load(&p[i]) : p[i], b = NTL ? __builtin_nontemporal_load(&p[i + 256]) : p[i + 256];
      u32x4 cc = NTL ? __builtin_nontemporal_load(&p[i + 512]) : p[i + 512], d = NTL ? __builtin_nontemporal_load(&p[i + 768]) : p[i + 768];
      if (NTS) { __builtin_nontemporal_store(a, &q[i]); __builtin_nontemporal_store(b, &q[i + 256]); __builtin_nontemporal_store(cc, &q[i + 512]); __builtin_nontemporal_store(d, &q[i + 768]); }
      else { q[i] = a; q[i + 256] = b; q[i + 512] = cc; q[i + 768] = d; }
    }
  }
}
// the decoder's mix: read 2 vectors, write 3 (0.4 N read, 0.6 N written of N = 5 vectors)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_mix_dec(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t nIn) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; 2 * i + 1 < nIn; i += stride) {
    u32x4 a = NTL ? __builtin_nontemporal_load(&in[2 * i]) : in[2 * i], b = NTL ? __builtin_nontemporal_load(&in[2 * i + 1]) : in[2 * i + 1];
    u32x4 c = a ^ b;
    if (NTS) { __builtin_nontemporal_store(a, &out[3 * i]); __builtin_nontemporal_store(b, &out[3 * i + 1]); __builtin_nontemporal_store(c, &out[3 * i + 2]); }
    else { out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c; }
  }
}

int main() {
  for (size_t mib : {256ul, 1024ul}) {
    const size_t bytes = mib << 20, n = bytes / 16;
    u32x4 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes + (bytes >> 1)) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double gb, auto launch) {
      for (int w = 0; w < 2; ++w) launch();
      hipEventRecord(e0);
      const int reps = 10;
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%4zu MiB %-44s %8.1f us  %7.1f GB/s (read + written)\n", mib, name, ms * 1e3 / reps, gb * reps / (ms * 1e-3));
    };
    const double GB = bytes / 1e9;
    char nm[96];
#define COPY(NTL, NTS, U, G) snprintf(nm, 96, "copy ntl=%d nts=%d inflight=%d grid=%d", NTL, NTS, U, G); \
    timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_copy<NTL, NTS, U>), dim3(G), dim3(256), 0, 0, a, b, n); });
    for (int g : {2048, 4096, 8192, 16384}) {
      COPY(false, false, 4, g) COPY(true, false, 4, g) COPY(false, true, 4, g) COPY(true, true, 4, g)
    }
    COPY(true, true, 8, 2048) COPY(true, true, 8, 4096) COPY(false, false, 8, 4096) COPY(true, true, 2, 8192) COPY(true, true, 1, 16384)
    for (int g : {768, 1536, 2048}) {
      snprintf(nm, 96, "copy 64K chunks persistent ntl=1 nts=1 grid=%d", g);
      timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_copy_chunks<true, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
      snprintf(nm, 96, "copy 64K chunks persistent ntl=1 nts=0 grid=%d", g);
      timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_copy_chunks<true, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
      snprintf(nm, 96, "copy 64K chunks persistent ntl=0 nts=0 grid=%d", g);
      timeit(nm, 2 * GB, [&] { hipLaunchKernelGGL((k_copy_chunks<false, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    for (int g : {4096, 8192}) {
      snprintf(nm, 96, "decoder mix (0.4 r : 0.6 w) ntl=0 nts=1 grid=%d", g);
      timeit(nm, GB + 1.5 * GB, [&] { hipLaunchKernelGGL((k_mix_dec<false, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
      snprintf(nm, 96, "decoder mix (0.4 r : 0.6 w) ntl=1 nts=1 grid=%d", g);
      timeit(nm, GB + 1.5 * GB, [&] { hipLaunchKernelGGL((k_mix_dec<true, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
      snprintf(nm, 96, "decoder mix (0.4 r : 0.6 w) ntl=0 nts=0 grid=%d", g);
      timeit(nm, GB + 1.5 * GB, [&] { hipLaunchKernelGGL((k_mix_dec<false, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
