// Microbenchmark: random 16-byte gathers per second from tables of several sizes, with U independent
// loads in flight per lane.  hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int U>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ tab, uint32_t mask, uint32_t iters, uint32_t* out) {
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x = x * 1664525u + 1013904223u;
      v[u] = tab[(x >> 4) & mask];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t max_bytes = 1ull << 30;
  uint4* tab;
  uint32_t* out;
  hipMalloc(&tab, max_bytes);
  hipMalloc(&out, 4);
  hipMemset(tab, 1, max_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const uint32_t grid = 256 * 8, iters_total = 64;
  for (size_t bytes : {1ull << 20, 4ull << 20, 16ull << 20, 128ull << 20, 1ull << 30}) {
    const uint32_t mask = (uint32_t)(bytes / 16 - 1);
    for (int U : {1, 2, 4, 8}) {
      const uint32_t iters = iters_total / U;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (U == 1) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out);
        if (U == 2) hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out);
        if (U == 4) hipLaunchKernelGGL(gather<4>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out);
        if (U == 8) hipLaunchKernelGGL(gather<8>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)grid * 256 * iters * U;
      printf("table %5zu MB  U=%d  %.3f ms  %.1f G gathers/s\n", bytes >> 20, U, ms, n / ms / 1e6);
    }
  }
  return 0;
}
