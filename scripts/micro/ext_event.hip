// Micro-benchmark: what timing a kernel with hipEventRecord brackets costs against start/stop events bound to the
// kernel's own dispatch (hipExtLaunchKernelGGL).  Build: hipcc -O2 --offload-arch=gfx950 -o ext_event ext_event.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin(uint32_t* out, uint32_t iters) {
  uint32_t x = threadIdx.x;
  for (uint32_t i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
  if (x == 7u) out[0] = x;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
  uint32_t* d;
  CK(hipMalloc(&d, 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int K = 6, reps = 200;
  std::vector<hipEvent_t> a(K + 1), s(K), e(K);
  for (auto& x : a) CK(hipEventCreate(&x));
  for (auto& x : s) CK(hipEventCreate(&x));
  for (auto& x : e) CK(hipEventCreate(&x));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  for (uint32_t iters : {2000u, 20000u}) {
    for (int mode = 0; mode < 3; ++mode) {
      float total = 0, inner = 0;
      for (int r = 0; r < reps + 5; ++r) {
        CK(hipEventRecord(t0, st));
        if (mode == 1) CK(hipEventRecord(a[0], st));
        for (int k = 0; k < K; ++k) {
          if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, s[k], e[k], 0, d, iters);
          else hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, d, iters);
          if (mode == 1) CK(hipEventRecord(a[k + 1], st));
        }
        CK(hipEventRecord(t1, st));
        CK(hipStreamSynchronize(st));
        if (r < 5) continue;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, t0, t1));
        total += ms;
        for (int k = 0; k < K; ++k) {
          float m2 = 0;
          if (mode == 1) CK(hipEventElapsedTime(&m2, a[k], a[k + 1]));
          if (mode == 2) CK(hipEventElapsedTime(&m2, s[k], e[k]));
          inner += m2;
        }
      }
      printf("iters %u mode %s: %d kernels in %.1f us per round, per kernel by its events %.2f us\n", iters,
             mode == 0 ? "no events      " : (mode == 1 ? "event records  " : "ext launch evts"), K, 1e3 * total / reps,
             mode ? 1e3 * inner / reps / K : 0.0);
    }
  }
  return 0;
}
