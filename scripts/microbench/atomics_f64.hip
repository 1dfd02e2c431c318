// Microbenchmark: throughput of no-return fp64 global atomic adds on gfx950 for the access shapes the Hessian
// assembly can choose between.  Each wave issues `iters` wave-instructions; instruction k of wave w adds 64 doubles
// laid out as (64 / run) runs of `run` contiguous doubles; runs start at pseudo-random offsets (aligned to `align`
// doubles) inside a table of `n` doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_atomic(double* tab, unsigned n, int run, int align, int iters, int mode) {
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  const unsigned r = lane / run, o = lane % run;
  for (int k = 0; k < iters; ++k) {
    unsigned h = hash(wave * 7919u + k * 104729u + r * 31u + 1u);
    unsigned base = (h % ((n - 64) / align)) * align;
    double* p = tab + base + o;
    if (mode == 0) unsafeAtomicAdd(p, 1.0);
    else if (mode == 1) *p = 1.0;                     // plain store, same shape
    else if (mode == 2) { double v = *p; *p = v + 1.0; }  // racy read-modify-write, same shape
  }
}
int main() {
  const unsigned n = 20u << 20;  // 160 MB of doubles, like the config-3 Hessian
  double* tab;
  CK(hipMalloc(&tab, sizeof(double) * n));
  CK(hipMemset(tab, 0, sizeof(double) * n));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 8, iters = 256;
  const double adds = (double)blocks * 256 * iters;
  struct Cfg { int run, align; } cfgs[] = {{64, 64}, {64, 8}, {64, 1}, {32, 8}, {32, 1}, {16, 8}, {16, 1}, {8, 8}, {8, 1}, {4, 1}, {2, 1}, {1, 1}};
  for (int mode = 0; mode < 3; ++mode)
    for (auto c : cfgs) {
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, tab, n, c.run, c.align, 8, mode);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, tab, n, c.run, c.align, iters, mode);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("mode %d (%s) run %2d align %2d: %8.3f ms  %7.2f G adds/s  %7.3f TB/s\n", mode, mode == 0 ? "atomic" : mode == 1 ? "store " : "rmw   ",
             c.run, c.align, ms, adds / ms * 1e-6, adds * 8 / ms * 1e-9);
    }
  return 0;
}
