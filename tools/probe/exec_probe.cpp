// exec_probe.cpp -- does a VALU wave-instruction cost less issue time when only part of the wavefront is enabled?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/exec_probe.cpp -o /tmp/exec_probe && /tmp/exec_probe
// Every workgroup is one wavefront; 24 wavefronts per CU (6 per SIMD) each run a chain-free stream of fp64 FMAs (8
// independent accumulators) with `active` lanes enabled.  If the SIMD skipped 16-lane passes whose EXEC bits are all
// zero, the 1- and 16-lane runs would take a quarter of the 64-lane time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void k(double *out, int iters, int active, double seed)
{
  const int lane = threadIdx.x;
  double a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  if (lane < active) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      }
    }
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
  const int blocks = 256 * 24, iters = 20000;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 64, 48, 32, 17, 16, 8, 1}) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, active, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d: %.3f ms  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", active, ms,
           ms * 1e-3 * 2.4e9 / (6.0 * iters * 64));
  }
  return 0;
}
