// peer_probe.cpp -- can two kernels of ONE process on two streams wait for each other through a generation word in device memory?
// (the in-kernel exchange of the reduced rows between thread ranks, gph_engine.hip: k_reduce_stage)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/peer_probe.cpp -o tools/probe/peer_probe && tools/probe/peer_probe
// For every scope (agent / system) and launch order: each kernel publishes flags[me] = gen and waits (bounded: 1 s) for flags[other];
// out = cycles of the 100-MHz counter waited, or -1 on a time-out.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE> __global__ void k(unsigned long long *flags, int me, int other, unsigned long long gen, long long *out, int nblocks_busy)
{
  if (blockIdx.x != 0) { /* filler blocks: keep some CUs busy for a while */ for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(64); return; }
  if (threadIdx.x != 0) return;
  __hip_atomic_store(&flags[me], gen, __ATOMIC_RELEASE, SCOPE);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  long long w = 0;
  while (__hip_atomic_load(&flags[other], __ATOMIC_ACQUIRE, SCOPE) < gen) {
    __builtin_amdgcn_s_sleep(32);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { w = -1; break; }
  }
  out[me] = w < 0 ? -1 : (long long)(__builtin_amdgcn_s_memrealtime() - t0);
}
int main()
{
  unsigned long long *flags; long long *out, h[2];
  hipMalloc(&flags, 64); hipMalloc(&out, 16);
  hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
  for (int scope = 0; scope < 2; scope++) for (int blocks = 1; blocks <= 512; blocks *= 512) for (int rep = 0; rep < 3; rep++) {
    hipMemset(flags, 0, 64); hipMemset(out, 0, 16); hipDeviceSynchronize();
    const unsigned long long gen = 1;
    if (scope == 0) { hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(64), 0, s[0], flags, 0, 1, gen, out, blocks);
                      hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(64), 0, s[1], flags, 1, 0, gen, out, blocks); }
    else { hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(blocks), dim3(64), 0, s[0], flags, 0, 1, gen, out, blocks);
           hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(blocks), dim3(64), 0, s[1], flags, 1, 0, gen, out, blocks); }
    hipDeviceSynchronize();
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("scope %s, %3d blocks per kernel: waited %lld / %lld ticks of 10 ns (-1 = timed out after 1 s)\n", scope ? "system" : "agent", blocks, h[0], h[1]);
  }
  return 0;
}
