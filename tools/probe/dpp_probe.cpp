// dpp_probe.cpp -- which lane does v_mov_b32_dpp wave_shl:1 / wave_shr:1 read on gfx950, and what does it cost next to ds_bpermute?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dpp_probe.cpp -o tools/probe/dpp_probe && tools/probe/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
  const int v = 100 + threadIdx.x;
  o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);        /* wave_shl:1 */
  o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   /* wave_shr:1 */
  o[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);  /* wave_rol:1 */
}
int main()
{
  int *d, h[192];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int r = 0; r < 3; r++) {
    printf("%s: lane0 <- %d, lane1 <- %d, lane15 <- %d, lane16 <- %d, lane31 <- %d, lane32 <- %d, lane62 <- %d, lane63 <- %d\n",
           r == 0 ? "wave_shl:1" : r == 1 ? "wave_shr:1" : "wave_rol:1", h[64 * r], h[64 * r + 1], h[64 * r + 15], h[64 * r + 16], h[64 * r + 31], h[64 * r + 32], h[64 * r + 62], h[64 * r + 63]);
  }
  return 0;
}
