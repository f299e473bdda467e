// walk_probe.cpp -- the chain walk of the sweep kernel (traceLineage's inner loop: follow `next`, read the 16-byte event
// record, patch its lineage count, accumulate three fp64 statistics, list the event, branch on its type) in TWO forms over
// the SAME LDS budget per locus (5 KB: image + sequence block, i.e. 32 resident loci per CU):
//   form A  one locus per wavefront, wave-uniform: scalar control flow, the record through one LDS broadcast read +
//           v_readfirstlane, fp64 arithmetic on replicated lanes -- what k_sweep does (32 wavefronts per CU)
//   form B  W loci per wavefront, a lane per locus: per-lane addresses into W images of the wavefront's LDS allocation
//           (stride 5 KB + 4 bytes: same-field accesses of the lanes fall into different banks), per-lane vector integer
//           code, divergent walk lengths (32 / W wavefronts per CU)
// VERDICT round 3, item 1b asked for this trade-off to be measured instead of argued.  Output: walk steps per CU-cycle.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/walk_probe.cpp -o tools/probe/walk_probe && tools/probe/walk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define EVN 96                 /* events per locus */
#define IMG 5120               /* LDS bytes per locus in form A (what bounds the residency of the real kernel) */
#define STRIDE (IMG + 4)       /* form B: image stride, odd in dwords -> lane l's field lands in bank (l + const) */
struct alignas(16) Ev { double time; short next, prev, node; signed char nlin; unsigned char type; };
typedef __attribute__((address_space(3))) char lchar;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// one walk = `len` steps from event `start`; returns a checksum so that nothing is optimised away
template <int W> __device__ __forceinline__ double walk(lchar *base, int start, int len, double thinv, double &dcoal_out)
{
  double age = 0.0, dcoal = 0.0, lnld = 0.0, mig_rate = 0.0;
  int ev = start, nev = 0;
  if constexpr (W == 1) {
    /* wave-uniform: every value that steers control flow goes through v_readfirstlane */
    ev = __builtin_amdgcn_readfirstlane(ev); len = __builtin_amdgcn_readfirstlane(len);
    for (int k = 0; k < len; k++) {
      const u4 w = *(const __attribute__((address_space(3))) u4 *)(base + ev * 16);
      union { double d; unsigned u[2]; } t; t.u[0] = w.x; t.u[1] = w.y;
      const int w2 = __builtin_amdgcn_readfirstlane((int)w.z), w3 = __builtin_amdgcn_readfirstlane((int)w.w);
      const int next = (short)w2, nlin = (int)(signed char)(w3 >> 16) - 1, type = (int)((unsigned)w3 >> 24);
      ((__attribute__((address_space(3))) signed char *)(base + ev * 16))[14] = (signed char)(nlin + 1);   /* (kept in range) */
      age += t.d;
      dcoal += 2 * nlin * t.d;
      ((__attribute__((address_space(3))) unsigned char *)(base + EVN * 16))[nev] = (unsigned char)ev;
      nev++;
      lnld -= (mig_rate + (2 * nlin) * thinv) * t.d;
      if (type == 3) mig_rate += 0.25; else if (type == 4) mig_rate -= 0.25;
      ev = next;
    }
  } else {
    for (int k = 0; k < len; k++) {       /* divergent: lanes with shorter walks idle until the longest is done */
      const u4 w = *(const __attribute__((address_space(3))) u4 *)(base + ev * 16);
      union { double d; unsigned u[2]; } t; t.u[0] = w.x; t.u[1] = w.y;
      const int next = (short)w.z, nlin = (int)(signed char)(w.w >> 16) - 1, type = (int)(w.w >> 24);
      ((__attribute__((address_space(3))) signed char *)(base + ev * 16))[14] = (signed char)(nlin + 1);
      age += t.d;
      dcoal += 2 * nlin * t.d;
      ((__attribute__((address_space(3))) unsigned char *)(base + EVN * 16))[nev] = (unsigned char)ev;
      nev++;
      lnld -= (mig_rate + (2 * nlin) * thinv) * t.d;
      if (type == 3) mig_rate += 0.25; else if (type == 4) mig_rate -= 0.25;
      ev = next;
    }
  }
  dcoal_out = dcoal;
  return age + lnld;
}

template <int W> __global__ __launch_bounds__(64) void k(const Ev *chains, const int *starts, const int *lens, int nwalk, double *out, long long *steps)
{
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x, wave = blockIdx.x;
  lchar *lds = (lchar *)sm;
  /* stage the W images (identical chains per locus slot, different start / length tables) */
  for (int l = 0; l < W; l++)
    for (int i = lane; i < EVN; i += 64)
      *(__attribute__((address_space(3))) u4 *)(lds + (size_t)l * (W == 1 ? IMG : STRIDE) + i * 16) = ((const u4 *)chains)[((wave * W + l) % 4096) * EVN + i];
  __syncthreads();
  double acc = 0.0, dc = 0.0;
  long long st = 0;
  if (W == 1 || lane < W) {
    const int locus = wave * W + (W == 1 ? 0 : lane);
    lchar *base = lds + (size_t)(W == 1 ? 0 : lane) * (W == 1 ? IMG : STRIDE);
    for (int i = 0; i < nwalk; i++) {
      const int idx = (locus * 131 + i) & 65535;
      const int len = lens[idx];
      acc += walk<W>(base, starts[idx], len, 1.0 / 3.0, dc);
      acc += dc * 1e-30;
      st += len;
    }
  }
  if (W == 1 ? lane == 0 : lane < W) { out[wave * 64 + lane] = acc; atomicAdd((unsigned long long *)steps, (unsigned long long)st); }
}

template <int W> static void run(const Ev *chains, const int *starts, const int *lens, double *out, long long *steps, int nwalk)
{
  const int loci = 256 * 32 * 4;            /* four rounds of the 32 loci a CU holds */
  const int waves = loci / W;
  const size_t lds = W == 1 ? IMG : (size_t)W * STRIDE;
  hipFuncSetAttribute((const void *)k<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  long long st = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(steps, 0, sizeof(long long));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(waves), dim3(64), lds, 0, chains, starts, lens, nwalk, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(&st, steps, sizeof st, hipMemcpyDeviceToHost);
  }
  const double cyc = best * 1e-3 * 2.4e9;
  printf("W = %2d loci per wavefront: %6d wavefronts (%2d resident per CU), %.3f ms, %lld walk steps: %.4f steps per CU-cycle, %.1f CU-cycles per step\n", W, waves,
         (int)(160 * 1024 / lds > 32 ? 32 : 160 * 1024 / lds), best, st, st / (cyc * 256), cyc * 256 / st);
}

int main()
{
  std::vector<Ev> ch(4096 * EVN);
  srand(7);
  for (int c = 0; c < 4096; c++) {
    int perm[EVN];
    for (int i = 0; i < EVN; i++) perm[i] = i;
    for (int i = EVN - 1; i > 0; i--) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (int i = 0; i < EVN; i++) {
      Ev &e = ch[c * EVN + perm[i]];
      e.time = 1e-6 * (1 + rand() % 100); e.next = (short)perm[(i + 1) % EVN]; e.prev = (short)perm[(i + EVN - 1) % EVN];
      e.node = (short)i; e.nlin = (signed char)(2 + rand() % 10); e.type = (unsigned char)(rand() % 8);
    }
  }
  std::vector<int> st(65536), ln(65536);
  for (int i = 0; i < 65536; i++) { st[i] = rand() % EVN; ln[i] = 4 + rand() % 17; }     /* 4 .. 20 intervals per walk, mean 12 */
  Ev *dch; int *dst, *dln; double *out; long long *steps;
  hipMalloc(&dch, ch.size() * sizeof(Ev)); hipMalloc(&dst, 65536 * 4); hipMalloc(&dln, 65536 * 4); hipMalloc(&out, sizeof(double) * 64 * 32768); hipMalloc(&steps, 8);
  hipMemcpy(dch, ch.data(), ch.size() * sizeof(Ev), hipMemcpyHostToDevice);
  hipMemcpy(dst, st.data(), 65536 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dln, ln.data(), 65536 * 4, hipMemcpyHostToDevice);
  printf("# chain walk, %d events per locus, walks of 4-20 steps, 600 walks per locus; LDS per locus %d B (form A) / %d B (form B)\n", EVN, IMG, STRIDE);
  const int nwalk = 600;
  run<1>(dch, dst, dln, out, steps, nwalk);
  run<2>(dch, dst, dln, out, steps, nwalk);
  run<4>(dch, dst, dln, out, steps, nwalk);
  run<8>(dch, dst, dln, out, steps, nwalk);
  run<16>(dch, dst, dln, out, steps, nwalk);
  return 0;
}
