// node_step_probe.cpp -- VERDICT round 5, item 2c: what would a (pattern, base) lane mapping of the pruning step cost?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe/node_step_probe.cpp -o /tmp/node_step_probe && /tmp/node_step_probe
//
// One pruning step = computeSubtreeConditionals_new twice + the product (LocusDataLikelihood.c:1596-1673), bit-faithful:
//   S = ((s0 + s1) + s2) + s3;  Sp = S * pe;  f_b = Sp + s_b * qe   per child, then q_b = f_b(left) * f_b(right).
// Form A (what k_sweep runs): a lane per PATTERN holds the four bases of a child in registers: 3 + 1 + 8 = 12 fp64 operations per
//   child, 4 for the product: 28 per step, at most 64 patterns per step, 18 of 64 lanes busy at the benchmark's mean P.
// Form B: a lane per (PATTERN, BASE): 4 lanes per pattern, 16 patterns per step.  Every lane needs S in the reference's order, i.e.
//   all four values of its quad: v_add_f64 is VOP3 and takes no DPP modifier on gfx950, so each of the four values is broadcast
//   over the quad with two v_mov_b32_dpp quad_perm (low and high half): 8 moves + 3 adds + 1 + 2 = 6 fp64 per child, 1 for the
//   product: 13 fp64 + 16 DPP moves per step -- and loci with 17 .. 64 patterns need 2 .. 4 steps where form A needs one.
// Both kernels run STEPS dependent steps on register data (the new node is the next step's left child, the previous one its right
// child: nothing can be hoisted), 8 wavefronts per SIMD; reported: cycles one step holds its SIMD, per wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define STEPS 4096

__device__ __forceinline__ void factor4(double s0, double s1, double s2, double s3, double pe, double qe, double &f0, double &f1, double &f2, double &f3)
{
  double S = s0;
  S += s1; S += s2; S += s3;
  const double Sp = S * pe;
  f0 = Sp + s0 * qe; f1 = Sp + s1 * qe; f2 = Sp + s2 * qe; f3 = Sp + s3 * qe;
}
__global__ __launch_bounds__(64) void form_a(double *out, double pe0, int steps)
{
  const int lane = threadIdx.x;
  double q0 = 0.25 + 1e-3 * lane, q1 = 0.25, q2 = 0.26, q3 = 0.24, p0 = 0.24, p1 = 0.26, p2 = 0.25, p3 = 0.25;
  double pe = pe0, qe = 1 - 4.0 * pe;
  for (int i = 0; i < steps; i++) {
    double a0, a1, a2, a3, b0, b1, b2, b3;
    factor4(q0, q1, q2, q3, pe, qe, a0, a1, a2, a3);
    factor4(p0, p1, p2, p3, pe, qe, b0, b1, b2, b3);
    p0 = q0; p1 = q1; p2 = q2; p3 = q3;
    q0 = a0 * b0 * 16.0; q1 = a1 * b1 * 16.0; q2 = a2 * b2 * 16.0; q3 = a3 * b3 * 16.0;    /* (x 16: keeps the values near 1/4; 4 extra multiplies in BOTH forms' accounting below) */
  }
  out[blockIdx.x * 64 + lane] = q0 + q1 + q2 + q3;
}
__device__ __forceinline__ double quad_bcast(double v, int which)
{
  union { double d; int i[2]; } u;
  u.d = v;
  switch (which) {
  case 0: u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0x00, 0xf, 0xf, true); u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0x00, 0xf, 0xf, true); break;
  case 1: u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0x55, 0xf, 0xf, true); u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0x55, 0xf, 0xf, true); break;
  case 2: u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0xaa, 0xf, 0xf, true); u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0xaa, 0xf, 0xf, true); break;
  default: u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0xff, 0xf, 0xf, true); u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0xff, 0xf, 0xf, true); break;
  }
  return u.d;
}
__device__ __forceinline__ double factor1(double s, double pe, double qe)
{
  double S = quad_bcast(s, 0);
  S += quad_bcast(s, 1); S += quad_bcast(s, 2); S += quad_bcast(s, 3);
  const double Sp = S * pe;
  return Sp + s * qe;
}
__global__ __launch_bounds__(64) void form_b(double *out, double pe0, int steps)
{
  const int lane = threadIdx.x;
  double q = 0.25 + 1e-3 * (lane & 3), p = 0.25 - 1e-3 * (lane & 3);
  double pe = pe0, qe = 1 - 4.0 * pe;
  for (int i = 0; i < steps; i++) {
    const double a = factor1(q, pe, qe), b = factor1(p, pe, qe);
    p = q;
    q = a * b * 16.0;
  }
  out[blockIdx.x * 64 + lane] = q;
}
int main()
{
  const int blocks = 256 * 4 * 8;
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2] = {0, 0};
  for (int rep = 0; rep < 3; rep++)
    for (int form = 0; form < 2; form++) {
      hipEventRecord(e0);
      if (form == 0) hipLaunchKernelGGL(form_a, dim3(blocks), dim3(64), 0, 0, out, 0.1, STEPS);
      else hipLaunchKernelGGL(form_b, dim3(blocks), dim3(64), 0, 0, out, 0.1, STEPS);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[form], e0, e1);
    }
  const double cyc_a = ms[0] * 1e-3 * 2.4e9 / (8.0 * STEPS), cyc_b = ms[1] * 1e-3 * 2.4e9 / (8.0 * STEPS);
  printf("# pruning step on register data, 8 wavefronts per SIMD, %d dependent steps per wavefront, cycles at 2.4 GHz per step and wavefront on its SIMD\n", STEPS);
  printf("form A  lane per pattern        (28 + 4 fp64 per step, <= 64 patterns per step):  %7.1f cycles per step\n", cyc_a);
  printf("form B  lane per (pattern,base) (13 + 1 fp64 + 16 v_mov_b32_dpp per step, <= 16 patterns per step):  %7.1f cycles per step\n", cyc_b);
  printf("per locus at P patterns (steps needed: A ceil(P/64), B ceil(P/16)):\n");
  for (int P : {8, 12, 16, 18, 24, 32, 48, 64}) {
    const int sa = (P + 63) / 64, sb = (P + 15) / 16;
    printf("  P = %2d:  A %7.1f   B %7.1f   B / A = %.2f\n", P, sa * cyc_a, sb * cyc_b, sb * cyc_b / (sa * cyc_a));
  }
  return 0;
}
