// gph_engine.hip -- host side of the engine + __global__ wrappers (C ABI: include/gphocs_hip.h).
//
// Launch geometry: one 64-lane workgroup (one wavefront) per locus, grid = loci; the
// per-wave dynamic LDS holds the whole mutable state of the locus, so the number of
// resident waves per CU is LDS-bound (160 KiB / lds_bytes) -- L >> 256 CUs x waves/CU
// keeps every XCD busy and needs no blockIdx remap (loci are independent, nothing is
// shared between workgroups, so L2 affinity is irrelevant).
//
// An MCMC iteration is ONE stream of launches (gph_engine_iteration_): per-locus kernel -> fixed-shape reduction
// -> [RCCL all-gather of the reduced row, on the same stream] -> k_global (the decision above the loci, one
// wavefront, gph_global.h) -> predicated commit / revert kernel -> ... and a single host synchronisation at its end.
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <vector>
#include <algorithm>
#if defined(GPH_HOSTEMU) && defined(GPH_EMU64)
#define GPH_EMU64_IMPL      /* this translation unit holds the micro-wave scheduler (gph_emu64.h; test builds only) */
#endif
#include "gph_kernels.h"
#include "gph_global.h"
#include "gph_comm.h"
#include "../../include/gphocs_hip.h"

#ifdef GPH_HOSTEMU
thread_local char *gph_sm = nullptr;
thread_local GphLds gph_lds;
GphLayout g_lay;
GphModel g_model;
GphGlobal *gph_G_emu = nullptr;
#ifdef GPH_BOUNDS
int gph_oob_word = 0;
#endif
#define GPH_KERNEL(name, ...) static void name(int gph_blk, __VA_ARGS__)
#define GPH_SWEEP_WAVES_ 6
#define GPH_SWEEP_ATTR
#define GPH_BLK gph_blk
#else
#define GPH_KERNEL(name, ...) __global__ __launch_bounds__(GPH_WAVE) void name(__VA_ARGS__)
#define GPH_BLK ((int)blockIdx.x)
// wavefronts per SIMD the sweep kernel is compiled for (the build passes it per capacity variant): 6 = 80 VGPRs, which
// every variant reaches without a spill -- beyond that LDS decides how many loci are resident; variant s is built for 8
#ifndef GPH_SWEEP_WAVES
#define GPH_SWEEP_WAVES 6
#endif
#define GPH_SWEEP_ATTR __attribute__((amdgpu_waves_per_eu(GPH_SWEEP_WAVES, GPH_SWEEP_WAVES)))
#define GPH_SWEEP_WAVES_ GPH_SWEEP_WAVES
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "gphocs_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return GPH_EHIP; } } while (0)
#endif

// j0 = first slot of the launch group (see GphDev).  GphCtx reads the model from the kernel-argument segment (the host
// knows it when it launches: the genealogy sweep at the head of an iteration, the locus-rate kernels), GphCtxG from the
// device-resident chain state KA.G (everything launched after a decision the host has not seen yet)
GPH_KERNEL(k_init, GphKargs KA, GphDev D, int j0, uint32_t seedz, const double *mutRate, int preDraws) { GphCtxG lx; lx.kb_init(D, j0 + GPH_BLK, seedz, mutRate ? mutRate[j0 + GPH_BLK] : 1.0, preDraws); }
GPH_SWEEP_ATTR GPH_KERNEL(k_sweep, GphKargs KA, GphDev D, int j0, int flags, double ftCoal, double ftMig, double mix_c, double mix_lnc) { GphCtx lx; lx.kb_sweep(D, j0 + GPH_BLK, flags, ftCoal, ftMig, mix_c, mix_lnc); }
GPH_KERNEL(k_tau_eval, GphKargs KA, GphDev D, int j0, int fuse) { GphCtxG lx; lx.kb_tau_eval(D, j0 + GPH_BLK, GPH_G->tau, fuse); }
GPH_KERNEL(k_tau_finish, GphKargs KA, GphDev D, int j0, int unused) { (void)unused; GphCtxG lx; lx.kb_tau_finish(D, j0 + GPH_BLK); }
GPH_KERNEL(k_mix_eval, GphKargs KA, GphDev D, int j0, int fuse) { GphCtxG lx; lx.kb_mix_eval(D, j0 + GPH_BLK, GPH_G->mix_c, fuse); }
GPH_KERNEL(k_mix_finish, GphKargs KA, GphDev D, int j0, int unused) { (void)unused; GphCtxG lx; lx.kb_mix_finish(D, j0 + GPH_BLK); }
GPH_KERNEL(k_sync, GphKargs KA, GphDev D, int j0, int refresh) { GphCtxG lx; lx.kb_sync(D, j0 + GPH_BLK, refresh); }
GPH_KERNEL(k_check, GphKargs KA, GphDev D, int j0, int unused) { (void)unused; GphCtxG lx; lx.kb_check(D, j0 + GPH_BLK); }
GPH_KERNEL(k_unit, GphKargs KA, GphDev D, int j0, int op, int arg, double *out, int stride) { GphCtxG lx; lx.kb_unit(D, j0 + GPH_BLK, op, arg, out, stride); }
GPH_KERNEL(k_lrate_prep, GphKargs KA, GphDev D, int j0, double finetune, GphLrPre *pre) { GphCtx lx; lx.kb_lrate_prep(D, j0 + GPH_BLK, finetune, pre); }
GPH_KERNEL(k_lrate_scan, GphKargs KA, GphDev D, int j0, GphLrArgs A) { (void)j0; GphCtx lx; lx.kb_lrate_scan(D, A); }
GPH_KERNEL(k_lrate_apply, GphKargs KA, GphDev D, int j0, const GphLrRec *rec) { GphCtx lx; lx.kb_lrate_apply(D, j0 + GPH_BLK, rec); }

// ---------------------------------------------------------------- small elementwise / reduction kernels
#ifndef GPH_RED_BLOCKS
#define GPH_RED_BLOCKS 256
#endif
static_assert(2 * GPH_CAP_K + 2 * GPH_CAP_B <= GPH_RED_COLS && GPH_OUT_SLOTS <= 64, "the reduction kernels fold at most GPH_RED_COLS columns");

// UpdateTheta / UpdateMigRates accepted: the genLogLikelihood touch-ups of GPhoCS.c:3084-3093 and :3192-3200, every
// accepted proposal of the iteration in the order it was accepted (the list is in the chain state); one thread per locus
GPH_HD void apply_list_locus(char *pg, const GphLayout &y, const GphApply *ap, int na)
{
  double *fs = (double *)(pg + y.o_fscal);
  double v = fs[FS_GENLNL];
  for (int i = 0; i < na; i++) {
    const int idx = ap[i].idx;
    if (ap[i].kind == 0) {
      const int nc = ((int16_t *)(pg + y.o_ncoal))[idx];
      const double cs = ((double *)(pg + y.o_coal))[idx];
      v -= (ap[i].lnc * nc + ap[i].diff * cs);
    } else {
      const int nm = ((int16_t *)(pg + y.o_nmig))[idx];
      const double ms = ((double *)(pg + y.o_migst))[idx];
      v += (ap[i].lnc * nm - ap[i].diff * ms);
    }
  }
  fs[FS_GENLNL] = v;
}
#ifndef GPH_HOSTEMU
__global__ void k_apply_list(GphKargs KA, GphDev D)
{
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int na = KA.G->napply;
  if (g >= D.L || na == 0) return;
  apply_list_locus(D.pages + (size_t)g * KA.lay.page_bytes, KA.lay, KA.G->apply, na);
}
// the stage above the loci (gph_global.h): one wavefront, every lane runs the same scalar code, lane 0's stores count
// ---- the stage(s) above the loci (gph_global.h) and the reductions that feed them, as ONE launch.
// The chain state (12 KB) and the ranks' reduced rows are staged through LDS by the whole block -- coalesced copies
// instead of a chain of dependent HBM round trips of a lone thread; thread 0 runs the stages in order.
struct GphStageList { int n; int stage[6]; int arg[6]; };
#define GPH_RED_SUBS 8
#define GPH_RED_THREADS (GPH_RED_SUBS * 64)
struct alignas(16) GphStageShared {
  char g[sizeof(GphGlobal)];
  double r[GPH_RED_ROW];
};
__device__ void stages_body(GphKargs &KA, const double *rows, int world, const GphStageList &SL, int iteration, GphStageShared &sh)
{
  typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(GphGlobal) % 16 == 0, "the chain state is copied in 16-byte units");
  const int tid = threadIdx.x, nt = blockDim.x;
  const gu32x4 *src = (const gu32x4 *)KA.G;
  gu32x4 *dst = (gu32x4 *)sh.g;
  for (int i = tid; i < (int)(sizeof(GphGlobal) / 16); i += nt) dst[i] = src[i];
  /* the ranks' rows combined in rank order (the same additions on every rank), one column per thread step */
  for (int c = tid; c < GPH_RED_ROW; c += nt) {
    const int within = c % GPH_RED_STRIDE, kind = within < 3 * GPH_RED_COLS ? within / GPH_RED_COLS : 3;
    double v = rows[c];
    for (int r = 1; r < world; r++) {
      const double w = rows[(size_t)r * GPH_RED_ROW + c];
      v = kind == 0 ? v + w : kind == 1 ? (w < v ? w : v) : (w > v ? w : v);
    }
    sh.r[c] = v;
  }
  __syncthreads();
  if (tid == 0) {
    GphGlobal &G = *(GphGlobal *)sh.g;
    G.iteration = iteration;
    GphRed R;
    R.rows = sh.r; R.world = 1;
    for (int k = 0; k < SL.n; k++) gg_stage(G, R, SL.stage[k], SL.arg[k]);
  }
  __syncthreads();
  gu32x4 *out = (gu32x4 *)KA.G;
  for (int i = tid; i < (int)(sizeof(GphGlobal) / 16); i += nt) out[i] = dst[i];
}
__global__ void __launch_bounds__(GPH_RED_THREADS) k_global(GphKargs KA, const double *rows, int world, GphStageList SL, int iteration)
{
  __shared__ __attribute__((aligned(16))) GphStageShared sh;
  stages_body(KA, rows, world, SL, iteration, sh);
}

// fixed-shape two-level reduction (deterministic run to run): block b owns a contiguous chunk of
// loci; inside it Q = 8 * (64 / column width) sub-sequences (g = g0+q, g0+q+Q, ...) are summed in index
// order and combined in sub-sequence order; the final pass adds the 256 block partials in a fixed shape.
// mode 0 = per-locus outputs (GPH_OUT_SLOTS columns), mode 1 = compact statistics (2K+2B columns)
// part: [3][GPH_RED_BLOCKS][GPH_RED_COLS] (sum, min, max) per section
// one sub-sequence of a column (g, g + step, ... below g1) folded in index order.  The loads of GPH_RED_UNROLL steps are
// issued together and folded as they arrive: one load in flight per thread made these loops a chain of memory
// latencies (a dozen to fifty of them per reduction, six reductions per iteration)
#define GPH_RED_UNROLL 16
__device__ __forceinline__ void reduce_run(const double *src, int stride, int col, int g, int g1, int step, double &s, double &mn, double &mx)
{
  for (; g < g1; g += step * GPH_RED_UNROLL) {
    double v[GPH_RED_UNROLL];
#pragma unroll
    for (int u = 0; u < GPH_RED_UNROLL; u++) {
      const int gg = g + u * step;
      v[u] = gg < g1 ? src[(size_t)gg * stride + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < GPH_RED_UNROLL; u++)
      if (g + u * step < g1) { s += v[u]; mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
  }
}
// cbase: first column of the window this call folds (at most 128 columns a call; the many-band build's statistics row is
// folded window by window -- every column's additions are the same whatever the windows)
__device__ void reduce_partial_body(const GphDev &D, int mode, int ncols, int cbase, int stride_cols, double *part, double (*sh)[GPH_RED_SUBS * 4][16])
{
  /* a wavefront covers 64 / cw loci at a time (cw = columns rounded up to 16, 32 or 64): every lane has work */
  const int cw = ncols <= 16 ? 16 : ncols <= 32 ? 32 : ncols <= 64 ? 64 : 128;
  const int b = blockIdx.x;
  const int chunk = (D.L + GPH_RED_BLOCKS - 1) / GPH_RED_BLOCKS;
  const int g0 = b * chunk, g1 = g0 + chunk < D.L ? g0 + chunk : D.L;
  const double *src = (mode == 0 ? D.out : D.stats) + cbase;
  const int stride = mode == 0 ? GPH_OUT_SLOTS : stride_cols;
  part += cbase;
  if (cw == 128) {
    /* more than 64 columns (the largest capacity variant): two columns per lane, one locus per wavefront step */
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int half = 0; half < 2; half++) {
      const int col = lane + 64 * half;
      double s = 0.0, mn = 1e300, mx = -1e300;
      if (col < ncols) reduce_run(src, stride, col, g0 + q, g1, GPH_RED_SUBS, s, mn, mx);
      for (int c0 = 64 * half; c0 < ncols && c0 < 64 * (half + 1); c0 += 16) {
        __syncthreads();
        if (col >= c0 && col < c0 + 16) { sh[0][q][col - c0] = s; sh[1][q][col - c0] = mn; sh[2][q][col - c0] = mx; }
        __syncthreads();
        if (threadIdx.x < 16 && c0 + (int)threadIdx.x < ncols) {
          const int c = threadIdx.x;
          double ts = 0.0, tmn = 1e300, tmx = -1e300;
          for (int k = 0; k < GPH_RED_SUBS; k++) {
            ts += sh[0][k][c];
            tmn = sh[1][k][c] < tmn ? sh[1][k][c] : tmn;
            tmx = sh[2][k][c] > tmx ? sh[2][k][c] : tmx;
          }
          part[(0 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = ts;
          part[(1 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = tmn;
          part[(2 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = tmx;
        }
      }
    }
    __syncthreads();
    return;
  }
  const int ng = 64 / cw, Q = GPH_RED_SUBS * ng;
  const int lane = threadIdx.x & 63, col = lane % cw, q = (threadIdx.x >> 6) * ng + lane / cw;
  double s = 0.0, mn = 1e300, mx = -1e300;
  if (col < ncols) reduce_run(src, stride, col, g0 + q, g1, Q, s, mn, mx);
  /* combine the Q sub-sequences of a column in sub-sequence order; 16 columns at a time through shared memory */
  for (int c0 = 0; c0 < ncols; c0 += 16) {
    __syncthreads();
    if (col >= c0 && col < c0 + 16) { sh[0][q][col - c0] = s; sh[1][q][col - c0] = mn; sh[2][q][col - c0] = mx; }
    __syncthreads();
    if (threadIdx.x < 16 && c0 + (int)threadIdx.x < ncols) {
      const int c = threadIdx.x;
      double ts = 0.0, tmn = 1e300, tmx = -1e300;
      for (int k = 0; k < Q; k++) {
        ts += sh[0][k][c];
        tmn = sh[1][k][c] < tmn ? sh[1][k][c] : tmn;
        tmx = sh[2][k][c] > tmx ? sh[2][k][c] : tmx;
      }
      part[(0 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = ts;
      part[(1 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = tmn;
      part[(2 * GPH_RED_BLOCKS + b) * GPH_RED_COLS + c0 + c] = tmx;
    }
  }
  __syncthreads();
}
// final pass: column c's block partials in block order, as GPH_RED_SUBS contiguous runs combined in run order -- a
// fixed shape, so the result does not depend on scheduling.  Up to 64 columns (every variant but the largest): thread
// (run, column), one run each; beyond: thread (h, column) folds runs h and h + 4.  A run's 3 x 32 partials are fetched
// 48 loads at a time: these are reads of other XCDs' writes, a microsecond or two each
__device__ void reduce_final_body(int ncols, int cbase, const double *part, double *red, double (*sh)[GPH_RED_SUBS][128])
{
  part += cbase; red += cbase;
  const int cw = ncols <= 64 ? 64 : 128;
  const int col = threadIdx.x % cw, h = threadIdx.x / cw;
  const int per = GPH_RED_BLOCKS / GPH_RED_SUBS;
  static_assert((GPH_RED_BLOCKS / GPH_RED_SUBS) % 16 == 0, "the final pass folds sixteen block partials per step");
  for (int sub = h; sub < GPH_RED_SUBS; sub += GPH_RED_THREADS / cw) {
    double s = 0.0, mn = 1e300, mx = -1e300;
    if (col < ncols)
      for (int b0 = sub * per; b0 < (sub + 1) * per; b0 += 16) {
        double vs[16], vn[16], vx[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          vs[u] = part[(0 * GPH_RED_BLOCKS + b0 + u) * GPH_RED_COLS + col];
          vn[u] = part[(1 * GPH_RED_BLOCKS + b0 + u) * GPH_RED_COLS + col];
          vx[u] = part[(2 * GPH_RED_BLOCKS + b0 + u) * GPH_RED_COLS + col];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) { s += vs[u]; mn = vn[u] < mn ? vn[u] : mn; mx = vx[u] > mx ? vx[u] : mx; }
      }
    if (col < 128) { sh[0][sub][col] = s; sh[1][sub][col] = mn; sh[2][sub][col] = mx; }
  }
  __syncthreads();
  if (h == 0 && col < ncols) {
    double s = 0.0, mn = 1e300, mx = -1e300;
    for (int k = 0; k < GPH_RED_SUBS; k++) {
      s += sh[0][k][col];
      mn = sh[1][k][col] < mn ? sh[1][k][col] : mn;
      mx = sh[2][k][col] > mx ? sh[2][k][col] : mx;
    }
    red[col] = s;
    red[GPH_RED_COLS + col] = mn;
    red[2 * GPH_RED_COLS + col] = mx;
  }
  __syncthreads();
}
// reduction of section 0 (nc0 columns of the per-locus outputs) and / or section 1 (nc1 columns of the statistics) into
// this rank's row; the LAST block to finish its partials (ticket) does the final pass and -- single rank -- runs the
// stages that consume the row in the same launch
// In-kernel exchange of the reduced rows between ranks whose kernels can address each other's memory (gph_comm_peer_exchange):
// rank r owns slots rows[(2 r + parity) * stride ...] and the generation word flags[r]; `gather` is this rank's copy of
// everybody's row of the current generation.
struct GphPeerX { int world, rank, stride, pad; double *rows; unsigned long long *flags; double *gather; unsigned long long gen; };
__global__ void __launch_bounds__(GPH_RED_THREADS) k_reduce_stage(GphKargs KA, GphDev D, int nc0, int nc1, double *part, unsigned *ticket, double *red,
                                                                  int do_stage, GphStageList SL, int iteration, GphPeerX X)
{
  __shared__ union { double p[3][GPH_RED_SUBS * 4][16]; double f[3][GPH_RED_SUBS][128]; GphStageShared st; } sh;
  __shared__ int s_last;
  if (nc0 > 0) reduce_partial_body(D, 0, nc0, 0, nc0, part, sh.p);
  for (int cb = 0; cb < nc1; cb += 128) reduce_partial_body(D, 1, nc1 - cb < 128 ? nc1 - cb : 128, cb, nc1, part + 3 * GPH_RED_BLOCKS * GPH_RED_COLS, sh.p);
  __syncthreads();
  /* ONE release per block (thread 0, after the block barrier: cumulative over the block's stores) -- an agent-scope
   * fence writes the L2 back, and 2048 wavefronts doing it cost more than the reduction itself */
  if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(ticket, 1u) == gridDim.x - 1; }
  __syncthreads();
  if (!s_last) return;
  /* every wavefront of the last block acquires for itself before it reads the other blocks' partials (they were
   * written through other XCDs' L2s): one agent-scope fence per wavefront of ONE block -- the memory model asks for
   * it, the cache-wide invalidate of thread 0's fence only happened to cover the other waves */
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (nc0 > 0) reduce_final_body(nc0, 0, part, red, sh.f);
  for (int cb = 0; cb < nc1; cb += 128) reduce_final_body(nc1 - cb < 128 ? nc1 - cb : 128, cb, part + 3 * GPH_RED_BLOCKS * GPH_RED_COLS, red + GPH_RED_STRIDE, sh.f);
  if (threadIdx.x == 0) { red[3 * GPH_RED_COLS] = (double)*D.err; *ticket = 0; }
  __syncthreads();
  const double *rows = red;
  int world = 1;
  if (X.world > 1) {
    /* publish this rank's row in its slot of this generation's parity, then the generation word (release, system scope:
     * the layout and the scopes are those of slots in peer-mapped memory); wait for every other rank's word; copy the rows
     * in rank order.  Two slots suffice: a rank can only finish generation g + 1 after everybody has published g + 1, i.e.
     * has finished reading generation g.  The wait is BOUNDED (a rank that never comes must not hang the device): after
     * ~4 s of the 100-MHz counter the launch's error word is raised and the run ends with GPH_EKERNEL. */
    double *mine = X.rows + (size_t)(2 * X.rank + (int)(X.gen & 1)) * X.stride;
    for (int c = threadIdx.x; c < GPH_RED_ROW; c += blockDim.x) mine[c] = red[c];
    /* EVERY storing wavefront drains and releases its own stores before the barrier: thread 0's release below covers only
     * its own wavefront's (MI355X_MICROARCH.md, inter-workgroup visibility) -- without this the other ranks read a row
     * that is partly the previous generation's, and a garbage conflict-locus index sent a later kernel to address 0 */
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      __hip_atomic_store(&X.flags[X.rank], X.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int r = 0; r < X.world; r++) {
        while (__hip_atomic_load(&X.flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < X.gen) {
          __builtin_amdgcn_s_sleep(32);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) {
            atomicMax(D.err, 9997); red[3 * GPH_RED_COLS] = 9997.0; r = X.world; break; }
        }
      }
    }
    __syncthreads();
    __threadfence_system();      /* every reading wavefront acquires for itself */
    for (int r = 0; r < X.world; r++) {
      const double *src = r == X.rank ? red : X.rows + (size_t)(2 * r + (int)(X.gen & 1)) * X.stride;
      for (int c = threadIdx.x; c < GPH_RED_ROW; c += blockDim.x) X.gather[(size_t)r * GPH_RED_ROW + c] = src[c];
    }
    __syncthreads();
    rows = X.gather; world = X.world;
  }
  /* (ONE call site: with two the compiler stops inlining the stage body, the by-reference kernel arguments go through a
   * 72-byte stack frame, and the kernel faults at address 0 on its first launch with stages -- measured, round 5) */
  if (do_stage) stages_body(KA, rows, world, SL, iteration, sh.st);
}
#endif

#ifndef GPH_HOSTEMU
// parity probe: the device's exp/log (gph_math.h) and its native sqrt / divide / floor
__global__ void k_debug_math(GphKargs KA, const double *x, const double *y, int n, double *o)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[i] = gph_exp(x[i]);
  o[n + i] = gph_log(x[i]);
  o[2 * n + i] = sqrt(fabs(x[i]));
  o[3 * n + i] = x[i] / y[i];
  o[4 * n + i] = floor(x[i]);
}
#endif

// ---------------------------------------------------------------- engine object
struct gph_engine {
  gph_config cfg;
  std::vector<int32_t> samplesPerPop, popFather, popSon0, popSon1, bandSrc, bandTgt;
  GphLayout lay;
  GphDev dev;
  // chain state above the loci (gph_types.h: GphGlobal): host mirror (pinned) and the copy in HBM the kernels read.
  // Between iterations both are equal; inside an iteration the device copy leads (resident mode) or the host
  // mirror does (host mode: a caller-supplied all-reduce hook forces a synchronisation per reduction anyway)
  GphGlobal *G_h = nullptr, *G_d = nullptr;
  bool G_dirty = true;               // the host mirror was changed since it was last pushed
  int64_t n_huge = 0;                // loci whose sequence block stays in HBM (the first slots)
  double *peer_rows = nullptr;       // in-kernel exchange of the reduced rows (gph_comm_peer_exchange), else null
  unsigned long long *peer_flags = nullptr;
  std::vector<void *> deferred_free; // device blocks released while the in-kernel exchange is on: hipFree waits for the device, and a peer's kernel may be waiting for THIS rank
  int32_t peer_stride = 0;
  int32_t last_error_code = 0;       // the last fatal error check_error reported (gph_engine_last_error)
  long long last_error_locus = -1;
  bool in_error_dump = false;
  int64_t L = 0;
  size_t cond_bytes = 0, pages_bytes = 0, seq_bytes_total = 0;
  std::vector<uint64_t> h_cond_off;
  std::vector<int32_t> h_P;          // per slot (sorted order)
  std::vector<int32_t> h_orig;       // slot -> original local index
  struct Bucket { int j0, count, lds_bytes; };   // a launch group: slots [j0, j0+count), dynamic LDS per wave
  std::vector<Bucket> buckets;
  double *d_mutRate = nullptr;
  int init_predraws = 0;             // rndu() draws every locus spent before its genealogy is sampled (VAR start-up: 1)
  bool var_rates = false;            // locus rates are part of the state (dumped as "R" lines)
  // UpdateLocusRate: per-slot records, input-order -> slot map, result scalars, scratch for pattern-rich loci
  GphLrRec *d_lrec = nullptr;
  GphLrPre *d_lpre = nullptr;
  double lr_hits = 0;                // proposals of the last update decided with the prepared likelihood
  int32_t *d_slot_of = nullptr;
  double *d_lr_result = nullptr, *d_lr_gscr = nullptr;
  char *d_ref_page = nullptr, *d_ref_seq = nullptr;   // the reference locus's node records / scalars and sequence block on this rank
  std::vector<uint64_t> h_seq_off;
  GphLrArgs lr;
  int lr_lds_bytes = 0;
  int lds_pad[16] = {0};             // GPH_LDS_PAD="class:bytes,...": extra dynamic LDS per kernel class (occupancy experiments)
  // reductions: block partials, this rank's reduced row (GPH_RED_ROW doubles) and every rank's row after the all-gather
  double *d_part = nullptr, *d_red = nullptr, *d_gather = nullptr;
#ifndef GPH_HOSTEMU
  unsigned *d_ticket = nullptr;                  // last-block ticket of the fused reduction
  // resident mode: reductions and stages requested since the last launch; they go out as ONE kernel when the next
  // per-locus kernel is launched (or the host needs the result)
  int pend_nc0 = 0, pend_nc1 = 0;
  GphStageList pend_sl = {0, {0}, {0}};
  int pend_iteration = 0;
#endif
  double *h_red = nullptr;                       // pinned: this rank's row (host mode)
  // several ranks, one process per GPU: a native communicator (RCCL all-gather on the engine's stream, or -- ranks
  // sharing one GPU, tests only -- a host shared-memory exchange) or a caller-supplied hook
  gph_comm *comm = nullptr;
  gph_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  bool force_host = false;           // GPH_HOST_DECISIONS=1: decisions on the host although nothing requires it (tests)
  int64_t n_syncs = 0, n_collectives = 0, n_launches = 0;   // host synchronisations / cross-rank exchanges / kernel launches so far
  uint32_t seedz = 0;
  bool loaded = false, seeded = false, model_set = false, initialized = false;
  GphKargs ka;                 // first argument of every kernel: model, layout, math constants, table addresses
  bool sync_pending = false;   // synchronizeEvents of the finished iteration rides at the head of the next sweep kernel
  bool fin_owed = false;       // the commit / revert of the last decided tau / sample-age proposal has not run yet
  bool mix_owed = false;       // the commit of the last decided mixing proposal has not run yet (it rides at the head of the next sweep kernel)
  int32_t slog_sel = 0;        // loci selected for the decision-level transcript (GPH_LOGSTEPS builds)
  bool mirror_current = true;  // the host mirror G_h holds what the device-side stages last wrote (false between a queued stage and pull_G)
  bool no_fuse = false;        // GPH_NO_FUSE=1 (tests): every finish as a kernel of its own; read once in gph_engine_create
  gph_counters counters = {0, 0, 0.0, 0};
  double last_ms[16] = {0};
  // per kernel class: launches, summed HIP-event ms; evaluations / bytes / nodes live in the chain state
  double cls_launches[16] = {0}, cls_ms[16] = {0};
  double cls_evals0[16] = {0}, cls_bytes0[16] = {0}, cls_nodes0[16] = {0};   // offsets of the last reset
  int last_which = 0;
  uint32_t timing_mask = 0xffffffffu;   // classes whose launches are bracketed by HIP events
#ifndef GPH_HOSTEMU
  hipStream_t stream = nullptr;
  // the launch group of the pattern-rich loci (P > 64: a few per cent of a data set, long wavefronts) runs NEXT TO the
  // main group on a stream of its own, forked from and joined to the engine's stream with two events per launch point:
  // one after the other, the small group cost a wavefront lifetime of its own per kernel class (configs[4]: 5 % of a sweep)
  hipStream_t stream_wide = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_stream = true;            // GPH_SIDE_STREAM=0: the groups one after the other on the engine's stream (A/B, tests)
  struct Tm { hipEvent_t a, b; int which; };
  std::vector<Tm> tm;                // event pairs of the launches since the last synchronisation
  size_t tm_used = 0;
#else
  std::vector<char> lds;
#endif
};

static int align_up(int x, int a) { return (x + a - 1) / a * a; }

// page + LDS layout for (n, K, B): everything sized by the ACTUAL dimensions, not by the
// reference's MAX_* caps (sizeof(Locus_SuperStruct) alone is 9.7 KB there)
static void build_layout(GphLayout &y, int n, int Kc, int K, int B, int rootPop, int Pmax, int cnt16)
{
  memset(&y, 0, sizeof y);
  y.n = n; y.N = 2 * n - 1; y.K = K; y.Kc = Kc; y.B = B; y.rootPop = rootPop;
  y.E = 2 * n + 4 * GPH_MAX_MIGS + 3 * B + K + 10;   // event pool, patch.c:92
  y.RB = GPH_MAX_MIGS + 2 * B;
  // the HBM page IS the page part of the static LDS image (GphLds, capacity-sized arrays):
  // staging is one fully coalesced 16-B/lane copy
#define OFS(f) ((int32_t)offsetof(GphLds, f))
  y.o_ev = OFS(ev); y.o_nd = OFS(nd); y.o_sv = OFS(sv); y.o_mig_age = OFS(mig_age);
  y.o_coal = OFS(coal); y.o_migst = OFS(migst); y.o_rb_age = OFS(rb_age); y.o_fscal = OFS(fscal);
  y.o_iscal = OFS(iscal);
  y.o_nev = OFS(nev);
  y.o_first = OFS(first);
  y.o_mig_i = OFS(mig_i); y.o_living = OFS(living); y.o_ncoal = OFS(ncoal); y.o_nmig = OFS(nmig); y.o_rb_i = OFS(rb_i);
  y.page_bytes = align_up(OFS(s_dcoal), 16);
#undef OFS
  // dynamic LDS: the locus' sequence block (same bytes as its HBM block, laid out by its own P: GPH_Q_* in
  // gph_types.h) + per-pattern terms of the root reduction for loci with more than one pattern per lane
  y.Pmax = Pmax;
  y.cnt16 = cnt16;
  y.dyn_bytes = 0;
  y.lds_bytes = GPH_Q_TERMS(Pmax, n, cnt16) + (Pmax > GPH_WAVE ? 8 * Pmax : 8 * GPH_WAVE);
  y.huge_P = 0x7fffffff;
}

static void build_model_static(gph_engine *e)
{
  GphModel &m = e->G_h->model;
  const gph_config &c = e->cfg;
  memset(&m, 0, sizeof m);
  for (int p = 0; p < c.K; p++) {
    m.popFather[p] = e->popFather[p];
    m.popSon0[p] = e->popSon0[p];
    m.popSon1[p] = e->popSon1[p];
    gph_popmask mask = 0;
    for (int d = 0; d < c.K; d++) {           // isAncestralTo, self-inclusive (MCMCcontrol.c:851,977,1015-1024)
      int x = d;
      while (x >= 0) { if (x == p) { mask |= (gph_popmask)1 << d; break; } x = e->popFather[x]; }
    }
    m.isAnc[p] = mask;
  }
  int cum = 0;
  for (int p = 0; p < c.Kc; p++) { m.samplesPerPop[p] = e->samplesPerPop[p]; cum += e->samplesPerPop[p]; m.cumSamples[p] = cum; }
  for (int b = 0; b < c.B; b++) { m.bandSrc[b] = e->bandSrc[b]; m.bandTgt[b] = e->bandTgt[b]; m.bandsInto[e->bandTgt[b]][b >> 5] |= 1u << (b & 31); }
  for (int p = 0; p < c.K; p++)
    for (int b = 0; b < c.B; b++) if ((m.isAnc[e->bandTgt[b]] >> p) & 1) m.bandsOver[p][b >> 5] |= 1u << (b & 31);
  // populationPostOrder(rootPop), patch.c:1936-1951
  std::vector<int> order;
  struct Rec { static void go(gph_engine *e, int pop, std::vector<int> &o) {
    if (pop >= e->cfg.Kc) { go(e, e->popSon0[pop], o); go(e, e->popSon1[pop], o); }
    o.push_back(pop); } };
  Rec::go(e, c.rootPop, order);
  for (size_t i = 0; i < order.size(); i++) m.postOrder[i] = order[i];
  GphGlobal &G = *e->G_h;
  G.n = c.n; G.K = c.K; G.Kc = c.Kc; G.B = c.B; G.rootPop = c.rootPop; G.Ltot = (double)c.L_total;
  G.tau_limit = (long long)1 << 62;
}

#ifdef GPH_WALKSTAT
long long gph_ws[4];
struct GphWsPrint { ~GphWsPrint() { fprintf(stderr, "WALKSTAT proposals %lld: prune steps %.2f, regraft steps %.2f, common prefix %.2f per proposal\n", gph_ws[3], (double)gph_ws[0] / gph_ws[3], (double)gph_ws[1] / gph_ws[3], (double)gph_ws[2] / gph_ws[3]); } } gph_ws_print;
#endif
// ---------------------------------------------------------------- runtime shim
// the model in the kernel-argument segment (every build but the many-band one, whose kernels all read G->model)
#if GPH_BIG_BANDS
#define GPH_KA_MODEL(ka, e) ((void)0)
#else
#define GPH_KA_MODEL(ka, e) ((ka).model = (e)->G_h->model)
#endif
#ifdef GPH_HOSTEMU
static int dev_alloc(void **p, size_t bytes) { *p = calloc(1, bytes ? bytes : 1); return *p ? 0 : GPH_EHIP; }
static void dev_free(void *p) { free(p); }
static int h2d(gph_engine *, void *d, const void *h, size_t n) { memcpy(d, h, n); return 0; }
static int d2h(gph_engine *, void *h, const void *d, size_t n) { memcpy(h, d, n); return 0; }
static int stream_sync(gph_engine *e) { e->n_syncs++; return 0; }
#define LAUNCH_PRE(e) do { g_model = (e)->G_h->model; g_lay = (e)->lay; gph_G_emu = (e)->G_h; \
    GphKargs &ka_ = (e)->ka; GPH_KA_MODEL(ka_, e); ka_.lay = (e)->lay; ka_.G = (e)->G_h; } while (0)
#define LAUNCH(e, which, name, ...) do { LAUNCH_PRE(e); GphKargs &ka_ = (e)->ka; \
    for (auto &bk_ : (e)->buckets) { (e)->lds.assign(bk_.lds_bytes + 8 * (e)->lay.Pmax + 64, 0); /* the host form keeps per-pattern terms for every P */ gph_sm = (e)->lds.data(); \
      for (int b_ = 0; b_ < bk_.count; b_++) name(b_, ka_, (e)->dev, bk_.j0, __VA_ARGS__); } \
    (e)->last_which = (which); (e)->cls_launches[which] += 1; (e)->n_launches++; } while (0)
#else
static int dev_alloc(void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16) == hipSuccess ? 0 : GPH_EHIP; }
static void dev_free(void *p) { if (p) (void)hipFree(p); }
static int collect_times(gph_engine *e);
static int flush_pending(gph_engine *e);
static int stream_sync(gph_engine *e)
{
  HIPCHK(hipStreamSynchronize(e->stream));
  e->n_syncs++;
  return collect_times(e);
}
static int h2d(gph_engine *e, void *d, const void *h, size_t n)
{
  HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
static int d2h(gph_engine *e, void *h, const void *d, size_t n)
{
  HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
// elapsed times of the launches bracketed since the last synchronisation (the stream is idle: every event is complete)
static int collect_times(gph_engine *e)
{
  for (size_t i = 0; i < e->tm_used; i++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->tm[i].a, e->tm[i].b) == hipSuccess) { e->last_ms[e->tm[i].which] = ms; e->cls_ms[e->tm[i].which] += ms; }
  }
  e->tm_used = 0;
  return 0;
}
static int tm_begin(gph_engine *e, int which)
{
  if (!((e->timing_mask >> which) & 1)) return -1;
  if (e->tm_used == e->tm.size()) {
    gph_engine::Tm t;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return -1;
    e->tm.push_back(t);
  }
  e->tm[e->tm_used].which = which;
  (void)hipEventRecord(e->tm[e->tm_used].a, e->stream);
  return (int)e->tm_used++;
}
static void tm_end(gph_engine *e, int slot) { if (slot >= 0) (void)hipEventRecord(e->tm[slot].b, e->stream); }
// timed launch: HIP events on the engine's own stream bracket the kernel.  One dispatch covers every
// locus with at most one pattern per lane (slots in decreasing P: longest wavefronts first), a second one
// the rare loci with more (they also need the per-pattern terms array in LDS).  No host synchronisation here.
#define LAUNCH_PRE(e) do { GphKargs &ka_ = (e)->ka; GPH_KA_MODEL(ka_, e); ka_.lay = (e)->lay; ka_.G = (e)->G_d; } while (0)
#define LAUNCH(e, which, name, ...) do { { int rcf_ = flush_pending(e); if (rcf_) return rcf_; } LAUNCH_PRE(e); GphKargs &ka_ = (e)->ka; \
    const int tms_ = tm_begin((e), (which)); \
    const bool fork_ = (e)->side_stream && (e)->buckets.size() >= 2; \
    if (fork_) { HIPCHK(hipEventRecord((e)->ev_fork, (e)->stream)); HIPCHK(hipStreamWaitEvent((e)->stream_wide, (e)->ev_fork, 0)); } \
    for (size_t bi_ = 0; bi_ < (e)->buckets.size(); bi_++) { auto &bk_ = (e)->buckets[bi_]; \
      ka_.lay.dyn_bytes = bk_.lds_bytes; \
      hipLaunchKernelGGL(name, dim3((unsigned)bk_.count), dim3(GPH_WAVE), bk_.lds_bytes + (e)->lds_pad[which], fork_ && bi_ + 1 < (e)->buckets.size() ? (e)->stream_wide : (e)->stream, ka_, (e)->dev, bk_.j0, __VA_ARGS__); \
      HIPCHK(hipGetLastError()); (e)->n_launches++; } \
    if (fork_) { HIPCHK(hipEventRecord((e)->ev_join, (e)->stream_wide)); HIPCHK(hipStreamWaitEvent((e)->stream, (e)->ev_join, 0)); } \
    tm_end((e), tms_); \
    (e)->last_which = (which); (e)->cls_launches[which] += 1; } while (0)
#endif

// one single-wave workgroup with its own dynamic-LDS size (the serial scan of UpdateLocusRate)
#ifdef GPH_HOSTEMU
#define LAUNCH1(e, which, name, ldsbytes, ...) do { LAUNCH_PRE(e); GphKargs &ka_ = (e)->ka; \
    (e)->lds.assign((size_t)(ldsbytes) + 64, 0); gph_sm = (e)->lds.data(); \
    name(0, ka_, (e)->dev, 0, __VA_ARGS__); \
    (e)->last_which = (which); (e)->cls_launches[which] += 1; } while (0)
#else
#define LAUNCH1(e, which, name, ldsbytes, ...) do { { int rcf_ = flush_pending(e); if (rcf_) return rcf_; } LAUNCH_PRE(e); GphKargs &ka_ = (e)->ka; \
    const int tms_ = tm_begin((e), (which)); \
    hipLaunchKernelGGL(name, dim3(1), dim3(GPH_WAVE), (ldsbytes), (e)->stream, ka_, (e)->dev, 0, __VA_ARGS__); \
    HIPCHK(hipGetLastError()); \
    tm_end((e), tms_); \
    (e)->last_which = (which); (e)->cls_launches[which] += 1; } while (0)
#endif

// host mirror -> HBM (the host changed settings or, in host mode, took a decision); a synchronous copy: the mirror
// may be modified again as soon as this returns
static int push_G(gph_engine *e)
{
  e->G_dirty = false;
#ifdef GPH_HOSTEMU
  return 0;
#else
  HIPCHK(hipMemcpyAsync(e->G_d, e->G_h, sizeof(GphGlobal), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
#endif
}
// HBM -> host mirror, with a host synchronisation (the end of an iteration in resident mode)
static int pull_G(gph_engine *e)
{
#ifdef GPH_HOSTEMU
  e->n_syncs++;
  return 0;
#else
  HIPCHK(hipMemcpyAsync(e->G_h, e->G_d, sizeof(GphGlobal), hipMemcpyDeviceToHost, e->stream));
  { int rcs = stream_sync(e); if (rcs) return rcs; }
  e->mirror_current = true;
  return 0;
#endif
}
#define PUSH_IF_DIRTY(e) do { if ((e)->G_dirty) { int rcp_ = push_G(e); if (rcp_) return rcp_; } } while (0)

// are the decisions above the loci taken on the device?  Yes unless something forces a host synchronisation at every
// reduction point anyway: a caller-supplied all-reduce hook, the host shared-memory exchange of ranks that share a
// GPU, the host-emulation build, or GPH_HOST_DECISIONS=1 (tests compare the two modes)
static bool resident(const gph_engine *e)
{
#ifdef GPH_HOSTEMU
  (void)e;
  return false;
#else
  if (e->force_host || e->allreduce || e->var_rates) return false;
  if (e->comm && !gph_comm_on_stream(e->comm)) return false;
  return true;
#endif
}
static int world_of(const gph_engine *e) { return e->comm ? gph_comm_world(e->comm) : 1; }

#ifdef GPH_HOSTEMU
static int flush_pending(gph_engine *) { return 0; }
#else
// resident mode: everything requested since the last per-locus launch -- the reduction(s) of its outputs and the stages
// above the loci -- as one kernel (single rank), or reduction -> RCCL all-gather -> stages (several ranks)
static int flush_pending(gph_engine *e)
{
  const bool red = e->pend_nc0 > 0 || e->pend_nc1 > 0;
  if (!red && e->pend_sl.n == 0) return 0;
  const GphStageList sl = e->pend_sl;
  const int nc0 = e->pend_nc0, nc1 = e->pend_nc1;
  e->pend_nc0 = e->pend_nc1 = 0;
  e->pend_sl.n = 0;
  const bool multi = e->comm && gph_comm_world(e->comm) > 1;
  LAUNCH_PRE(e);
  if (red) {
    /* several ranks whose kernels can address each other's rows (gph_comm_peer_exchange): the exchange happens INSIDE the
     * reduction kernel's last block and the stages run there on everybody's rows -- one launch per reduction point, as
     * with a single rank */
    const bool inkernel = multi && e->peer_rows != nullptr;
    const int fuse = sl.n > 0 && (!multi || inkernel);
    GphPeerX X{1, 0, 0, 0, nullptr, nullptr, nullptr, 0};
    if (inkernel) X = GphPeerX{gph_comm_world(e->comm), gph_comm_rank(e->comm), e->peer_stride, 0, e->peer_rows, e->peer_flags, e->d_gather, gph_comm_peer_next_gen(e->comm)};
    hipLaunchKernelGGL(k_reduce_stage, dim3(GPH_RED_BLOCKS), dim3(GPH_RED_THREADS), 0, e->stream, e->ka, e->dev, nc0, nc1, e->d_part, e->d_ticket,
                       e->d_red, fuse, sl, e->pend_iteration, X);
    HIPCHK(hipGetLastError());
    e->n_launches++;
    if (inkernel) {
      e->n_collectives++;
    } else if (multi) {
      if (gph_comm_allgather_stream(e->comm, e->d_red, e->d_gather, GPH_RED_ROW, (void *)e->stream)) return GPH_EHIP;
      e->n_collectives++;
    } else if (e->comm) {
      e->n_collectives++;   /* a one-rank communicator: the row is already everybody's */
      if (gph_comm_allgather_stream(e->comm, e->d_red, e->d_gather, GPH_RED_ROW, (void *)e->stream)) return GPH_EHIP;
    }
    if (fuse) return 0;
  }
  if (sl.n > 0) {
    const double *rows = e->comm ? e->d_gather : e->d_red;
    hipLaunchKernelGGL(k_global, dim3(1), dim3(GPH_RED_THREADS), 0, e->stream, e->ka, rows, world_of(e), sl, e->pend_iteration);
    HIPCHK(hipGetLastError());
    e->n_launches++;
  }
  return 0;
}
#endif

// reduce the per-locus outputs (section 0) or the page statistics (section 1) over the local loci into this rank's row
static int reduce_local(gph_engine *e, int sec, int ncols)
{
#ifdef GPH_HOSTEMU
  double *red = e->d_red + sec * GPH_RED_STRIDE;
  for (int c = 0; c < ncols; c++) {
    double s = 0, mn = 1e300, mx = -1e300;
    for (int64_t g = 0; g < e->L; g++) {
      double v;
      if (sec == 0) v = e->dev.out[(size_t)g * GPH_OUT_SLOTS + c];
      else v = e->dev.stats[(size_t)g * ncols + c];
      s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
    }
    red[c] = s; red[GPH_RED_COLS + c] = mn; red[2 * GPH_RED_COLS + c] = mx;
  }
  red[3 * GPH_RED_COLS] = (double)*e->dev.err;
  return 0;
#else
  if (resident(e)) {
    /* lazily: fused with the stages that consume the row (flush_pending) */
    if (e->pend_sl.n > 0) { int rcf = flush_pending(e); if (rcf) return rcf; }
    if (sec == 0) e->pend_nc0 = ncols; else e->pend_nc1 = ncols;
    return 0;
  }
  { int rcf = flush_pending(e); if (rcf) return rcf; }
  GphStageList none = {0, {0}, {0}};
  LAUNCH_PRE(e);
  hipLaunchKernelGGL(k_reduce_stage, dim3(GPH_RED_BLOCKS), dim3(GPH_RED_THREADS), 0, e->stream, e->ka, e->dev, sec == 0 ? ncols : 0, sec == 1 ? ncols : 0,
                     e->d_part, e->d_ticket, e->d_red, 0, none, 0, GphPeerX{1, 0, 0, 0, nullptr, nullptr, nullptr, 0});
  HIPCHK(hipGetLastError());
  e->n_launches += 1;
  return 0;
#endif
}
static int reduce_stats(gph_engine *e) { return reduce_local(e, 1, 2 * e->cfg.K + 2 * e->cfg.B); }

// which columns of the reduced row a stage consumes (host mode with a caller-supplied all-reduce hook: only these
// travel): sums / minima / maxima as (section, column) pairs; every stage that follows a locus kernel also takes the
// counters and the error words
struct StageCols { int ns = 0, nm = 0, nx = 0; int s[GPH_RED_COLS + 32][2], m[8][2], x[8][2]; };
static void stage_columns(const gph_engine *e, int stage, StageCols &c)
{
  const int C = 2 * e->cfg.K + 2 * e->cfg.B;
  auto S = [&](int sec, int col) { c.s[c.ns][0] = sec; c.s[c.ns][1] = col; c.ns++; };
  auto M = [&](int sec, int col) { c.m[c.nm][0] = sec; c.m[c.nm][1] = col; c.nm++; };
  auto X = [&](int sec, int col) { c.x[c.nx][0] = sec; c.x[c.nx][1] = col; c.nx++; };
  auto counters = [&]() { S(0, 8); S(0, 9); S(0, 10); S(0, 13); X(0, 11); X(0, -1); };
  auto totals = [&]() { for (int k = 0; k < C; k++) S(1, k); };
  switch (stage) {
  case GS_INIT_DONE: counters(); S(0, 0); S(0, 1); totals(); break;
  case GS_SWEEP_DONE: counters(); for (int k = 0; k < 8; k++) S(0, k); S(0, 12); M(0, 15); totals(); break;
  case GS_TOTALS: totals(); break;
  case GS_TAU_DECIDE: case GS_SAGE_DECIDE: counters(); S(0, 0); S(0, 1); S(0, 3); S(0, 4); M(0, 14); break;
  case GS_MIX_DECIDE: counters(); S(0, 0); break;
  case GS_REFRESH_DONE: counters(); M(0, 0); S(0, 1); S(0, 2); break;
  case GS_CHECK_DONE: counters(); M(0, 0); S(0, 1); S(0, 2); totals(); break;
  case GS_COUNT_ONLY: counters(); S(0, 0); S(0, 1); S(0, 3); S(0, 4); M(0, 14); break;   /* + what the stepwise evaluate calls return */
  default: break;
  }
}
static bool stage_reads_row(int stage)
{
  switch (stage) {
  case GS_INIT_DONE: case GS_SWEEP_DONE: case GS_TOTALS: case GS_TAU_DECIDE: case GS_SAGE_DECIDE: case GS_MIX_DECIDE:
  case GS_REFRESH_DONE: case GS_CHECK_DONE: case GS_COUNT_ONLY: return true;
  default: return false;
  }
}
static double *colp(double *row, const int sc[2], int kind)   /* kind 0 sum, 1 min, 2 max; column -1 = the error word */
{
  if (sc[1] < 0) return row + 3 * GPH_RED_COLS;
  return row + sc[0] * GPH_RED_STRIDE + kind * GPH_RED_COLS + sc[1];
}

// one stage above the loci (gph_global.h: gg_stage) on the reduced row(s) of the launch(es) before it.
// resident mode: [all-gather on the stream] + k_global, no host synchronisation.
// host mode: synchronise, fetch this rank's row, combine it over the ranks (hook or shared-memory exchange), run
// gg_stage on the host mirror and push the mirror.
static int run_stage(gph_engine *e, int stage, int arg, int iteration)
{
  const bool reads = stage_reads_row(stage);
#ifndef GPH_HOSTEMU
  if (resident(e)) {
    /* queued: goes out fused with the reduction before it (and the stages next to it) at the next launch */
    if (e->pend_sl.n == 6) { int rcf = flush_pending(e); if (rcf) return rcf; }
    e->pend_sl.stage[e->pend_sl.n] = stage;
    e->pend_sl.arg[e->pend_sl.n] = arg;
    e->pend_sl.n++;
    e->pend_iteration = iteration;
    e->mirror_current = false;     /* the stage writes the chain state on the device: the mirror is stale until pull_G */
    (void)reads;
    return 0;
  }
  { int rcf = flush_pending(e); if (rcf) return rcf; }
#endif
  GphRed R;
  R.rows = e->h_red; R.world = 1;
  if (reads) {
#ifdef GPH_HOSTEMU
    memcpy(e->h_red, e->d_red, sizeof(double) * GPH_RED_ROW);
    e->n_syncs++;
#else
    HIPCHK(hipMemcpyAsync(e->h_red, e->d_red, sizeof(double) * GPH_RED_ROW, hipMemcpyDeviceToHost, e->stream));
    { int rcs = stream_sync(e); if (rcs) return rcs; }
#endif
    if (e->allreduce || e->comm) {
      StageCols c;
      stage_columns(e, stage, c);
      double sums[GPH_RED_COLS + 32], mins[16];
      for (int k = 0; k < c.ns; k++) sums[k] = *colp(e->h_red, c.s[k], 0);
      for (int k = 0; k < c.nm; k++) mins[k] = *colp(e->h_red, c.m[k], 1);
      for (int k = 0; k < c.nx; k++) mins[c.nm + k] = -*colp(e->h_red, c.x[k], 2);   /* a maximum as the minimum of the negatives */
      int rc = 0;
      if (e->allreduce) rc = e->allreduce(e->allreduce_user, sums, c.ns, mins, c.nm + c.nx);
      else rc = gph_comm_allreduce_host(e->comm, sums, c.ns, mins, c.nm + c.nx);
      if (rc) return GPH_EHIP;
      e->n_collectives++;
      for (int k = 0; k < c.ns; k++) *colp(e->h_red, c.s[k], 0) = sums[k];
      for (int k = 0; k < c.nm; k++) *colp(e->h_red, c.m[k], 1) = mins[k];
      for (int k = 0; k < c.nx; k++) *colp(e->h_red, c.x[k], 2) = -mins[c.nm + k];
    }
  }
  e->G_h->iteration = iteration;
  gg_stage(*e->G_h, R, stage, arg);
  return push_G(e);
}

// one locus in the canonical text form (same format as oracle/gphocs_oracle_io.c go_dump_state's per-locus part): pg = its
// page, cb = its conditionals (or null)
static void dump_one_locus(const gph_engine *e, FILE *f, long long global_locus, const char *pg, const char *cb, int P)
{
  const GphLayout &y = e->lay;
  const double *fs = (const double *)(pg + y.o_fscal);
  const int32_t *is = (const int32_t *)(pg + y.o_iscal);
  const GphNode *nd = (const GphNode *)(pg + y.o_nd);
  const int16_t *ne = (const int16_t *)(pg + y.o_nev);
  const int16_t *first = (const int16_t *)(pg + y.o_first);
  const GphEv *evr = (const GphEv *)(pg + y.o_ev);
  fprintf(f, "LOCUS %lld root %d dataLnL %a genLnL %a rng %u %u %u\n", (long long)global_locus, is[IS_ROOT],
          fs[FS_DATALNL], fs[FS_GENLNL], (unsigned)is[IS_RX], (unsigned)is[IS_RY], (unsigned)is[IS_RZ]);
  if (e->var_rates) fprintf(f, "R %a\n", fs[FS_MUTRATE]);
  for (int i = 0; i < y.N; i++)
    fprintf(f, "N %d %d %d %d %a %d %d\n", i, nd[i].father, nd[i].left, nd[i].right, nd[i].age, nd[i].npop, i < y.n ? -1 : ne[i]);
  for (int pop = 0; pop < y.K; pop++) {
    fprintf(f, "C %d", pop);
    int guard = 0;
    for (int ev = first[pop]; ev >= 0 && guard++ < y.E; ev = evr[ev].next)
      fprintf(f, " %d:%d:%d:%d:%a", ev, evr[ev].type, evr[ev].node, evr[ev].nlin, evr[ev].time);
    fprintf(f, "\n");
  }
  fprintf(f, "S");
  for (int pop = 0; pop < y.K; pop++)
    fprintf(f, " %a %d", ((const double *)(pg + y.o_coal))[pop], ((const int16_t *)(pg + y.o_ncoal))[pop]);
  for (int b = 0; b < y.B; b++)
    fprintf(f, " %a %d", ((const double *)(pg + y.o_migst))[b], ((const int16_t *)(pg + y.o_nmig))[b]);
  fprintf(f, "\n");
  fprintf(f, "M %d", is[IS_NUM_MIGS]);
  for (int i = 0; i < is[IS_NUM_MIGS]; i++) {
    int mg = ((const int16_t *)(pg + y.o_living))[i];
    const int16_t *mi = (const int16_t *)(pg + y.o_mig_i) + mg * MG_COUNT;
    fprintf(f, " %d:%d:%d:%d:%d:%d:%d:%a", mg, mi[MG_BRANCH], mi[MG_BAND], mi[MG_SPOP], mi[MG_TPOP], mi[MG_SEV],
            mi[MG_TEV], ((const double *)(pg + y.o_mig_age))[mg]);
  }
  fprintf(f, "\n");
  if (cb) {
    for (int i = y.n; i < y.N; i++) {
      const double *c = (const double *)(cb + ((size_t)((int)(((uint32_t)is[IS_CBIT0 + (i >> 5)] >> (i & 31)) & 1) * (y.n - 1) + (i - y.n)) * P) * 32);
      fprintf(f, "K %d", i);
      for (int k = 0; k < 4 * P; k++) fprintf(f, " %a", c[k]);
      fprintf(f, "\n");
    }
  }
}

// after a host synchronisation: the error the stages recorded, the counters
// hipFree synchronises the device.  With the in-kernel exchange another rank's reduction kernel may be spinning for this rank's
// next row: a free between two reduction points would then wait for a kernel that waits for us.  Such blocks are kept until the
// engine is destroyed.
static void eng_free(gph_engine *e, void *p)
{
  if (!p) return;
  if (e->peer_rows) e->deferred_free.push_back(p);
  else dev_free(p);
}

static int check_error(gph_engine *e)
{
  const int code = e->G_h->error;
  if (code != 0) {
    const long long gl = e->G_h->error_locus;      /* -1: not a per-locus error (every gg_* path that raises one says so) */
    e->last_error_code = code;
    e->last_error_locus = gl;
    if (code == 75) fprintf(stderr, "gphocs_hip: synchronizeEvents found an inconsistency (Fatal Error 0075/0076)\n");
    else if (code == 9999) fprintf(stderr, "gphocs_hip: checkAll failed at iteration %d\n", e->G_h->iteration);
    else if (gl >= 0) fprintf(stderr, "gphocs_hip: Fatal Error %04d reported by a locus kernel, first in locus %lld (iteration %d)\n", code, gl, e->G_h->iteration);
    else fprintf(stderr, "gphocs_hip: Fatal Error %04d reported by a locus kernel\n", code);
    /* what printGenealogyAndExit prints upstream (GPhoCS.c:660-676: the locus's genealogy and its event chains), in the
     * canonical dump format, when the failing locus is one of this rank's; the state is the one the kernel left behind */
    const long long lo = gl - e->cfg.locus_begin;
    if (gl >= 0 && lo >= 0 && lo < e->L && !e->in_error_dump) {
      e->in_error_dump = true;
      int64_t slot = -1;
      for (int64_t j = 0; j < e->L; j++) if (e->h_orig[j] == lo) { slot = j; break; }
      std::vector<char> pg(e->lay.page_bytes);
      if (slot >= 0 && !d2h(e, pg.data(), (const char *)e->dev.pages + (size_t)slot * e->lay.page_bytes, e->lay.page_bytes)) {
        fprintf(stderr, "gphocs_hip: genealogy and event chains of locus %lld:\n", gl);
        dump_one_locus(e, stderr, gl, pg.data(), nullptr, 0);
      }
      e->in_error_dump = false;
    }
    return GPH_EKERNEL;
  }
  return 0;
}
// the mirror is current (host mode: always; resident mode: after pull_G)
static int finish_sync(gph_engine *e)
{
  { int rcf = flush_pending(e); if (rcf) return rcf; }
  if (resident(e)) { int rc = pull_G(e); if (rc) return rc; }
  return check_error(e);
}

static int run_stage_now(gph_engine *e, int stage, int arg);
// run a deferred synchronizeEvents pass now (anything but the next genealogy sweep is about to touch the pages).
// now = the caller is a stepwise entry point that goes on to edit the host mirror of the chain state and push it: the
// stage that checks the pass (Fatal Error 0075/0076, the class-8 counters) must then have run -- on the host, result
// checked -- before that, or the push would overwrite what a queued device-side stage wrote
// the commit of a decided mixing proposal that was left for the next sweep kernel: as a kernel of its own, now (something
// other than the sweep is about to read or write the pages)
static int mix_finish_owed(gph_engine *e)
{
  if (e->mix_owed) { e->mix_owed = false; LAUNCH(e, 7, k_mix_finish, 0); }
  return 0;
}
static int flush_sync(gph_engine *e, bool now)
{
  { int rcm = mix_finish_owed(e); if (rcm) return rcm; }     /* a mixing commit left for the next sweep kernel goes first */
  if (!e->sync_pending) return 0;
  e->sync_pending = false;
  PUSH_IF_DIRTY(e);
  LAUNCH(e, 8, k_sync, 0);
  int rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (rc) return rc;
  if (now) return run_stage_now(e, GS_REFRESH_DONE, 0);
  return run_stage(e, GS_REFRESH_DONE, 0, e->G_h->iteration);
}

// exp/log/rndu constants of gph_math.h / gph_locus.h (GphKargs::mathc) and, on the device, the addresses of the two
// 128-entry libm tables
static void fill_math_constants(GphKargs &ka)
{
  const double ec[8] = GPH_EXP_CONSTS, lc[18] = GPH_LOG_CONSTS;
  const double rc[6] = {1.0 / 30269.0, 1.0 / 30307.0, 1.0 / 30323.0, 30269.0, 30307.0, 30323.0};
  memcpy(&ka.mathc[0], ec, sizeof ec);
  memcpy(&ka.mathc[8], lc, sizeof lc);
  memcpy(&ka.mathc[26], rc, sizeof rc);
}
#ifndef GPH_HOSTEMU
static int fill_math_tables(GphKargs &ka)
{
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(gph_log_t_d)) != hipSuccess) return 1;
  ka.log_t = (const double *)p;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(gph_exp_t_d)) != hipSuccess) return 1;
  ka.exp_t = (const uint64_t *)p;
  return 0;
}
#define SETDEV(e) do { if (hipSetDevice((e)->cfg.device) != hipSuccess) return GPH_EHIP; } while (0)
#else
#define SETDEV(e) ((void)0)
#endif

// a stage whose result the CALLER needs now (the stepwise entry points of the C ABI): host path whatever the mode
static int run_stage_now(gph_engine *e, int stage, int arg)
{
  const bool fh = e->force_host;
  e->force_host = true;
  int rc = run_stage(e, stage, arg, e->G_h->iteration);
  e->force_host = fh;
  if (rc) return rc;
  return check_error(e);
}

// ---------------------------------------------------------------- C ABI
extern "C" {

int gph_engine_create(const gph_config *cfg, gph_engine **out)
{
  if (!cfg || !out) return GPH_EARG;
  if (cfg->n > 200 || cfg->K > 39 || cfg->B > 100)
    fprintf(stderr, "gphocs_hip: n=%d K=%d B=%d exceed even the reference's compile-time caps (NS 200, 2*NSPECIES-1 = 39, MAX_MIG_BANDS 100: upstream src/patch.h:17-22)\n", cfg->n, cfg->K, cfg->B);
  if (cfg->n < 2 || cfg->n > GPH_CAP_LEAVES || cfg->K > GPH_CAP_K || cfg->B > GPH_CAP_B || cfg->K != 2 * cfg->Kc - 1) {
    fprintf(stderr, "gphocs_hip: unsupported dimensions n=%d K=%d B=%d (this library variant: n<=%d leaves, K<=%d populations, B<=%d bands; the largest variant covers the reference's own caps 200 / 39 / 100, upstream src/patch.h:17-22)\n",
            cfg->n, cfg->K, cfg->B, GPH_CAP_LEAVES, GPH_CAP_K, GPH_CAP_B);
    return GPH_EARG;
  }
  gph_engine *e = new gph_engine();
  e->cfg = *cfg;
  e->samplesPerPop.assign(cfg->samplesPerPop, cfg->samplesPerPop + cfg->Kc);
  e->popFather.assign(cfg->popFather, cfg->popFather + cfg->K);
  e->popSon0.assign(cfg->popSon0, cfg->popSon0 + cfg->K);
  e->popSon1.assign(cfg->popSon1, cfg->popSon1 + cfg->K);
  if (cfg->B > 0) { e->bandSrc.assign(cfg->bandSrc, cfg->bandSrc + cfg->B); e->bandTgt.assign(cfg->bandTgt, cfg->bandTgt + cfg->B); }
  e->cfg.samplesPerPop = e->samplesPerPop.data();
  e->cfg.popFather = e->popFather.data();
  e->cfg.popSon0 = e->popSon0.data();
  e->cfg.popSon1 = e->popSon1.data();
  e->cfg.bandSrc = e->bandSrc.data();
  e->cfg.bandTgt = e->bandTgt.data();
  memset(&e->dev, 0, sizeof e->dev);
  memset(&e->ka, 0, sizeof e->ka);
  fill_math_constants(e->ka);
  if (const char *fh = getenv("GPH_HOST_DECISIONS")) e->force_host = atoi(fh) != 0;
  if (const char *nf = getenv("GPH_NO_FUSE")) e->no_fuse = atoi(nf) != 0;
#ifndef GPH_HOSTEMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) {
    fprintf(stderr, "gphocs_hip: no usable HIP device %d (found %d) -- this library has no CPU path\n", cfg->device, ndev);
    delete e;
    return GPH_EHIP;
  }
  if (hipSetDevice(cfg->device) != hipSuccess || fill_math_tables(e->ka)) {
    fprintf(stderr, "gphocs_hip: cannot resolve the device math tables\n");
    delete e;
    return GPH_EHIP;
  }
  if (const char *ov = getenv("GPH_SIDE_STREAM")) e->side_stream = atoi(ov) != 0;
  /* GPH_NONBLOCKING_STREAM=1 (tests): the engine's stream does not order itself against the legacy null stream -- what the
   * in-kernel exchange of thread ranks needs (gph_engine_set_comm) */
  const bool nb_ = getenv("GPH_NONBLOCKING_STREAM") && atoi(getenv("GPH_NONBLOCKING_STREAM")) != 0;
  if ((nb_ ? hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) : hipStreamCreate(&e->stream)) != hipSuccess ||
      hipStreamCreateWithFlags(&e->stream_wide, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void **)&e->G_h, sizeof(GphGlobal), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void **)&e->G_d, sizeof(GphGlobal)) != hipSuccess ||
      hipHostMalloc((void **)&e->h_red, sizeof(double) * GPH_RED_ROW * 64, hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void **)&e->d_red, sizeof(double) * GPH_RED_ROW) != hipSuccess ||
      hipMemset(e->d_red, 0, sizeof(double) * GPH_RED_ROW) != hipSuccess ||
      hipMalloc((void **)&e->d_ticket, 64) != hipSuccess || hipMemset(e->d_ticket, 0, 64) != hipSuccess ||
      hipStreamSynchronize(nullptr) != hipSuccess) { gph_engine_destroy(e); return GPH_EHIP; }
  e->d_gather = e->d_red;
#else
  e->G_h = (GphGlobal *)calloc(1, sizeof(GphGlobal));
  e->G_d = e->G_h;
  e->h_red = (double *)calloc(GPH_RED_ROW, sizeof(double));
  e->d_red = (double *)calloc(GPH_RED_ROW, sizeof(double));
  e->d_gather = e->d_red;
#endif
  memset(e->G_h, 0, sizeof(GphGlobal));
  e->G_h->error_locus = -1;          /* (0 is a locus: an error that is not per-locus must not name it -- ADVICE round 5) */
  build_model_static(e);
  *out = e;
  return 0;
}

void gph_engine_destroy(gph_engine *e)
{
  if (!e) return;
  dev_free(e->dev.pages); dev_free(e->dev.shadow); dev_free(e->dev.cond); dev_free((void *)e->dev.cond_off);
  dev_free((void *)e->dev.seq); dev_free((void *)e->dev.seq_off); dev_free((void *)e->dev.orig); dev_free((void *)e->dev.P); dev_free(e->dev.out); dev_free(e->dev.stats); dev_free(e->d_mutRate);
  dev_free(e->d_lrec); dev_free(e->d_lpre); dev_free(e->d_slot_of); dev_free(e->d_lr_result); dev_free(e->d_lr_gscr); dev_free(e->d_ref_page); dev_free(e->d_ref_seq);
  dev_free(e->d_part); dev_free(e->dev.err);
  for (void *p : e->deferred_free) dev_free(p);
  e->deferred_free.clear();
  dev_free((void *)e->dev.slog_map); dev_free(e->dev.slog); dev_free(e->dev.slog_n);
  if (e->d_gather != e->d_red) dev_free(e->d_gather);
  dev_free(e->d_red);
#ifdef GPH_HOSTEMU
  free(e->h_red);
  free(e->G_h);
#else
  dev_free(e->G_d);
  dev_free(e->d_ticket);
  if (e->h_red) (void)hipHostFree(e->h_red);
  if (e->G_h) (void)hipHostFree(e->G_h);
  for (auto &t : e->tm) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->stream_wide) (void)hipStreamDestroy(e->stream_wide);
  if (e->stream) (void)hipStreamDestroy(e->stream);
#endif
  delete e;
}

int gph_engine_set_allreduce(gph_engine *e, gph_allreduce_fn fn, void *user)
{
  if (!e) return GPH_EARG;
  e->allreduce = fn;
  e->allreduce_user = user;
  return 0;
}

// native communicator (gph_comm.h): RCCL over xGMI, one rank per GPU, all-gathers queued on the engine's stream; or the
// host shared-memory exchange for ranks that share a GPU.  The engine does not own it.
int gph_engine_set_comm(gph_engine *e, gph_comm *c)
{
  if (!e) return GPH_EARG;
  SETDEV(e);
  e->comm = c;
  if (e->d_gather != e->d_red) { dev_free(e->d_gather); e->d_gather = e->d_red; }
  if (c && gph_comm_world(c) > 1 && gph_comm_on_stream(c)) {
    if (gph_comm_world(c) > 64) return GPH_EARG;
    if (dev_alloc((void **)&e->d_gather, sizeof(double) * GPH_RED_ROW * gph_comm_world(c))) return GPH_EHIP;
  }
  /* (the generation of the exchange belongs to the GROUP -- gph_comm_peer_next_gen: a second engine on the same communicator
   * goes on where the first one stopped instead of finding flags of generations it has not reached yet) */
  e->peer_rows = nullptr; e->peer_flags = nullptr; e->peer_stride = 0;
#ifndef GPH_HOSTEMU
  if (c && gph_comm_world(c) > 1 && gph_comm_on_stream(c)) {
    double *rows = nullptr; unsigned long long *flags = nullptr; int32_t stride = 0;
    int least_ = 0, greatest_ = 0;
    if (hipDeviceGetStreamPriorityRange(&least_, &greatest_) != hipSuccess) { least_ = greatest_ = 0; }
    const int levels_ = least_ - greatest_ + 1;
    if (gph_comm_peer_exchange(c, &rows, &flags, &stride) && stride >= GPH_RED_ROW && gph_comm_world(c) > levels_) {
      /* more thread ranks than stream priority levels: two ranks would share a hardware-queue pool and could end up behind
       * each other (tools/peer_exchange_stress.py: world 4 gave up 4 times in 10) -- the event-ordered gather instead */
      if (gph_comm_rank(c) == 0)
        fprintf(stderr, "gphocs_hip: the in-kernel exchange is offered for up to %d thread ranks on one device (one stream priority level "
                        "each); %d ranks use the event-ordered gather\n", levels_, gph_comm_world(c));
    } else if (gph_comm_peer_exchange(c, &rows, &flags, &stride) && stride >= GPH_RED_ROW) {
      e->peer_rows = rows; e->peer_flags = flags; e->peer_stride = stride;
      /* a reduction kernel of this engine may WAIT for another rank's kernel: the engine's stream must not be a blocking
       * stream then -- a blocking stream orders itself against the legacy null stream, and a null-stream operation of the
       * other rank's host thread (hipMemset at load time) would wait for this rank's waiting kernel: a deadlock until the
       * bounded wait gives up (found with tools/probe/peer_dbg.py; the plain two-stream probe, peer_probe.cpp, runs) */
      /* ... and the ranks' streams must sit on DIFFERENT hardware queues (a kernel that waits for a kernel queued behind it on
       * the same queue waits for ever).  HIP keeps a pool of hardware queues PER PRIORITY level and hands streams of one level
       * to its pool round-robin: streams of different priority levels never share a queue, so rank r takes level r mod levels
       * -- by construction for up to `levels` thread ranks (3 on this runtime: low / normal / high), round-robin luck beyond */
      hipStream_t ns = nullptr;
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
      const int levels = least - greatest + 1;
      const int prio = levels > 1 ? least - (gph_comm_rank(c) % levels) : 0;
      if (hipStreamSynchronize(e->stream) != hipSuccess || hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, prio) != hipSuccess) return GPH_EHIP;
      (void)hipStreamDestroy(e->stream);
      e->stream = ns;
    }
  }
#endif
  return 0;
}

GphGlobal *gph_engine_global_(gph_engine *e) { if (!e) return nullptr; e->G_dirty = true; return e->G_h; }
const GphGlobal *gph_engine_global_ro_(gph_engine *e) { return e ? e->G_h : nullptr; }

int gph_engine_load_loci(gph_engine *e, int64_t L, const int64_t *poff, const uint8_t *leafcodes,
                         const uint16_t *numPhases, const int32_t *counts, const double *mutRates)
{
  if (!e || L <= 0 || !poff || !leafcodes || !numPhases || !counts) return GPH_EARG;
  if (e->loaded) { fprintf(stderr, "gphocs_hip: gph_engine_load_loci called twice on one engine\n"); return GPH_ESTATE; }
  SETDEV(e);
  const int n = e->cfg.n;
  int Pmax = 1;
  for (int64_t g = 0; g < L; g++) { int P = (int)(poff[g + 1] - poff[g]); if (P > Pmax) Pmax = P; if (P < 0) return GPH_EARG; }   /* P == 0: a locus with no informative column (all N) is legal upstream */
  /* pattern counts as 16-bit words when every count of the data set allows it (a count is at most the length of its
   * locus): 2 bytes per pattern of every resident sequence block */
  int cnt16 = 1;
  for (int64_t i = 0; i < poff[L]; i++) if (counts[i] < 0 || counts[i] > 65535) { cnt16 = 0; break; }
  if (const char *ov = getenv("GPH_CNT16")) cnt16 = cnt16 && atoi(ov) != 0;     /* tests: the 32-bit form */
  build_layout(e->lay, n, e->cfg.Kc, e->cfg.K, e->cfg.B, e->cfg.rootPop, Pmax, cnt16);
  /* A locus whose sequence block does not fit the LDS budget of a launch group keeps it in HBM ("huge": its own launch
   * group, the generic (pattern, base) paths read the block through GphSeq -- gph_rt.h; the reference mallocs any P,
   * LocusDataLikelihood.c:251).  The budget is what ONE pattern-rich locus may cost every other one in resident
   * wavefronts: the whole group is launched with the LDS of its largest block (default 32 KB = 5 wavefronts per CU, or
   * the image + 16 KB for the big-tree builds;
   * GPH_HUGE_LDS=bytes -- tests force the HBM path on small loci with it). */
  {
    int budget = 32 * 1024;
    if (budget < (int)sizeof(GphLds) + 16 * 1024) budget = (int)sizeof(GphLds) + 16 * 1024;     /* (the big-tree builds' images alone are 9 - 44 KB) */
    if (const char *ov = getenv("GPH_HUGE_LDS")) {
      budget = atoi(ov);
      if (budget <= 0) { fprintf(stderr, "gphocs_hip: GPH_HUGE_LDS=%s is not a byte count\n", ov); return GPH_EARG; }
    }
    /* the budget covers image + block: what a workgroup can have at all is the CU's 160 KB */
    if (budget > 160 * 1024) budget = 160 * 1024;
    int hp = GPH_WAVE;     /* (a block of up to GPH_WAVE patterns always fits: the lane-per-pattern paths read LDS directly) */
    while (hp < Pmax && (int)sizeof(GphLds) + GPH_Q_BYTES(hp + 1, n, cnt16) <= budget) hp++;
    e->lay.huge_P = hp >= Pmax ? 0x7fffffff : hp;
  }
  e->L = L;
  // slots in decreasing P (stable): within a dispatch the longest wavefronts start first; loci with more
  // than one pattern per lane (P > 64) form their own launch group
  e->h_orig.resize(L);
  for (int64_t g = 0; g < L; g++) e->h_orig[g] = (int32_t)g;
  std::stable_sort(e->h_orig.begin(), e->h_orig.end(), [&](int32_t a, int32_t b) {
    return (poff[a + 1] - poff[a]) > (poff[b + 1] - poff[b]); });
  e->buckets.clear();
  {
    int64_t nwide = 0, nhuge = 0;
    while (nhuge < L && (poff[e->h_orig[nhuge] + 1] - poff[e->h_orig[nhuge]]) > e->lay.huge_P) nhuge++;
    nwide = nhuge;
    while (nwide < L && (poff[e->h_orig[nwide] + 1] - poff[e->h_orig[nwide]]) > GPH_WAVE) nwide++;
    e->n_huge = nhuge;
    if (nhuge > 0 && e->cfg.locus_begin == 0)
      fprintf(stderr, "gphocs_hip: %lld loci with more than %d phased patterns (up to %d) keep their sequence block in HBM\n",
              (long long)nhuge, e->lay.huge_P, Pmax);
    if (nhuge > 0) {        /* sequence block in HBM: no dynamic LDS beyond a token allocation */
      gph_engine::Bucket bk;
      bk.j0 = 0; bk.count = (int)nhuge;
      bk.lds_bytes = 64;
      e->buckets.push_back(bk);
    }
    if (nwide > nhuge) {    /* the generic (pattern, base) mapping, block in LDS (the root sum needs no terms array on the device) */
      const int pw = (int)(poff[e->h_orig[nhuge] + 1] - poff[e->h_orig[nhuge]]);
      gph_engine::Bucket bk;
      bk.j0 = (int)nhuge; bk.count = (int)(nwide - nhuge);
      bk.lds_bytes = GPH_Q_TERMS(pw, n, cnt16);
#ifdef GPH_HOSTEMU
      bk.lds_bytes += 8 * pw;
#endif
      e->buckets.push_back(bk);
    }
    if (nwide < L) {
      const int pn = (int)(poff[e->h_orig[nwide] + 1] - poff[e->h_orig[nwide]]);
      gph_engine::Bucket bk;
      bk.j0 = (int)nwide; bk.count = (int)(L - nwide);
      bk.lds_bytes = GPH_Q_BYTES(pn, n, cnt16);
      /* the per-lane terms of the root reduction through LDS (ordered_sum64_lds: one vector instruction per pattern
       * instead of three) take 8 bytes per pattern behind a locus's block: never at the price of a resident workgroup.
       * LDS is handed out in 1280-byte granules (160 KB / 128, measured: profiles/HISTORY.md, round 1) and the sweep kernel keeps
       * at most GPH_SWEEP_WAVES workgroups per SIMD: the group gets the room of the granules its largest block needs
       * anyway, or -- when even the largest locus's terms do not cost a workgroup -- room for everybody's; a locus takes
       * the LDS form when its terms end inside the allocation (gph_locus.h: lik_compute) */
      {
        auto resident = [](int bytes) { const int g = (bytes + 1279) / 1280; const int w = 128 / g; return w < 4 * GPH_SWEEP_WAVES_ ? w : 4 * GPH_SWEEP_WAVES_; };
        const int base = (int)sizeof(GphLds) + bk.lds_bytes;
        const int full = 8 * ((pn + 7) & ~7);
        e->lay.lds_sum = 1;
        if (resident(base + full) == resident(base)) bk.lds_bytes += full;
        else {
          /* the largest size with as many resident workgroups as the bare block: granules first, then the step of the
           * wavefront cap (4 * GPH_SWEEP_WAVES workgroups per CU do not need whole granules to themselves) */
          int room = (base + 1279) / 1280 * 1280;
          while (resident(room + 1280) == resident(base)) room += 1280;
          bk.lds_bytes = (room - (int)sizeof(GphLds)) & ~15;
        }
        if (const char *ov = getenv("GPH_LDS_SUM")) {     /* tests: force either form for every locus */
          e->lay.lds_sum = atoi(ov) != 0;
          bk.lds_bytes = GPH_Q_BYTES(pn, n, cnt16) + (e->lay.lds_sum ? full : 0);
        }
      }
      e->buckets.push_back(bk);
    }
  }
  /* (a group that was given the rest of its granules for the root sum's terms can be larger than block + 512 bytes) */
  e->lay.lds_bytes = 0;
  for (auto &bk : e->buckets) if (bk.lds_bytes > e->lay.lds_bytes) e->lay.lds_bytes = bk.lds_bytes;
  e->h_cond_off.resize(L + 1);
  e->h_P.resize(L);
  std::vector<uint64_t> seq_off(L + 1);
  std::vector<double> rates(L, 1.0);
  uint64_t off = 0, soff = 0;
  for (int64_t j = 0; j < L; j++) {
    int64_t g = e->h_orig[j];
    int P = (int)(poff[g + 1] - poff[g]);
    e->h_P[j] = P;
    e->h_cond_off[j] = off;
    off += (uint64_t)2 * (n - 1) * P * 32;
    seq_off[j] = soff;
    soff += GPH_Q_BYTES(P, n, cnt16);
    if (P > e->lay.huge_P) soff += (uint64_t)8 * ((P + 1) & ~1);     /* the root reduction's terms of the host build behind a block that lies in HBM */
    if (mutRates) rates[j] = mutRates[g];
  }
  e->h_cond_off[L] = off;
  seq_off[L] = soff;
  e->h_seq_off = seq_off;
  std::vector<char> seq(soff, 0);
  for (int64_t j = 0; j < L; j++) {
    int64_t g = e->h_orig[j];
    int P = e->h_P[j];
    char *blk = seq.data() + seq_off[j];
    for (int p = 0; p < P; p++) {
      for (int i = 0; i < n; i++) {
        uint8_t c = leafcodes[(size_t)(poff[g] + p) * n + i];
        if (c > 4) return GPH_EARG;
        blk[GPH_Q_LEAF + p * GPH_Q_NH(n) + (i >> 1)] |= (char)(c << ((i & 1) << 2));
      }
      ((uint16_t *)(blk + GPH_Q_PHASES(P, n)))[p] = numPhases[poff[g] + p];
      if (cnt16) ((uint16_t *)(blk + GPH_Q_COUNT(P, n)))[p] = (uint16_t)counts[poff[g] + p];
      else ((int32_t *)(blk + GPH_Q_COUNT(P, n)))[p] = counts[poff[g] + p];
    }
  }
  e->cond_bytes = off;
  e->seq_bytes_total = soff;
  e->pages_bytes = (size_t)L * e->lay.page_bytes;
  int rc = 0;
  rc |= dev_alloc((void **)&e->dev.pages, e->pages_bytes);
  rc |= dev_alloc((void **)&e->dev.shadow, e->pages_bytes);
  rc |= dev_alloc((void **)&e->dev.cond, e->cond_bytes);
  rc |= dev_alloc((void **)&e->dev.cond_off, sizeof(uint64_t) * (L + 1));
  rc |= dev_alloc((void **)&e->dev.seq, seq.size());
  rc |= dev_alloc((void **)&e->dev.seq_off, sizeof(uint64_t) * (L + 1));
  rc |= dev_alloc((void **)&e->dev.P, sizeof(int32_t) * 2 * L);
  rc |= dev_alloc((void **)&e->dev.orig, sizeof(int32_t) * L);
  rc |= dev_alloc((void **)&e->dev.out, sizeof(double) * GPH_OUT_SLOTS * L);
  rc |= dev_alloc((void **)&e->dev.stats, sizeof(double) * (2 * e->cfg.K + 2 * e->cfg.B) * L);
  rc |= dev_alloc((void **)&e->d_part, sizeof(double) * 2 * 3 * GPH_RED_BLOCKS * GPH_RED_COLS);
  rc |= dev_alloc((void **)&e->dev.err, sizeof(int32_t));
#ifdef GPH_HOSTEMU
  if (e->dev.err) *e->dev.err = 0;
#else
  /* (null-stream operations; the engine's stream may be a non-blocking one, which does not order itself against them) */
  if (!rc && (hipMemset(e->dev.err, 0, sizeof(int32_t)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)) rc = GPH_EHIP;
#endif
  if (mutRates) rc |= dev_alloc((void **)&e->d_mutRate, sizeof(double) * L);
  if (rc) { fprintf(stderr, "gphocs_hip: device allocation failed\n"); return GPH_EHIP; }
  rc |= h2d(e, (void *)e->dev.cond_off, e->h_cond_off.data(), sizeof(uint64_t) * (L + 1));
  rc |= h2d(e, (void *)e->dev.seq, seq.data(), seq.size());
  rc |= h2d(e, (void *)e->dev.seq_off, seq_off.data(), sizeof(uint64_t) * (L + 1));
  {
    std::vector<int32_t> pu((size_t)2 * L);      /* {phased, unphased} patterns per slot */
    for (int64_t j = 0; j < L; j++) {
      const int64_t g = e->h_orig[j];
      int U = 0;
      for (int64_t p = poff[g]; p < poff[g + 1]; p++) U += numPhases[p] > 0;
      pu[2 * j] = e->h_P[j]; pu[2 * j + 1] = U;
    }
    rc |= h2d(e, (void *)e->dev.P, pu.data(), sizeof(int32_t) * 2 * L);
  }
  rc |= h2d(e, (void *)e->dev.orig, e->h_orig.data(), sizeof(int32_t) * L);
  if (mutRates) rc |= h2d(e, e->d_mutRate, rates.data(), sizeof(double) * L);
  if (rc) return GPH_EHIP;
  e->dev.L = (int32_t)L;
  e->dev.Ltot = (int32_t)e->cfg.L_total;
  e->dev.locus_begin = e->cfg.locus_begin;
  e->loaded = true;
#ifndef GPH_HOSTEMU
  // per-locus kernels use up to the wide group's dynamic LDS size; allow > 64 KiB
  const void *ks[] = {(const void *)k_init, (const void *)k_sweep, (const void *)k_tau_eval, (const void *)k_tau_finish,
                      (const void *)k_mix_eval, (const void *)k_mix_finish,
                      (const void *)k_sync, (const void *)k_check, (const void *)k_lrate_apply, (const void *)k_lrate_prep,
                      (const void *)k_unit};
  int pad_max = 0;
  if (const char *pe = getenv("GPH_LDS_PAD")) {
    for (const char *q = pe; *q;) {
      int w = -1, by = 0;
      if (sscanf(q, "%d:%d", &w, &by) == 2 && w >= 0 && w < 16 && by >= 0) { e->lds_pad[w] = by; pad_max = by > pad_max ? by : pad_max; }
      while (*q && *q != ',') q++;
      if (*q == ',') q++;
    }
  }
  /* image + the largest launch group's block (+ an experiment's padding) must fit the CU's LDS: said with the numbers, whatever
   * made it too big (the message used to blame GPH_LDS_PAD for a block that was too large by itself) */
  if (e->lay.lds_bytes + pad_max + (int)sizeof(GphLds) > 160 * 1024) {
    fprintf(stderr, "gphocs_hip: %d-byte image + %d-byte sequence block%s = %d bytes of LDS per locus, the CU has %d%s\n", (int)sizeof(GphLds),
            e->lay.lds_bytes, pad_max ? " + GPH_LDS_PAD" : "", e->lay.lds_bytes + pad_max + (int)sizeof(GphLds), 160 * 1024,
            getenv("GPH_HUGE_LDS") ? " (GPH_HUGE_LDS is set: lower it, the block then stays in HBM)" : "");
    return GPH_EARG;
  }
  for (auto k : ks) HIPCHK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, e->lay.lds_bytes + pad_max));
#endif
  return 0;
}

int gph_engine_set_model(gph_engine *e, const double *theta, const double *popAge, const double *sampleAge,
                         const double *migRate, const double *bandStart, const double *bandEnd)
{
  if (!e || !theta || !popAge || !sampleAge) return GPH_EARG;
  GphModel &m = e->G_h->model;
  for (int p = 0; p < e->cfg.K; p++) { gg_set_theta(*e->G_h, p, theta[p]); m.popAge[p] = popAge[p]; m.sampleAge[p] = sampleAge[p]; }
  for (int b = 0; b < e->cfg.B; b++) { gg_set_mig(*e->G_h, b, migRate[b]); m.bandStart[b] = bandStart[b]; m.bandEnd[b] = bandEnd[b]; }
  e->model_set = true;
  e->G_dirty = true;
  return 0;
}

int gph_engine_seed(gph_engine *e, uint32_t seed)
{
  if (!e) return GPH_EARG;
  e->seedz = 170u * (seed % 178u) + 137u;   // utils.c:421
  e->seeded = true;
  return 0;
}

int gph_engine_init_genealogies(gph_engine *e, double *sumGen, double *sumData)
{
  if (!e || !e->loaded || !e->seeded || !e->model_set) return GPH_ESTATE;
  SETDEV(e);
  /* the pages are rewritten from scratch: nothing decided on the old chain is owed to them any more (a commit left for the
   * next sweep kernel would stage in the OLD chain's shadow page over the fresh genealogies) */
  { int rcf = flush_pending(e); if (rcf) return rcf; }
  e->mix_owed = e->fin_owed = e->sync_pending = false;
  if (e->G_h->mix_flag || e->G_h->tau_flag) { e->G_h->mix_flag = 0; e->G_h->tau_flag = 0; e->G_dirty = true; }
  PUSH_IF_DIRTY(e);
  LAUNCH(e, 3, k_init, e->seedz, (const double *)e->d_mutRate, e->init_predraws);
  int rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (!rc) rc = reduce_stats(e);
  if (rc) return rc;
  e->G_h->nrec = 0;
  e->G_h->iteration = -1;
  if ((rc = run_stage_now(e, GS_INIT_DONE, 0))) return rc;
  { GphRed R; R.rows = e->h_red; R.world = 1;
    if (sumGen) *sumGen = R.sum(0, 0);
    if (sumData) *sumData = R.sum(0, 1); }
  e->initialized = true;
  return rc;
}

// ---- the stepwise entry points: one reference proposal function each (include/gphocs_hip.h), the result is on the
// host when the call returns.  gph_engine_iteration_() below runs the same launches without the round trips.
int gph_engine_genealogy_sweep(gph_engine *e, int32_t flags, double ftCoal, double ftMig, gph_sweep_result *out)
{
  if (!e || !e->initialized || !out) return GPH_ESTATE;
  SETDEV(e);
  PUSH_IF_DIRTY(e);
  const int with_sync = e->sync_pending ? 8 : 0;   /* flag 8: synchronizeEvents first, in the same kernel */
  e->sync_pending = false;
  GphGlobal &G = *e->G_h;
  const double d0 = G.dataLogLikelihood, l0 = G.logLikelihood;
  const int64_t a0 = G.acc[0], a1 = G.acc[1], a2 = G.acc[2], a7 = G.acc[7];
  int with_mix = 0;
  if (e->mix_owed) {
    /* the owed decision (flag, factor) travels as kernel arguments off the host mirror: it must be what the decision stage wrote */
    if (!e->mirror_current) { int rcm = finish_sync(e); if (rcm) return rcm; }
    e->mix_owed = false;
    if (e->G_h->mix_flag) with_mix = 16;
  }
  LAUNCH(e, 0, k_sweep, (int)flags | with_sync | with_mix, ftCoal, ftMig, e->G_h->mix_c, e->G_h->mix_lnc);
  int rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (!rc) rc = reduce_stats(e);
  if (rc) return rc;
  if ((rc = run_stage_now(e, GS_SWEEP_DONE, with_sync))) return rc;
  /* the per-class deltas, as the caller adds them (GPhoCS.c:1495-1545) */
  GphRed R; R.rows = e->h_red; R.world = 1;
  out->accepted_internal = G.acc[0] - a0;
  out->accepted_mignode = G.acc[1] - a1;
  out->accepted_spr = G.acc[2] - a2;
  out->total_mig_nodes = G.acc[7] - a7;
  out->dData_internal = R.sum(0, 3);
  out->dLog_internal = R.sum(0, 4);
  out->dLog_mignode = R.sum(0, 5);
  out->dData_spr = R.sum(0, 6);
  out->dLog_spr = R.sum(0, 7);
  (void)d0; (void)l0;
  return 0;
}

int gph_engine_tau_evaluate(gph_engine *e, const gph_tau_args *a, gph_tau_result *out)
{
  if (!e || !e->initialized || !a || !out) return GPH_ESTATE;
  if (a->num_aff > 2 * GPH_MAXB) return GPH_EARG;
  SETDEV(e);
  { int rcs = flush_sync(e, true); if (rcs) return rcs; }
  GphTauArgs &A = e->G_h->tau;
  memset(&A, 0, sizeof A);
  A.ap = a->ap; A.son0 = a->son0; A.son1 = a->son1; A.isRoot = a->isRoot; A.num_aff = a->num_aff; A.mode = a->mode;
  A.tauold = a->tauold; A.taunew = a->taunew; A.taub0 = a->taub0; A.taub1 = a->taub1;
  A.taufactor0 = a->taufactor0; A.taufactor1 = a->taufactor1;
  for (int i = 0; i < a->num_aff; i++) {
    A.aff_bands[i] = a->aff_bands[i];
    A.start_or_end[i] = a->start_or_end[i];
    A.new_band_ages[i] = a->new_band_ages[i];
  }
  int rc = push_G(e);
  if (rc) return rc;
  LAUNCH(e, 1, k_tau_eval, 0);     /* stepwise: the finish of the previous proposal was a call of its own */
  if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
  if ((rc = run_stage_now(e, GS_COUNT_ONLY, 1))) return rc;
  // first conflicting locus in serial (input) order: loci after it were never touched by
  // the reference; slot 14 holds the global index of a conflicting locus (1e300 = none)
  GphRed R; R.rows = e->h_red; R.world = 1;   /* the row as combined over the ranks by run_stage */
  out->ntj0 = (int64_t)R.sum(0, 0);
  out->ntj1 = (int64_t)R.sum(0, 1);
  out->genDelta = R.sum(0, 3);
  out->dataDelta = R.sum(0, 4);
  out->first_conflict_locus = R.mn(0, 14) < 1e299 ? (int64_t)R.mn(0, 14) : -1;
  return 0;
}

int gph_engine_tau_commit(gph_engine *e)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  e->G_h->tau_flag = 1;
  e->G_h->tau_limit = (long long)1 << 62;
  gg_fill_fin(*e->G_h, 1, e->G_h->tau_limit);
  int rc = push_G(e);
  if (rc) return rc;
  LAUNCH(e, 5, k_tau_finish, 0);
  return 0;
}

int gph_engine_tau_revert(gph_engine *e, int64_t first_conflict)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  e->G_h->tau_flag = 0;
  e->G_h->tau_limit = first_conflict >= 0 ? (long long)first_conflict : (long long)1 << 62;
  gg_fill_fin(*e->G_h, 0, e->G_h->tau_limit);
  int rc = push_G(e);
  if (rc) return rc;
  LAUNCH(e, 5, k_tau_finish, 0);
  return 0;
}

int gph_engine_mixing_evaluate(gph_engine *e, double c, double *dataDelta)
{
  if (!e || !e->initialized || !dataDelta) return GPH_ESTATE;
  SETDEV(e);
  { int rcs = flush_sync(e, true); if (rcs) return rcs; }
  e->G_h->mix_c = c;
  int rc = push_G(e);
  if (rc) return rc;
  LAUNCH(e, 2, k_mix_eval, 0);
  if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
  if ((rc = run_stage_now(e, GS_COUNT_ONLY, 2))) return rc;
  GphRed R; R.rows = e->h_red; R.world = 1;
  *dataDelta = R.sum(0, 0);
  return 0;
}

int gph_engine_mixing_commit(gph_engine *e, double c, double lnc)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  e->G_h->mix_flag = 1; e->G_h->mix_c = c; e->G_h->mix_lnc = lnc;
  int rc = push_G(e);
  if (rc) return rc;
  LAUNCH(e, 7, k_mix_finish, 0);
  return 0;
}

// mixing reject (GPhoCS.c:4881-4887): revertToSaved restores every locus exactly, and
// the evaluated state only ever lived in the shadow pages -- nothing to do.
int gph_engine_mixing_revert(gph_engine *e) { return e ? 0 : GPH_EARG; }

static int finish_owed(gph_engine *e);
// every finish that has been left for a later kernel, now (the caller is about to edit the main pages by other means)
static int owed_finishes(gph_engine *e)
{
  int rc = finish_owed(e);
  if (!rc) rc = mix_finish_owed(e);
  return rc;
}
static int apply_list(gph_engine *e)
{
  { int rco = owed_finishes(e); if (rco) return rco; }
#ifdef GPH_HOSTEMU
  for (int64_t g = 0; g < e->L; g++) apply_list_locus(e->dev.pages + (size_t)g * e->lay.page_bytes, e->lay, e->G_h->apply, e->G_h->napply);
#else
  { int rcf = flush_pending(e); if (rcf) return rcf; }
  LAUNCH_PRE(e);
  hipLaunchKernelGGL(k_apply_list, dim3((unsigned)((e->L + 255) / 256)), dim3(256), 0, e->stream, e->ka, e->dev);
  HIPCHK(hipGetLastError());
  e->n_launches++;
#endif
  return 0;
}

int gph_engine_apply_theta(gph_engine *e, int32_t pop, double lnc, double thetaold, double thetanew)
{
  if (!e || !e->initialized || pop < 0 || pop >= e->cfg.K) return GPH_EARG;
  SETDEV(e);
  { int rco = owed_finishes(e); if (rco) return rco; }    /* the touch-up edits the MAIN page: a commit that still sits in the shadow page goes first */
  GphGlobal &G = *e->G_h;
  G.napply = 1;
  G.apply[0].kind = 0; G.apply[0].idx = pop; G.apply[0].lnc = lnc; G.apply[0].diff = (1 / thetanew - 1 / thetaold);
  int rc = push_G(e);
  if (rc) return rc;
  return apply_list(e);
}

int gph_engine_apply_migrate(gph_engine *e, int32_t band, double lnc, double old_rate, double new_rate)
{
  if (!e || !e->initialized || band < 0 || band >= e->cfg.B) return GPH_EARG;
  SETDEV(e);
  { int rco = owed_finishes(e); if (rco) return rco; }    /* the touch-up edits the MAIN page: a commit that still sits in the shadow page goes first */
  GphGlobal &G = *e->G_h;
  G.napply = 1;
  G.apply[0].kind = 1; G.apply[0].idx = band; G.apply[0].lnc = lnc; G.apply[0].diff = (new_rate - old_rate);
  int rc = push_G(e);
  if (rc) return rc;
  return apply_list(e);
}

int gph_engine_get_totals(gph_engine *e, double *cs, double *nc, double *ms, double *nm)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  const int K = e->cfg.K, B = e->cfg.B;
  PUSH_IF_DIRTY(e);
  { int rcm = mix_finish_owed(e); if (rcm) return rcm; }    /* the statistics of an accepted mixing proposal are scaled by its commit */
  int rc = reduce_stats(e);
  if (rc) return rc;
  if ((rc = run_stage_now(e, GS_TOTALS, 0))) return rc;
  const GphGlobal &G = *e->G_h;
  for (int p = 0; p < K; p++) { if (cs) cs[p] = G.tot_coal[p]; if (nc) nc[p] = G.tot_ncoal[p]; }
  for (int b = 0; b < B; b++) { if (ms) ms[b] = G.tot_mig[b]; if (nm) nm[b] = G.tot_nmig[b]; }
  return rc;
}

int gph_engine_synchronize(gph_engine *e, int32_t refresh, double *oldGen, double *newGen)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  if (!refresh) {
    /* nothing is returned without a refresh: the pass is deferred into the next sweep kernel (one page
     * round trip less per iteration); any other page-touching call runs it first */
    e->sync_pending = true;
    if (oldGen) *oldGen = 0.0;
    if (newGen) *newGen = 0.0;
    return 0;
  }
  { int rcs = flush_sync(e, true); if (rcs) return rcs; }
  PUSH_IF_DIRTY(e);
  LAUNCH(e, 8, k_sync, (int)refresh);
  int rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (rc) return rc;
  const double l0 = e->G_h->logLikelihood;
  if ((rc = run_stage_now(e, GS_REFRESH_DONE, 0))) return rc;   /* arg 0: the caller applies the difference */
  (void)l0;
  GphRed R; R.rows = e->h_red; R.world = 1;
  if (oldGen) *oldGen = R.sum(0, 1);
  if (newGen) *newGen = R.sum(0, 2);
  return rc;
}

int gph_engine_check_all(gph_engine *e, int32_t *ok, double *sumData, double *sumGen)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  { int rcs = flush_sync(e, true); if (rcs) return rcs; }
  PUSH_IF_DIRTY(e);
  LAUNCH(e, 4, k_check, 0);
  int rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (!rc) rc = reduce_stats(e);
  if (rc) return rc;
  e->G_h->nrec = 0;
  rc = run_stage_now(e, GS_CHECK_DONE, 0);
  if (ok) *ok = e->G_h->error != 9999;
  if (e->G_h->error == 9999) { e->G_h->error = 0; e->G_h->error_locus = -1; e->G_dirty = true; rc = 0; }
  { GphRed R; R.rows = e->h_red; R.world = 1;
    if (sumData) *sumData = R.sum(0, 1);
    if (sumGen) *sumGen = R.sum(0, 2); }
  return rc;
}

int gph_engine_get_counters(gph_engine *e, gph_counters *out, int32_t reset)
{
  if (!e || !out) return GPH_EARG;
  /* summed over ALL ranks (the counters ride in the reduced rows); e->counters = the values at the last reset */
  const GphGlobal &G = *e->G_h;
  double ev = 0, nd = 0, by = 0;
  for (int k = 0; k < 16; k++) { ev += G.cls_evals[k]; nd += G.cls_nodes[k]; by += G.cls_bytes[k]; }
  out->evals = (int64_t)ev - e->counters.evals;
  out->eval_nodes = (int64_t)nd - e->counters.eval_nodes;
  out->eval_bytes = by - e->counters.eval_bytes;
  out->not_enough_migs = (int64_t)G.cnt_notenough - e->counters.not_enough_migs;
  if (reset) { e->counters.evals = (int64_t)ev; e->counters.eval_nodes = (int64_t)nd; e->counters.eval_bytes = by; e->counters.not_enough_migs = (int64_t)G.cnt_notenough; }
  return 0;
}

int gph_engine_last_kernel_ms(gph_engine *e, int32_t which, double *ms)
{
  if (!e || !ms || which < 0 || which >= 16) return GPH_EARG;
  *ms = e->last_ms[which];
  return 0;
}

int64_t gph_engine_num_loci(gph_engine *e) { return e ? e->L : 0; }

int gph_engine_class_stats(gph_engine *e, int32_t which, double *out5, int32_t reset)
{
  if (!e || !out5 || which < 0 || which >= 16) return GPH_EARG;
  const GphGlobal &G = *e->G_h;
  out5[0] = e->cls_launches[which]; out5[1] = e->cls_ms[which]; out5[2] = G.cls_evals[which] - e->cls_evals0[which];
  out5[3] = G.cls_bytes[which] - e->cls_bytes0[which]; out5[4] = G.cls_nodes[which] - e->cls_nodes0[which];
  if (reset) {
    e->cls_launches[which] = e->cls_ms[which] = 0;
    e->cls_evals0[which] = G.cls_evals[which]; e->cls_bytes0[which] = G.cls_bytes[which]; e->cls_nodes0[which] = G.cls_nodes[which];
  }
  return 0;
}

// host synchronisations, cross-rank exchanges and kernel launches since the engine was created
int gph_engine_host_stats(gph_engine *e, int64_t *syncs, int64_t *collectives, int64_t *launches, int32_t *resident_mode)
{
  if (!e) return GPH_EARG;
  if (syncs) *syncs = e->n_syncs;
  if (collectives) *collectives = e->n_collectives;
  if (launches) *launches = e->n_launches;
  if (resident_mode) *resident_mode = resident(e) ? 1 : 0;
  return 0;
}
// classes whose launches are bracketed by HIP events (bit k = class k of gph_engine_last_kernel_ms); default all
int gph_engine_set_timing(gph_engine *e, uint32_t class_mask)
{
  if (!e) return GPH_EARG;
  e->timing_mask = class_mask;
  return 0;
}

// out[5n]: exp(x), log(x), sqrt(|x|), x/y, floor(x) evaluated ON THE DEVICE
int gph_debug_math(const double *x, const double *y, int32_t n, double *out, int32_t device)
{
  if (!x || !y || !out || n <= 0) return GPH_EARG;
#ifdef GPH_HOSTEMU
  (void)device;
  for (int i = 0; i < n; i++) {
    out[i] = gph_exp(x[i]); out[n + i] = gph_log(x[i]); out[2 * n + i] = sqrt(fabs(x[i]));
    out[3 * n + i] = x[i] / y[i]; out[4 * n + i] = floor(x[i]);
  }
  return 0;
#else
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipMalloc((void **)&dx, sizeof(double) * n));
  HIPCHK(hipMalloc((void **)&dy, sizeof(double) * n));
  HIPCHK(hipMalloc((void **)&dout, sizeof(double) * 5 * n));
  HIPCHK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice));
  GphKargs ka;
  memset(&ka, 0, sizeof ka);
  fill_math_constants(ka);
  if (fill_math_tables(ka)) return GPH_EHIP;
  hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, 0, ka, dx, dy, (int)n, dout);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout, sizeof(double) * 5 * n, hipMemcpyDeviceToHost));
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
  return 0;
#endif
}

#ifndef GPH_BUILD_ID
#define GPH_BUILD_ID "unidentified"
#endif
const char *gph_build_id(void) { return GPH_BUILD_ID; }
const char *gph_build_compiler(void)
{
#ifdef __clang_version__
  return "clang " __clang_version__;
#else
  return "g++ " __VERSION__;
#endif
}
const char *gph_runtime_version(void)
{
  static char buf[96];
#ifdef GPH_HOSTEMU
  snprintf(buf, sizeof buf, "host emulation (no HIP runtime)");
#else
  int rt = 0, dr = 0;
  if (hipRuntimeGetVersion(&rt) != hipSuccess) rt = -1;
  if (hipDriverGetVersion(&dr) != hipSuccess) dr = -1;
  snprintf(buf, sizeof buf, "HIP runtime %d, driver %d", rt, dr);
#endif
  return buf;
}

// Decision-level transcript (upstream -DLOG_STEPS: GPhoCS.c:2363-2401, 2540-2577, 2654-2718, patch.c:1451-1454):
// the three genealogy sweeps append one record per printed fragment for the selected loci -- kind 1 node-age proposal
// {node, t, tnew}, 2 considerEventMove {event, source pop, target pop, old age, new age, new event}, 3 decision {accepted,
// lnacceptance}, 4 migration-node proposal {migration node, t, tnew}, 5 SPR {node, father, father's population}.  Compiled
// into the test builds only (GPH_LOGSTEPS: the host build of tests/hostemu and libgphocs_hip_plain.so).
int gph_engine_steplog_enable(gph_engine *e, const int64_t *loci, int32_t n, int32_t cap)
{
#if !defined(GPH_LOGSTEPS) && !defined(GPH_BBCOUNT)    /* (GPH_BBCOUNT: tools/bbcount.sh borrows the record buffer for basic-block counters) */
  (void)e; (void)loci; (void)n; (void)cap;
  return GPH_EARG;      /* not compiled into this build of the library */
#else
  if (!e || !e->loaded || n < 0 || cap < 1 || (n > 0 && !loci)) return GPH_EARG;
  SETDEV(e);
  /* (before the first initialisation nothing is in flight, and the host mirror of the chain state -- not yet pushed -- must
   * not be overwritten by a read-back) */
  if (e->initialized) { int rcs = flush_sync(e, true); if (!rcs) rcs = finish_sync(e); if (rcs) return rcs; }
  eng_free(e, (void *)e->dev.slog_map); eng_free(e, e->dev.slog); eng_free(e, e->dev.slog_n);
  e->dev.slog_map = nullptr; e->dev.slog = nullptr; e->dev.slog_n = nullptr; e->dev.slog_cap = 0;
  e->slog_sel = n;
  if (n == 0) return 0;
  std::vector<int32_t> map((size_t)e->L, -1);
  for (int64_t j = 0; j < e->L; j++)
    for (int k = 0; k < n; k++)
      if (e->h_orig[j] + e->cfg.locus_begin == loci[k]) map[j] = k;
  std::vector<int32_t> zero((size_t)n, 0);
  /* allocate into locals and publish them to the kernels' view only when every copy has succeeded: a partial failure
   * must not leave dev.slog_map pointing at uninitialised memory */
  int32_t *d_map = nullptr, *d_n = nullptr;
  double *d_log = nullptr;
  if (dev_alloc((void **)&d_map, sizeof(int32_t) * e->L) || dev_alloc((void **)&d_log, sizeof(double) * 8 * (size_t)n * cap) ||
      dev_alloc((void **)&d_n, sizeof(int32_t) * n) ||
      h2d(e, d_map, map.data(), sizeof(int32_t) * e->L) || h2d(e, d_n, zero.data(), sizeof(int32_t) * n)) {
    eng_free(e, d_map); eng_free(e, d_log); eng_free(e, d_n);
    e->slog_sel = 0;
    return GPH_EHIP;
  }
  e->dev.slog_map = d_map; e->dev.slog = d_log; e->dev.slog_n = d_n;
  e->dev.slog_cap = cap;
  return 0;
#endif
}
// records of selected locus `idx` so far (out: 8 doubles per record, at most max_records of them; *nrec = records
// written by the kernels, which may exceed the capacity given to _enable); reset != 0 empties the buffer
int gph_engine_steplog_fetch(gph_engine *e, int32_t idx, double *out, int32_t max_records, int32_t *nrec, int32_t reset)
{
#if !defined(GPH_LOGSTEPS) && !defined(GPH_BBCOUNT)
  (void)e; (void)idx; (void)out; (void)max_records; (void)nrec; (void)reset;
  return GPH_EARG;
#else
  if (!e || !e->dev.slog || idx < 0 || idx >= e->slog_sel || !nrec) return GPH_EARG;
  SETDEV(e);
  { int rcs = finish_sync(e); if (rcs) return rcs; }
  int32_t n = 0;
  if (d2h(e, &n, e->dev.slog_n + idx, sizeof n)) return GPH_EHIP;
  *nrec = n;
#ifdef GPH_BBCOUNT
  n = e->dev.slog_cap;       /* the whole buffer: it holds counters, not records */
#endif
  const int32_t m = std::min(std::min(n, e->dev.slog_cap), max_records);
  if (out && m > 0 && d2h(e, out, e->dev.slog + (size_t)idx * e->dev.slog_cap * 8, sizeof(double) * 8 * (size_t)m)) return GPH_EHIP;
  if (reset) { const int32_t z = 0; if (h2d(e, e->dev.slog_n + idx, &z, sizeof z)) return GPH_EHIP; }
  return 0;
#endif
}

int gph_engine_hbm_bytes(gph_engine *e, double *bytes)
{
  if (!e || !bytes) return GPH_EARG;
  *bytes = 2.0 * e->pages_bytes + (double)e->cond_bytes + (double)e->seq_bytes_total + 8.0 * GPH_OUT_SLOTS * e->L;
  return 0;
}

// starting locus rates in input order (VAR start-up or any caller-supplied rates); `draws` = rndu() draws every
// locus's stream has spent producing them.  Before gph_engine_init_genealogies.
int gph_engine_set_locus_rates(gph_engine *e, const double *rates, int32_t draws, int32_t variable)
{
  if (!e || !e->loaded || e->initialized || !rates || draws < 0) return GPH_ESTATE;
  SETDEV(e);
  std::vector<double> r(e->L);
  for (int64_t j = 0; j < e->L; j++) r[j] = rates[e->h_orig[j]];
  if (!e->d_mutRate && dev_alloc((void **)&e->d_mutRate, sizeof(double) * e->L)) return GPH_EHIP;
  if (h2d(e, e->d_mutRate, r.data(), sizeof(double) * e->L)) return GPH_EHIP;
  e->init_predraws = draws;
  e->var_rates = variable != 0;
  return 0;
}

// UpdateLocusRate (GPhoCS.c:4598-4680): a serial scan by one wavefront + a parallel write-back (gph_kernels.h)
// one vector of doubles from the rank that owns it to every rank, through the sum all-reduce (the others add
// zeros: exact), in pieces small enough for any caller-supplied hook (<= 24 doubles a call)
static int xreduce(gph_engine *e, double *sums, int nsum, double *mins, int nmin)
{
  int rc = 0;
  if (e->allreduce) rc = e->allreduce(e->allreduce_user, sums, nsum, mins, nmin);
  else if (e->comm) rc = gph_comm_allreduce_host(e->comm, sums, nsum, mins, nmin);
  else return 0;
  e->n_collectives++;
  return rc ? GPH_EHIP : 0;
}
static bool multi_rank(const gph_engine *e) { return e->allreduce || (e->comm && gph_comm_world(e->comm) > 1); }
static int lr_share(gph_engine *e, bool mine, std::vector<double> &v)
{
  if (!multi_rank(e)) return 0;
  for (size_t off = 0; off < v.size(); off += 24) {
    double buf[24];
    const int nn = (int)(v.size() - off < 24 ? v.size() - off : 24);
    for (int i = 0; i < nn; i++) buf[i] = mine ? v[off + i] : 0.0;
    int rc = xreduce(e, buf, nn, nullptr, 0);
    if (rc) return rc;
    for (int i = 0; i < nn; i++) v[off + i] = buf[i];
  }
  return 0;
}

int gph_engine_locus_rate_update(gph_engine *e, double finetune, double alpha, gph_locus_rate_result *io)
{
  if (!e || !e->initialized || !io) return GPH_ESTATE;
  io->accepted = 0;
  if (finetune <= 0.0) return 0;                       /* GPhoCS.c:4606 */
  const bool owner = e->cfg.locus_begin == 0;          /* this rank holds the reference locus (genRateRef = 0) */
  SETDEV(e);
  if (!multi_rank(e) && (!owner || e->cfg.L_total != e->L)) {
    fprintf(stderr, "gphocs_hip: UpdateLocusRate over a shard of the loci needs the all-reduce hook (gph_engine_set_allreduce)\n");
    return GPH_EARG;
  }
  { int rcs = flush_sync(e, true); if (rcs) return rcs; }
  PUSH_IF_DIRTY(e);
  const GphLayout &y = e->lay;
  const int n = e->cfg.n, N = 2 * n - 1;
  int rc = 0;
  /* ---- the reference locus as every rank needs it: node records + rate, likelihood, root (every call), sequence
   * block (first call: it never changes).  32-bit words travel as doubles */
  std::vector<char> refpg(y.page_bytes, 0);
  int jr = -1;
  if (owner) {
    for (int64_t j = 0; j < e->L; j++) if (e->h_orig[j] == 0) jr = (int)j;
    if ((rc = d2h(e, refpg.data(), e->dev.pages + (size_t)jr * y.page_bytes, y.page_bytes))) return rc;
  }
  {
    std::vector<double> v(4 * N + 4, 0.0);
    if (owner) {
      const uint32_t *w = (const uint32_t *)(refpg.data() + y.o_nd);
      for (int i = 0; i < 4 * N; i++) v[i] = (double)w[i];
      v[4 * N] = ((const double *)(refpg.data() + y.o_fscal))[FS_MUTRATE];
      v[4 * N + 1] = ((const double *)(refpg.data() + y.o_fscal))[FS_DATALNL];
      v[4 * N + 2] = (double)((const int32_t *)(refpg.data() + y.o_iscal))[IS_ROOT];
      v[4 * N + 3] = (double)e->h_P[jr];
    }
    if ((rc = lr_share(e, owner, v))) return rc;
    if (!owner) {
      uint32_t *w = (uint32_t *)(refpg.data() + y.o_nd);
      for (int i = 0; i < 4 * N; i++) w[i] = (uint32_t)v[i];
      ((double *)(refpg.data() + y.o_fscal))[FS_MUTRATE] = v[4 * N];
      ((double *)(refpg.data() + y.o_fscal))[FS_DATALNL] = v[4 * N + 1];
      ((int32_t *)(refpg.data() + y.o_iscal))[IS_ROOT] = (int32_t)v[4 * N + 2];
    }
    e->lr.ref_P = (int32_t)v[4 * N + 3];
    e->lr.rref0 = v[4 * N];
    e->lr.likref0 = v[4 * N + 1];
  }
  const int Pr = e->lr.ref_P, Pmax = y.Pmax > Pr ? y.Pmax : Pr;
  if (!e->d_lrec) {
    std::vector<int32_t> slot_of(e->L);
    for (int64_t j = 0; j < e->L; j++) slot_of[e->h_orig[j]] = (int32_t)j;
    // dynamic LDS of the scan: guest sequence block | reference sequence block | guest nodes | reference nodes | scratch
    const int seqb = align_up(GPH_Q_TERMS(Pmax, n, y.cnt16) + (Pmax > GPH_WAVE ? 8 * Pmax : 0), 16), ndb = N * (int)sizeof(GphNode);
    const int fixed = 2 * seqb + 2 * ndb;
    /* the compiled program of the reference locus is a lane-per-node construction: not in the big-tree variant */
    const int progb = GPH_BIG_TREE ? 0 : (n - 1) * GPH_WAVE * 16;
    const int lfb = !GPH_BIG_TREE && Pr >= 1 && Pr <= GPH_WAVE ? n * Pr * 32 : 0;      // the reference locus's leaves as conditional arrays
    const int budget = 96 * 1024 - (int)sizeof(GphLds) - progb - align_up(N * 8, 16) - lfb;
    int Pscr = (budget - fixed) / ((n - 1) * 32);
    if (const char *ov = getenv("GPH_LR_PSCR")) Pscr = atoi(ov) < Pscr ? atoi(ov) : Pscr;   /* tests: force the global-scratch path */
    if (Pscr > Pmax) Pscr = Pmax;
    if (Pscr < 0) Pscr = 0;
    const double finetune0 = e->lr.finetune; (void)finetune0;
    const int32_t refP = e->lr.ref_P; const double r0 = e->lr.rref0, l0 = e->lr.likref0;
    memset(&e->lr, 0, sizeof e->lr);
    e->lr.ref_P = refP; e->lr.rref0 = r0; e->lr.likref0 = l0;
    e->lr.o_rseq = seqb; e->lr.o_gnd = 2 * seqb; e->lr.o_rnd = 2 * seqb + ndb; e->lr.o_scr = fixed; e->lr.Pscr = Pscr;
    // behind the scratch: the reference locus's compiled program (one 16-byte entry per step and lane) and edge probabilities
    e->lr.o_prog = fixed + (n - 1) * Pscr * 32;
    e->lr.o_pe = e->lr.o_prog + progb;
    e->lr.o_lf = lfb ? e->lr.o_pe + align_up(N * 8, 16) : 0;
    e->lr_lds_bytes = e->lr.o_pe + align_up(N * 8, 16) + lfb;
    /* The scan keeps TWO sequence blocks in its dynamic LDS (the reference locus's, and the one it re-evaluates), each sized for
     * the largest locus of the rank; loci whose block stays in HBM for the per-locus kernels ("huge", round 5) are staged here like
     * any other (lr_load copies from HBM).  What does not fit next to the static image cannot run: ALL ranks learn it in one
     * exchange and fail together before any kernel of the update (a rank returning alone would leave the others in the next
     * collective -- ADVICE round 5). */
    {
      const int lds_max = 160 * 1024;
      double too_big[1] = {e->lr_lds_bytes + (int)sizeof(GphLds) > lds_max ? 1.0 : 0.0};
      if (multi_rank(e) && (rc = xreduce(e, too_big, 1, nullptr, 0))) return rc;
      if (too_big[0] != 0.0) {
        if (e->lr_lds_bytes + (int)sizeof(GphLds) > lds_max)
          fprintf(stderr, "gphocs_hip: locus-mut-rate VAR: the serial scan of UpdateLocusRate needs %d bytes of LDS for two sequence "
                          "blocks of up to %d phased patterns (%d bytes each) next to the %d-byte image; the CU has %d\n",
                  e->lr_lds_bytes, Pmax, seqb, (int)sizeof(GphLds), lds_max);
        memset(&e->lr, 0, sizeof e->lr);
        return GPH_EARG;
      }
    }
    e->lr.ref_seq_bytes = GPH_Q_BYTES(Pr, n, y.cnt16);
    rc |= dev_alloc((void **)&e->d_lrec, sizeof(GphLrRec) * e->L);
    rc |= dev_alloc((void **)&e->d_lpre, sizeof(GphLrPre) * e->L);
    rc |= dev_alloc((void **)&e->d_slot_of, sizeof(int32_t) * e->L);
    rc |= dev_alloc((void **)&e->d_lr_result, sizeof(double) * 16);
    rc |= dev_alloc((void **)&e->d_lr_gscr, Pmax > Pscr ? sizeof(double) * 4 * (size_t)(n - 1) * Pmax : 16);
    rc |= dev_alloc((void **)&e->d_ref_page, y.page_bytes);
    rc |= dev_alloc((void **)&e->d_ref_seq, e->lr.ref_seq_bytes + 16);
    if (rc) return GPH_EHIP;
    if (h2d(e, e->d_slot_of, slot_of.data(), sizeof(int32_t) * e->L)) return GPH_EHIP;
    e->lr.result = e->d_lr_result; e->lr.rec = e->d_lrec; e->lr.pre = e->d_lpre; e->lr.slot_of = e->d_slot_of; e->lr.gscr = e->d_lr_gscr;
    /* the reference locus's sequence block: the owner's copy goes round once */
    std::vector<char> seqblk(e->lr.ref_seq_bytes + 16, 0);
    if (owner && e->lr.ref_seq_bytes > 0 &&
        (rc = d2h(e, seqblk.data(), e->dev.seq + e->h_seq_off[jr], e->lr.ref_seq_bytes))) return rc;
    {
      std::vector<double> v(e->lr.ref_seq_bytes / 4, 0.0);
      if (owner) for (size_t i = 0; i < v.size(); i++) v[i] = (double)((const uint32_t *)seqblk.data())[i];
      if ((rc = lr_share(e, owner, v))) return rc;
      if (!owner) for (size_t i = 0; i < v.size(); i++) ((uint32_t *)seqblk.data())[i] = (uint32_t)v[i];
    }
    if (e->lr.ref_seq_bytes > 0 && h2d(e, e->d_ref_seq, seqblk.data(), e->lr.ref_seq_bytes)) return GPH_EHIP;
    e->lr.ref_seq = e->d_ref_seq;
    e->lr.ref_page = e->d_ref_page;
#ifndef GPH_HOSTEMU
    HIPCHK(hipFuncSetAttribute((const void *)k_lrate_scan, hipFuncAttributeMaxDynamicSharedMemorySize, e->lr_lds_bytes));
#endif
  }
  if (h2d(e, e->d_ref_page, refpg.data(), y.page_bytes)) return GPH_EHIP;
  e->lr.finetune = finetune; e->lr.alpha = alpha;
  e->lr.first = owner ? 1 : 0;
  LAUNCH(e, 11, k_lrate_prep, finetune, e->d_lpre);
  { int rcp = reduce_local(e, 0, GPH_OUT_SLOTS); if (!rcp) rcp = run_stage_now(e, GS_COUNT_ONLY, 11); if (rcp) return rcp; }
  /* ---- the scan, rank after rank in locus order: state = {next locus, rref, likref, dataLogLikelihood,
   * logLikelihood, rateVar, accepted, prepared rates used}; the rank whose block starts at `next` scans and
   * publishes the state, the others add zeros */
  double st[8] = {0.0, e->lr.rref0, e->lr.likref0, io->dataLogLikelihood, io->logLikelihood, io->rateVar, 0.0, 0.0};
  double res[16] = {0};
  for (int guard = 0; (int64_t)st[0] < e->cfg.L_total; guard++) {
    if (guard > 1 << 20) return GPH_ESTATE;
    const bool mine = (int64_t)st[0] == e->cfg.locus_begin;
    double nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mine) {
      e->lr.rref0 = st[1]; e->lr.likref0 = st[2];
      e->lr.dataLnL = st[3]; e->lr.logL = st[4]; e->lr.rateVar = st[5];
      LAUNCH1(e, 9, k_lrate_scan, e->lr_lds_bytes, e->lr);
      if ((rc = d2h(e, res, e->d_lr_result, sizeof res))) return rc;
#ifndef GPH_HOSTEMU
      collect_times(e);
#endif
      if (res[4] != 0.0) { fprintf(stderr, "gphocs_hip: Fatal Error %04d reported by the locus-rate scan\n", (int)res[4]); return GPH_EKERNEL; }
      nx[0] = (double)(e->cfg.locus_begin + e->L); nx[1] = res[13]; nx[2] = res[14];
      nx[3] = res[1]; nx[4] = res[2]; nx[5] = res[3]; nx[6] = st[6] + res[0]; nx[7] = st[7] + res[5];
    }
    if (multi_rank(e)) { if ((rc = xreduce(e, nx, 8, nullptr, 0))) return rc; }
    else if (!mine) return GPH_ESTATE;
    if (nx[0] <= st[0]) { fprintf(stderr, "gphocs_hip: the ranks' locus blocks do not tile 0..L_total\n"); return GPH_EARG; }
    memcpy(st, nx, sizeof st);
  }
  if (owner) {
    /* the reference locus's record: its last accepted rate / likelihood and the number of accepted proposals */
    GphLrRec r;
    const int32_t *is = (const int32_t *)(refpg.data() + y.o_iscal);
    r.rate = st[1]; r.lnl = st[2]; r.rx = (uint32_t)is[IS_RX]; r.ry = (uint32_t)is[IS_RY]; r.rz = (uint32_t)is[IS_RZ];
    r.flag = (int32_t)st[6] << 2;
    if (h2d(e, e->d_lrec + jr, &r, sizeof r)) return GPH_EHIP;
  }
  LAUNCH(e, 10, k_lrate_apply, (const GphLrRec *)e->d_lrec);
  rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (!rc) rc = run_stage_now(e, GS_COUNT_ONLY, 10);
  if (rc) return rc;
  io->accepted = (int64_t)st[6];
  io->dataLogLikelihood = st[3];
  io->logLikelihood = st[4];
  io->rateVar = st[5];
  e->lr_hits = st[7];
#ifdef GPH_LRSTAMP
  fprintf(stderr, "evaluator stamps: setup+exp %.0f, nodes %.0f, root %.0f cycles per call (%.0f calls, %.1f steps)\n", res[8] / res[11], res[9] / res[11], res[10] / res[11], res[11], res[12] / res[11]);
#endif
  if (getenv("GPH_LR_VERBOSE"))
    fprintf(stderr, "gphocs_hip: locus-rate scan: %.0f of %lld proposals decided with the prepared likelihood; %.3g shader cycles in %.3g s (%.0f MHz)\n",
            st[7], (long long)(e->cfg.L_total - 1), res[6], res[7] / 1e8, res[7] > 0 ? res[6] / (res[7] / 1e8) / 1e6 : 0.0);
  return 0;
}

// The parts of an iteration, one per function performMCMC calls (GPhoCS.c:1495-1821).  gph_engine_iteration_ queues them
// all and synchronises once; gph_engine_part_ runs ONE and returns with its result on the host (the reference's own
// performMCMC driving the engine through the functions of GPhoCS.h:84-100: oracle/integration_binding.c).
// e->fin_owed: the commit / revert of the last decided UpdateTau / UpdateSampleAge proposal has not run yet -- it rides
// at the head of the next evaluate kernel (the next population's, or mixing's): the decision stage froze what it needs
// in the chain state (GphTauFin), and the stage that proposes the next move goes out in the same launch as the decision.
// GPH_NO_FUSE=1 (tests) runs every finish as a kernel of its own, as the stepwise entry points do.
static int finish_owed(gph_engine *e)
{
  if (e->fin_owed) { e->fin_owed = false; LAUNCH(e, 5, k_tau_finish, 0); }
  return 0;
}
static int part_sweep(gph_engine *e, int32_t iteration, double ftCoal, double ftMig)
{
  int rc;
  // the three genealogy proposals run fused in one launch (GPhoCS.c:1495-1538); the deferred synchronizeEvents of the
  // previous iteration rides at its head
  const int with_sync = e->sync_pending ? 8 : 0;
  e->sync_pending = false;
  /* a mixing commit left over from the previous iteration: the host mirror of the chain state is current here (the
   * previous iteration ended with a synchronisation), so the flag and the factor travel as kernel arguments */
  int with_mix = 0;
  if (e->mix_owed) {
    if (!e->mirror_current) { int rcm = finish_sync(e); if (rcm) return rcm; }
    e->mix_owed = false;
    if (e->G_h->mix_flag) with_mix = 16;
  }
  LAUNCH(e, 0, k_sweep, 7 | with_sync | with_mix, ftCoal, ftMig, e->G_h->mix_c, e->G_h->mix_lnc);
  if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
  if ((rc = reduce_stats(e))) return rc;
  return run_stage(e, GS_SWEEP_DONE, with_sync, iteration);
}
static int part_lrate(gph_engine *e, int32_t iteration, const double *lr_alpha_finetune, int64_t *lr_accepted, double *lr_rateVar)
{
  // UpdateLocusRate, GPhoCS.c:1554-1563, 4598-4680 (host-driven: resident() is false with variable rates)
  int rc;
  GphGlobal &Gh = *e->G_h;
  gph_locus_rate_result R;
  R.accepted = 0; R.dataLogLikelihood = Gh.dataLogLikelihood; R.logLikelihood = Gh.logLikelihood; R.rateVar = *lr_rateVar;
  if ((rc = gph_engine_locus_rate_update(e, lr_alpha_finetune[1], lr_alpha_finetune[0], &R))) return rc;
  Gh.dataLogLikelihood = R.dataLogLikelihood; Gh.logLikelihood = R.logLikelihood; *lr_rateVar = R.rateVar;
  *lr_accepted += R.accepted;
  gg_rec(Gh, REC_LRATE, 0, (long long)R.accepted);
  if ((rc = push_G(e))) return rc;
  if ((rc = reduce_stats(e))) return rc;
  return run_stage(e, GS_TOTALS, 0, iteration);
}
static int part_tau(gph_engine *e, int32_t iteration, bool first_proposed)
{
  int rc;
  const GphGlobal &Gh = *e->G_h;
  const int K = Gh.K, Kc = Gh.Kc;
  const bool no_fuse = e->no_fuse;
  for (int ap = Kc; ap < K; ++ap) {
    if ((ap > Kc || !first_proposed) && (rc = run_stage(e, GS_TAU_PROPOSE, ap, iteration))) return rc;
    if (no_fuse && (rc = finish_owed(e))) return rc;
    { const int fuse = e->fin_owed; e->fin_owed = false; LAUNCH(e, 1, k_tau_eval, fuse); }
    if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
    if ((rc = run_stage(e, GS_TAU_DECIDE, ap, iteration))) return rc;
    e->fin_owed = true;
  }
  return run_stage(e, GS_TAU_END, 0, iteration);
}
static int part_sage(gph_engine *e, int32_t iteration)
{
  int rc;
  const GphGlobal &Gh = *e->G_h;
  const bool no_fuse = e->no_fuse;
  for (int pop = 0; pop < Gh.Kc; ++pop) {
    if (!Gh.updateSampleAge[pop]) continue;
    if ((rc = run_stage(e, GS_SAGE_PROPOSE, pop, iteration))) return rc;
    if (no_fuse && (rc = finish_owed(e))) return rc;
    { const int fuse = e->fin_owed; e->fin_owed = false; LAUNCH(e, 1, k_tau_eval, fuse); }
    if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
    if ((rc = run_stage(e, GS_SAGE_DECIDE, pop, iteration))) return rc;
    e->fin_owed = true;
  }
  return run_stage(e, GS_SAGE_END, 0, iteration);
}
static int part_mix(gph_engine *e, int32_t iteration)
{
  int rc;
  const GphGlobal &Gh = *e->G_h;
  const bool no_fuse = e->no_fuse;
  if ((rc = run_stage(e, GS_MIX_PROPOSE, 0, iteration))) return rc;
  if (Gh.ftMixing > 0.0) {
    if (no_fuse && (rc = finish_owed(e))) return rc;
    { const int fuse = e->fin_owed; e->fin_owed = false; LAUNCH(e, 2, k_mix_eval, fuse); }
    if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
    if ((rc = run_stage(e, GS_MIX_DECIDE, 0, iteration))) return rc;
    if (no_fuse) { LAUNCH(e, 7, k_mix_finish, 0); }
    else e->mix_owed = true;        /* at the head of the next sweep kernel (part_sweep), or mix_finish_owed() */
  }
  return 0;
}
// the genLogLikelihood refresh of every locus after sampleMigRates (GPhoCS.c:1749-1757), inside synchronizeEvents' pass
static int part_refresh(gph_engine *e, int32_t iteration)
{
  int rc;
  LAUNCH(e, 8, k_sync, 1);
  if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
  return run_stage(e, GS_REFRESH_DONE, 1, iteration);
}
// checkAll, patch.c:2745-2884: consistency checks + accumulator resynchronisation
static int part_check(gph_engine *e, int32_t iteration)
{
  int rc;
  if ((rc = flush_sync(e, false))) return rc;
  LAUNCH(e, 4, k_check, 0);
  if ((rc = reduce_local(e, 0, GPH_OUT_SLOTS))) return rc;
  if ((rc = reduce_stats(e))) return rc;
  return run_stage(e, GS_CHECK_DONE, 0, iteration);
}

// ---------------------------------------------------------------- one MCMC iteration, device-resident
// The iteration body of performMCMC (GPhoCS.c:1476-1821) as ONE stream of launches: every per-locus loop is a kernel
// over the loci, every `omp atomic` accumulation a fixed-shape reduction (+ one RCCL all-gather of the reduced row over
// several GPUs), everything the reference does on its main thread between two loops a k_global stage (gph_global.h)
// that reads the reduced rows in HBM and leaves the model, the pending proposal and the accept flag in the chain
// state for the next kernel.  The host synchronises ONCE, at the end, to read the chain state back (trace writer,
// next sweep's kernel arguments).  In host mode (caller-supplied all-reduce hook, ranks sharing a GPU, UpdateLocusRate,
// host emulation) the same launches run with a synchronisation per stage and gg_stage on the host mirror.
// lr: UpdateLocusRate's settings and running state (`locus-mut-rate VAR`), NULL = constant / fixed rates.
int gph_engine_iteration_(gph_engine *e, int32_t iteration, const double *lr_alpha_finetune, int64_t *lr_accepted, double *lr_rateVar)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  int rc;
  GphGlobal &Gh = *e->G_h;    /* settings only: the running state is on the device until finish_sync */
  PUSH_IF_DIRTY(e);
  if ((rc = part_sweep(e, iteration, Gh.ftCoalTime, Gh.ftMigTime))) return rc;
  if (lr_alpha_finetune && (rc = part_lrate(e, iteration, lr_alpha_finetune, lr_accepted, lr_rateVar))) return rc;
  if ((rc = run_stage(e, GS_THETA, 0, iteration))) return rc;
  /* the first tau proposal touches no locus: its stage rides in the launch of the stages before it, ahead of the
   * per-locus touch-ups of the accepted theta / migration-rate proposals */
  if (Gh.Kc < Gh.K && (rc = run_stage(e, GS_TAU_PROPOSE, Gh.Kc, iteration))) return rc;
  if ((rc = apply_list(e))) return rc;
  if ((rc = part_tau(e, iteration, true))) return rc;
  if ((rc = part_sage(e, iteration))) return rc;
  if (Gh.doMixing && (rc = part_mix(e, iteration))) return rc;
  if ((rc = finish_owed(e))) return rc;    /* no evaluate kernel followed the last decision */
  if ((iteration == Gh.startMig || (iteration + 1) % Gh.samplesPerLog == 0) && (rc = mix_finish_owed(e))) return rc;   /* the refresh / checkAll pass reads the pages */
  if (iteration == Gh.startMig) {
    // sampleMigRates, then the genLogLikelihood refresh of every locus (GPhoCS.c:1738-1757) inside synchronizeEvents' pass
    if ((rc = run_stage(e, GS_STARTMIG, 0, iteration))) return rc;
    if ((rc = part_refresh(e, iteration))) return rc;
  } else {
    e->sync_pending = true;   /* synchronizeEvents (GPhoCS.c:1705-1714): at the head of the next sweep kernel */
  }
  if ((iteration + 1) % Gh.samplesPerLog == 0 && (rc = part_check(e, iteration))) return rc;
  return finish_sync(e);
}

// ONE part of an iteration, its result on the host when this returns: what a caller that keeps the reference's own
// performMCMC runs behind each of the functions of GPhoCS.h:84-100 (gph_mcmc_update_* below; the caller has set the
// step size(s) of the part in the chain state and, for GPH_PART_REFRESH, the freshly sampled migration rates)
int gph_engine_part_(gph_engine *e, int32_t part, int32_t iteration, const double *lr_alpha_finetune, int64_t *lr_accepted, double *lr_rateVar)
{
  if (!e || !e->initialized) return GPH_ESTATE;
  SETDEV(e);
  int rc = 0;
  GphGlobal &Gh = *e->G_h;
  PUSH_IF_DIRTY(e);
  if (part != GPH_PART_SWEEP && part != GPH_PART_SYNC) { int rcm = mix_finish_owed(e); if (rcm) return rcm; }
  if (part == GPH_PART_LRATE || part == GPH_PART_TAU || part == GPH_PART_SAGE || part == GPH_PART_MIX) {
    int rcs = flush_sync(e, true);    /* a deferred synchronizeEvents pass must not be overtaken by a kernel that edits the pages */
    if (rcs) return rcs;
  }
  switch (part) {
  case GPH_PART_SWEEP: rc = part_sweep(e, iteration, Gh.ftCoalTime, Gh.ftMigTime); break;
  case GPH_PART_LRATE: rc = lr_alpha_finetune ? part_lrate(e, iteration, lr_alpha_finetune, lr_accepted, lr_rateVar) : GPH_EARG; break;
  case GPH_PART_THETA: rc = run_stage(e, GS_THETA_ONLY, 0, iteration); if (!rc) rc = apply_list(e); break;
  case GPH_PART_MIGR: rc = run_stage(e, GS_MIGR_ONLY, 0, iteration); if (!rc) rc = apply_list(e); break;
  case GPH_PART_TAU: rc = part_tau(e, iteration, false); break;
  case GPH_PART_SAGE: rc = part_sage(e, iteration); break;
  case GPH_PART_MIX: rc = part_mix(e, iteration); break;
  case GPH_PART_SYNC: e->sync_pending = true; break;
  case GPH_PART_REFRESH: e->sync_pending = false; rc = part_refresh(e, iteration); break;
  case GPH_PART_CHECK: Gh.nrec = 0; e->G_dirty = true; PUSH_IF_DIRTY(e); rc = part_check(e, iteration); break;
  default: return GPH_EARG;
  }
  if (rc) return rc;
  if ((rc = finish_owed(e))) return rc;
  if (part != GPH_PART_MIX && (rc = mix_finish_owed(e))) return rc;    /* (after GPH_PART_MIX the commit may wait for GPH_PART_SWEEP) */
  return finish_sync(e);
}

// kernel-level fixtures (gph_kernels.h: kb_unit): single calls of the per-locus functions on the current chain state,
// every call undone.  out: [local loci][stride] doubles in input order, stride >= 3 (n - 1).
//   op 0: per internal node tnew, lnLd, dprior;  op 1: full recompute value;  op 2: rubber band (pre) of ancestral
//   population `arg`: delta, n0, n1, lik -- the proposal (bounds, factors) is derived from the model exactly as
//   oracle/ref_harness.c `unit` derives it;  ops 3-6 (ref_harness.c `unit2`): executeGenSPR of node arg onto a fixed list
//   of branches (stride >= 1 + 5 N: calls, then target, age, return code, value, root per call), scaleAllNodeAges by
//   1 + arg / 1000 + revert + full recompute (delta, value), rubberBandRipple do / undo (moved events, two deltas),
//   traceLineage(arg, 0 / 1) + evaluation (stride >= 13: 1, res, target, father's new population, old / new migration
//   events, both prior deltas, father's new age, data delta, generator state);  op 7 (`unit2` H): rubberBandRipple do / undo over
//   the MIG_BAND_START / MIG_BAND_END events of every band (UpdateTau's start_or_end list): moved events, two deltas
int gph_engine_unit(gph_engine *e, int32_t op, int32_t arg, double *out, int32_t stride)
{
  if (!e || !e->initialized || !out || op < 0 || op > 8 || stride < 3 * (e->cfg.n - 1) || stride < 4) return GPH_EARG;
  if ((op == 3 && stride < 1 + 5 * (2 * e->cfg.n - 1)) || (op == 6 && stride < 13)) return GPH_EARG;
  if ((op == 3 || op == 6) && (arg < 0 || arg >= 2 * e->cfg.n - 1)) return GPH_EARG;
  if (op == 8 && (arg < 0 || arg > 100000)) return GPH_EARG;      /* (the checked build's self-test: arg = a node index, possibly out of range) */
  SETDEV(e);
  { int rcs = flush_sync(e, true); if (!rcs) rcs = finish_sync(e); if (rcs) return rcs; }
  if (op == 2) {
    if (arg < e->cfg.Kc || arg >= e->cfg.K) return GPH_EARG;
    GphGlobal &G = *e->G_h;
    const GphModel &M = G.model;
    GphTauArgs &A = G.tau;
    memset(&A, 0, sizeof A);
    const int ap = arg, s0 = M.popSon0[ap], s1 = M.popSon1[ap], isRoot = ap == e->cfg.rootPop;
    const double tauold = M.popAge[ap];
    double taub0 = gg_max2(M.popAge[s0], M.popAge[s1]);
    taub0 = gg_max2(taub0, M.sampleAge[s0]);
    taub0 = gg_max2(taub0, M.sampleAge[s1]);
    const double taub1 = isRoot ? GG_OLDAGE : M.popAge[M.popFather[ap]];
    const double taunew = taub0 + 0.55 * (gg_min2(taub1, tauold * 1.4) - taub0);
    A.ap = ap; A.son0 = s0; A.son1 = s1; A.isRoot = isRoot;
    A.tauold = tauold; A.taunew = taunew; A.taub0 = taub0; A.taub1 = taub1;
    A.taufactor0 = (taunew - taub0) / (tauold - taub0);
    A.taufactor1 = isRoot ? A.taufactor0 : (taunew - taub1) / (tauold - taub1);
    e->G_dirty = true;
  }
  PUSH_IF_DIRTY(e);
  double *d_out = nullptr;
  const size_t bytes = sizeof(double) * (size_t)stride * e->L;
  if (dev_alloc((void **)&d_out, bytes)) return GPH_EHIP;
  int rc = 0;
#ifdef GPH_HOSTEMU
  memset(d_out, 0, bytes);
#else
  if (hipMemsetAsync(d_out, 0, bytes, e->stream) != hipSuccess) rc = GPH_EHIP;
#endif
  if (!rc) {
    auto launch = [&]() -> int { LAUNCH(e, 12, k_unit, (int)op, (int)arg, d_out, (int)stride); return 0; };
    rc = launch();
  }
  if (!rc) rc = reduce_local(e, 0, GPH_OUT_SLOTS);
  if (!rc) rc = run_stage_now(e, GS_COUNT_ONLY, 12);
  if (!rc) rc = d2h(e, out, d_out, bytes);
  dev_free(d_out);
  return rc;
}

// the last fatal error a call returned GPH_EKERNEL for: the reference's "Fatal Error NNNN" code and the global index of the
// first locus that reported it (-1: not attributable to one locus, e.g. checkAll's accumulator test); 0 / -1 if none
int gph_engine_last_error(gph_engine *e, int64_t *locus, int32_t *code)
{
  if (!e) return GPH_EARG;
  if (locus) *locus = e->last_error_locus;
  if (code) *code = e->last_error_code;
  return 0;
}

// tests only: break the event chain of population `pop` of one locus (the `next` link of its first event becomes -1), so
// that the next kernel's bounded chain walks raise a fatal code for exactly that locus (tests/test_host_logic.py,
// tests/test_gpu_parity.py: the failure must name the locus and print its genealogy)
int gph_engine_debug_break_chain(gph_engine *e, int64_t global_locus, int32_t pop)
{
  if (!e || !e->initialized || pop < 0 || pop >= e->cfg.K) return GPH_EARG;
  SETDEV(e);
  /* (several ranks: EVERY rank calls this -- a deferred synchronizeEvents pass runs here, with its exchange; the rank that
   * does not hold the locus returns GPH_EARG after it) */
  { int rcs = flush_sync(e, true); if (!rcs) rcs = finish_sync(e); if (rcs) return rcs; }
  const long long lo = global_locus - e->cfg.locus_begin;
  if (lo < 0 || lo >= e->L) return GPH_EARG;
  int64_t slot = -1;
  for (int64_t j = 0; j < e->L; j++) if (e->h_orig[j] == lo) { slot = j; break; }
  if (slot < 0) return GPH_ESTATE;
  const GphLayout &y = e->lay;
  std::vector<char> pg(y.page_bytes);
  char *dp = (char *)e->dev.pages + (size_t)slot * y.page_bytes;
  int rc = d2h(e, pg.data(), dp, y.page_bytes);
  if (rc) return rc;
  const int16_t *first = (const int16_t *)(pg.data() + y.o_first);
  GphEv *evr = (GphEv *)(pg.data() + y.o_ev);
  if (first[pop] < 0) return GPH_ESTATE;
  evr[first[pop]].next = -1;
  return h2d(e, dp, pg.data(), y.page_bytes);
}

// tests only: the CHECKED build's first out-of-range index (gph_rt.h: GPH_BOUNDS): source line + 100000 x file (1 gph_locus.h,
// 2 gph_kernels.h), 8000xx the typed accessors of the image, 9000xx those of the dynamic LDS; 0 = none so far.  *checked = 0 in
// a build without the checks (every product library).
int gph_engine_debug_oob(gph_engine *e, int32_t *where, int32_t *checked)
{
  if (!e || !where || !checked) return GPH_EARG;
  *where = 0;
#ifndef GPH_BOUNDS
  *checked = 0;
  return 0;
#else
  *checked = 1;
  SETDEV(e);
  if (e->initialized) { int rcs = flush_sync(e, true); if (!rcs) rcs = finish_sync(e); if (rcs) return rcs; }
  /* read and clear: the word belongs to the library (every engine of the process), a report must not be seen twice */
#ifdef GPH_HOSTEMU
  *where = gph_oob_word;
  gph_oob_word = 0;
#else
  int w = 0, z = 0;
  HIPCHK(hipMemcpyFromSymbol(&w, HIP_SYMBOL(gph_oob_word), sizeof(int)));
  if (w != 0) HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(gph_oob_word), &z, sizeof(int)));
  *where = w;
#endif
  return 0;
#endif
}

#if defined(GPH_HOSTEMU) && defined(GPH_EMU64)
// tests only (the wave64 host build): micro-waves started and rendezvous passed by the calling thread so far
int gph_debug_emu64_stats(long long *waves, long long *rendezvous)
{
  if (waves) *waves = gph_emu::tw.waves_total;
  if (rendezvous) *rendezvous = gph_emu::tw.rendezvous_total;
  return gph_emu::g_enabled;
}
#endif

// canonical text dump (same format as oracle/gphocs_oracle_io.c go_dump_state's per-locus part)
int gph_engine_dump_loci(gph_engine *e, const char *path, int32_t withCond, int32_t append)
{
  if (!e || !e->initialized || !path) return GPH_ESTATE;
  SETDEV(e);
  { int rcs = flush_sync(e, true); if (!rcs) rcs = finish_sync(e); if (rcs) return rcs; }
  const GphLayout &y = e->lay;
  std::vector<char> pages(e->pages_bytes), cond(withCond ? e->cond_bytes : 0);
  int rc = d2h(e, pages.data(), e->dev.pages, e->pages_bytes);
  if (!rc && withCond) rc = d2h(e, cond.data(), e->dev.cond, e->cond_bytes);
  if (rc) return rc;
  FILE *f = fopen(path, append ? "a" : "w");
  if (!f) return GPH_EARG;
  std::vector<int32_t> slot_of(e->L);
  for (int64_t j = 0; j < e->L; j++) slot_of[e->h_orig[j]] = (int32_t)j;
  /* GPH_DUMP_STRIDE=k: every k-th locus (global index) only -- full-size parity runs */
  const int dstride = getenv("GPH_DUMP_STRIDE") && atoi(getenv("GPH_DUMP_STRIDE")) > 0 ? atoi(getenv("GPH_DUMP_STRIDE")) : 1;
  for (int64_t go = 0; go < e->L; go++) {
    if ((go + e->cfg.locus_begin) % dstride != 0) continue;
    const int64_t g = slot_of[go];
    dump_one_locus(e, f, go + e->cfg.locus_begin, pages.data() + (size_t)g * y.page_bytes, withCond ? cond.data() + e->h_cond_off[g] : nullptr, e->h_P[g]);
  }
  fclose(f);
  return 0;
}

} // extern "C"
