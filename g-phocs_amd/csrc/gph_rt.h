// gph_rt.h -- execution-model shim.
//
// Product build (hipcc, gfx950): one 64-lane wavefront per locus, the locus
// staged in LDS (`gph_sm`, dynamic shared memory of a single-wave workgroup),
// layout/model tables in __constant__ memory so every table access is a scalar
// load.  All chain/tree logic is wave-uniform (every lane executes it on the
// same LDS words); integer loads that steer control flow go through
// v_readfirstlane so branches are scalar.  Only the pruning inner loops and the
// stage-in/out copies are lane-parallel.
//
// GPH_HOSTEMU build (g++, tests/hostemu only): the same per-locus code compiled
// for the host with a 1-lane "wave" so it can be run under sanitizers/gdb in the
// GPU-less build container.  It is a debugging aid for tests; it is NOT linked
// into libgphocs_hip.so and there is no CPU fallback in the product.
#pragma once
#include <stdint.h>
#include <math.h>
#include "gph_types.h"
#include "gph_math.h"

#ifdef GPH_HOSTEMU
#include <string.h>
#define GPH_DEV inline
#define GPH_DEVNI static
#define GPH_DEVHOT inline
#define GPH_LDS
#ifdef GPH_EMU64
// the host build with a 64-lane MICRO-WAVE for the device forms of the lane-parallel functions (gph_emu64.h): outside a
// micro-wave everything below is the one-lane host build (lane 0 of 1, rendezvous = nothing)
#include "gph_emu64.h"
#define GPH_LANE (gph_emu::lane())
#define GPH_NLANES (gph_emu::nlanes())
#define GPH_SYNC() gph_emu::rendezvous(__LINE__)
#define GPH_WAVE_FENCE() gph_emu::rendezvous(__LINE__)
// a store to the LDS image through an accessor: everybody has computed its value | everybody has stored
#define GPH_EMU_ST(k) gph_emu::rendezvous(800 + (k))
// the device builtins the lane-parallel forms are written in
#define __ballot(p) gph_emu::ballot((p), __LINE__)
#define __builtin_amdgcn_readlane(v, l) gph_emu::readlane32((int)(v), (l), __LINE__)
#define __builtin_amdgcn_ds_bpermute(a, v) gph_emu::bpermute((a), (int)(v), __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) (static_cast<void>(sizeof(char[(ctrl) == 0x130 ? 1 : -1])), gph_emu::dpp_wave_shl1((src), __LINE__))
#else
#define GPH_LANE 0
#define GPH_NLANES 1
#define GPH_SYNC() ((void)0)
#define GPH_WAVE_FENCE() ((void)0)
#define GPH_EMU_ST(k) ((void)0)
#endif
#define RFL(x) (x)
extern thread_local char *gph_sm;
extern thread_local GphLds gph_lds;
extern GphLayout g_lay;
extern GphModel g_model;
extern GphGlobal *gph_G_emu;
typedef const GphGlobal gph_cglobal;
typedef const GphTauArgs gph_ctau;
typedef const GphTauFin gph_cfin;
#define GPH_G ((gph_cglobal *)gph_G_emu)
#else
#include <hip/hip_runtime.h>
#define GPH_DEV __device__ inline
#define GPH_DEVNI __device__ __noinline__
// hot-path functions are inlined into the kernels: an out-of-line call costs ~150 instructions of
// callee-saved SGPR/VGPR save+restore (v_writelane/scratch) on a path that is instruction-bound
#ifndef GPH_DEVHOT
#define GPH_DEVHOT __device__ __attribute__((always_inline)) inline
#endif
#define GPH_LDS __attribute__((address_space(3)))
// lane id recomputed where it is used (2 VALU) instead of threadIdx.x kept alive in a VGPR through the whole
// kernel (the register allocator spilled it to scratch: a memory round trip per use); workgroup = one wave
#define GPH_LANE ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define GPH_NLANES GPH_WAVE
#define GPH_SYNC() __syncthreads()
// hand-off between lanes of the SAME wavefront (the only kind there is: one wave per workgroup).
// LDS and vector-memory operations of one wave are issued and performed in order and the CU's L1
// is write-through, so no wait for completion is needed -- only a compiler barrier (a
// wavefront-scope fence emits no instruction).  Used in the pruning loop, where waiting for each
// node's store to be acknowledged (what __syncthreads' workgroup fence does) cost ~25 % of lik_compute.
#define GPH_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define GPH_EMU_ST(k) ((void)0)
#define RFL(x) __builtin_amdgcn_readfirstlane((int)(x))
extern __shared__ __attribute__((aligned(16))) char gph_sm[];   // dynamic part: sequence block + per-pattern terms
__shared__ GphLds gph_lds;   // static part: the locus image (gph_types.h); this header is included by one TU only
typedef __attribute__((address_space(4))) const GphKargs gph_ckargs;
typedef __attribute__((address_space(4))) const GphModel gph_cmodel;
typedef __attribute__((address_space(4))) const GphGlobal gph_cglobal;
typedef __attribute__((address_space(4))) const GphTauArgs gph_ctau;
typedef __attribute__((address_space(4))) const GphTauFin gph_cfin;
// the model tables: GphCtxT<false> (the genealogy sweep) reads the copy in the kernel-argument segment, GphCtxT<true>
// (every kernel that runs after a decision the host has not seen) the chain state in HBM, through the pointer in the
// kernel-argument segment and the constant address space -- scalar loads either way
#define g_model (this->gmodel())
#define g_lay (((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->lay)
#define GPH_G ((gph_cglobal *)(((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->G))
#endif

// ---- GPH_BOUNDS (a CHECKED build: tests only, libgphocs_hip_chk.so and the sanitizer host builds): every index the per-locus code
// puts into an array of the LDS image, into the dynamic LDS or into the locus's conditional arrays is compared with the array's
// extent; the first violation leaves its source line (+ 100000 x file: 1 gph_locus.h, 2 gph_kernels.h; 900000 + k: the typed
// accessors of the dynamic part) in a device word the host reads back (gph_engine_debug_oob), and the access goes to element 0
// instead.  What AddressSanitizer does for the host forms, for the DEVICE forms on the MI355X (no GPU sanitizer on this pool).
// The product build compiles GPH_IX(i, n) to i.
#ifdef GPH_BOUNDS
#ifdef GPH_HOSTEMU
extern int gph_oob_word;
inline int gph_ix_(int i, int n, int where) { if ((unsigned)i >= (unsigned)n) { if (gph_oob_word == 0) gph_oob_word = where; return 0; } return i; }
#else
__device__ int gph_oob_word;
__device__ inline int gph_ix_(int i, int n, int where) { if ((unsigned)i >= (unsigned)n) { atomicCAS(&gph_oob_word, 0, where); return 0; } return i; }
#endif
#define GPH_IX(i, n) gph_ix_((i), (n), __LINE__ + 100000 * GPH_FILE_ID)
#define GPH_IXW(i, n, w) gph_ix_((i), (n), (w))
#else
#define GPH_IX(i, n) (i)
#define GPH_IXW(i, n, w) (i)
#endif
#define GPH_FILE_ID 0

#ifdef GPH_HOSTEMU
#define GPH_GLB
#else
#define GPH_GLB __attribute__((address_space(1)))
#endif
typedef GPH_GLB double gdbl;   // conditional-likelihood arrays live in global memory (L2 / Infinity Cache)
// wave-uniform helpers: every lane holds the same value; moving it through SGPRs lets the
// compiler use scalar loads for table look-ups and scalar branches for control flow
#ifdef GPH_HOSTEMU
#define RFLD(x) (x)
#define UNI(c) (c)
#else
__device__ inline double gph_rfl64(double x)
{
  union { double d; int32_t i[2]; } v;
  v.d = x;
  v.i[0] = __builtin_amdgcn_readfirstlane(v.i[0]);
  v.i[1] = __builtin_amdgcn_readfirstlane(v.i[1]);
  return v.d;
}
#define RFLD(x) gph_rfl64(x)
// a wave-uniform condition computed on the vector ALU (fp64 compares): the compare's lane mask is tested
// as a scalar (v_cmp -> SGPR pair -> s_cmp_lg_u64 -> s_cbranch_scc) instead of being turned into a 0/1
// VGPR and read back (v_cndmask + v_readfirstlane + s_bitcmp)
#define UNI(c) (__builtin_amdgcn_ballot_w64((bool)(c)) != 0)
#endif
// ---- the lane primitives the per-locus code is written in, ONE form for both builds (the host build has one lane)
// GPH_EACH(k, n): items 0..n-1 dealt round-robin to the lanes, any n.  GPH_EACH1(k, n): n <= 64, item k on lane k
// (no loop on the device).  Items must be independent: the host runs them one after the other.
#define GPH_EACH(k, n) for (int k = GPH_LANE; k < (n); k += GPH_NLANES)
#ifdef GPH_HOSTEMU
#define GPH_EACH1(k, n) for (int k = 0; k < (n); k++)
#else
#define GPH_EACH1(k, n) if (const int k = GPH_LANE; k < (n))
#endif
// the value lane `i` holds of a per-lane variable (the host's single lane holds item i when it asks for it)
#if defined(GPH_HOSTEMU) && defined(GPH_EMU64)
#define GPH_LANEVAL32(v, i) (v)
#define GPH_LANEVAL64(v, i) (v)
#define gph_readlane64(v, l) gph_emu::readlane64((v), (l), __LINE__)
#define gph_bcast64(v, l) gph_emu::readlane64((v), (l), __LINE__)
#define gph_bcast32(v, l) gph_emu::readlane32((v), (l), __LINE__)
#elif defined(GPH_HOSTEMU)
#define GPH_LANEVAL32(v, i) (v)
#define GPH_LANEVAL64(v, i) (v)
#else
__device__ inline double gph_readlane64(double v, int l)
{
  union { double d; int32_t i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
  return u.d;
}
// lane l's value in every lane, through the LDS crossbar (l wave-uniform)
__device__ inline double gph_bcast64(double v, int l)
{
  union { double d; int32_t i[2]; } u;
  u.d = v;
  const int a = l << 2;
  u.i[0] = __builtin_amdgcn_ds_bpermute(a, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(a, u.i[1]);
  return u.d;
}
__device__ inline int gph_bcast32(int v, int l) { return __builtin_amdgcn_ds_bpermute(l << 2, v); }
#define GPH_LANEVAL32(v, i) __builtin_amdgcn_readlane((int)(v), (i))
#define GPH_LANEVAL64(v, i) gph_readlane64((v), (i))
#endif
// a 16-byte record (GphNode, GphEv) of the LDS image with one access (ds_read_b128)
struct gph_w4 { uint32_t x, y, z, w; };
template <class T> GPH_DEVHOT gph_w4 gph_ld16(const T *p)   /* p points into the LDS image */
{
  static_assert(sizeof(T) == 16, "16-byte records only");
  gph_w4 r;
#ifdef GPH_HOSTEMU
  memcpy(&r, p, 16);
#else
  typedef uint32_t gph_u4 __attribute__((ext_vector_type(4)));
  const gph_u4 w = *(const GPH_LDS gph_u4 *)p;
  r.x = w.x; r.y = w.y; r.z = w.z; r.w = w.w;
#endif
  return r;
}
// coalesced 16-byte copies between HBM and the LDS image, lanes = consecutive 16-byte words
#define GPH_LDSP(p) ((GPH_LDS void *)(p))      /* the address of a member of the static LDS image, as an LDS pointer */
#ifdef GPH_HOSTEMU
GPH_DEV void gph_copy16_in(void *lds, const void *glb, int n16) { memcpy(lds, glb, (size_t)n16 << 4); }
GPH_DEV void gph_copy16_out(void *glb, const void *lds, int n16) { memcpy(glb, lds, (size_t)n16 << 4); }
#else
typedef uint32_t gph_u32x4 __attribute__((ext_vector_type(4)));
GPH_DEV void gph_copy16_in(GPH_LDS void *lds, const void *glb, int n16)
{
  GPH_EACH(i, n16) ((GPH_LDS gph_u32x4 *)lds)[i] = ((const gph_u32x4 *)glb)[i];
}
GPH_DEV void gph_copy16_out(void *glb, const GPH_LDS void *lds, int n16)
{
  GPH_EACH(i, n16) ((gph_u32x4 *)glb)[i] = ((const GPH_LDS gph_u32x4 *)lds)[i];
}
#endif
// n16 <= 64: one word per lane, no loop
#ifdef GPH_HOSTEMU
GPH_DEV void gph_copy16_in1(void *lds, const void *glb, int n16) { memcpy(lds, glb, (size_t)n16 << 4); }
#else
GPH_DEV void gph_copy16_in1(GPH_LDS void *lds, const void *glb, int n16)
{
  GPH_EACH1(i, n16) ((GPH_LDS gph_u32x4 *)lds)[i] = ((const gph_u32x4 *)glb)[i];
}
#endif
// two such copies with every load of the first ACH / BCH kilobytes ISSUED before the first one is waited for (a
// copy loop -- load 1 KB, wait, write LDS, repeat -- serialises one memory round trip per kilobyte at the head of a
// wavefront); what lies beyond BCH KB of the second block (pattern-rich loci) follows in a loop
#ifdef GPH_HOSTEMU
template <int ACH, int BCH> GPH_DEV void gph_copy16_in2(void *la, const void *ga, int na, void *lb, const void *gb, int nb)
{
  memcpy(la, ga, (size_t)na << 4);
  memcpy(lb, gb, (size_t)nb << 4);
}
#else
template <int ACH, int BCH> GPH_DEV void gph_copy16_in2(GPH_LDS void *la, const void *ga, int na, GPH_LDS void *lb, const void *gb, int nb)
{
  const int lane = GPH_LANE;
  const gph_u32x4 *ps = (const gph_u32x4 *)ga, *ss = (const gph_u32x4 *)gb;
  gph_u32x4 pr[ACH], sr[BCH];
#pragma unroll
  for (int k = 0; k < ACH; k++) if (lane + 64 * k < na) pr[k] = ps[lane + 64 * k];
#pragma unroll
  for (int k = 0; k < BCH; k++) if (lane + 64 * k < nb) sr[k] = ss[lane + 64 * k];
  GPH_LDS gph_u32x4 *pd = (GPH_LDS gph_u32x4 *)la, *sd = (GPH_LDS gph_u32x4 *)lb;
#pragma unroll
  for (int k = 0; k < ACH; k++) if (lane + 64 * k < na) pd[lane + 64 * k] = pr[k];
#pragma unroll
  for (int k = 0; k < BCH; k++) if (lane + 64 * k < nb) sd[lane + 64 * k] = sr[k];
  for (int i = lane + 64 * BCH; i < nb; i += GPH_NLANES) sd[i] = ss[i];
}
#endif
// the wave-uniform integer scalars of a locus: lane i of ONE vector register holds scalar i (a read is a v_readlane
// with a constant lane, a write a v_writelane); the host keeps an array
#ifdef GPH_HOSTEMU
template <int NN> struct GphPad {
  int32_t v[NN] = {};
  int get(int i) const { return v[i]; }
  void set(int i, int x) { v[i] = x; }
  void load(const int32_t *src, int n) { for (int k = 0; k < NN; k++) v[k] = k < n ? src[k] : 0; }
  void store(int32_t *dst, int n) const { for (int k = 0; k < n; k++) dst[k] = v[k]; }
};
#define GPH_PADGET(i) (r_pad.get(i))
#ifdef GPH_EMU64
#define GPH_PADSET(i, x) do { const int pv_ = (x); GPH_EMU_ST(1); r_pad.set((i), pv_); GPH_EMU_ST(2); } while (0)
#else
#define GPH_PADSET(i, x) (r_pad.set((i), (x)))
#endif
#else
// (up to 64 scalars: one register; the 200-leaf build's node sets need more: scalars 64.. sit in a second register, which a
// build that never indexes beyond 63 never materialises)
template <int NN> struct GphPad {
  static_assert(NN <= 128, "one scalar per lane of two registers");
  int32_t v = 0, v2 = 0;
  __device__ inline void load(const int32_t *src, int n)
  {
    const int lane = GPH_LANE;
    v = lane < n ? ((const GPH_LDS int32_t *)src)[lane] : 0;
    if constexpr (NN > 64) v2 = lane + 64 < n ? ((const GPH_LDS int32_t *)src)[lane + 64] : 0;
  }
  __device__ inline void store(int32_t *dst, int n) const
  {
    const int lane = GPH_LANE;
    if (lane < n) ((GPH_LDS int32_t *)dst)[lane] = v;
    if constexpr (NN > 64) { if (lane + 64 < n) ((GPH_LDS int32_t *)dst)[lane + 64] = v2; }
  }
};
#define GPH_PADGET(i) ((i) < 64 ? __builtin_amdgcn_readlane(r_pad.v, (i) & 63) : __builtin_amdgcn_readlane(r_pad.v2, (i) & 63))
#define GPH_PADSET(i, x) do { const int pv_ = RFL(x); \
    if ((i) < 64) asm("v_writelane_b32 %0, %1, %2" : "+v"(r_pad.v) : "s"(pv_), "i"((i) & 63)); \
    else asm("v_writelane_b32 %0, %1, %2" : "+v"(r_pad.v2) : "s"(pv_), "i"((i) & 63)); } while (0)
#endif
// a / b given y = RN(1/b).  Device: q0 = a*y, r = fma(-q0, b, a) (exact), q = fma(r, y, q0) is the correctly rounded
// quotient (Markstein) without the ~12-instruction divide expansion; tools/verify_fma_div*.c compare it with the
// hardware division on 1.4e9 operands, all-ones significands of b included.  The host divides -- which also makes
// every golden-trace test of the host build a check of that equivalence.
GPH_DEV double gph_quot(double a, double b, double y)
{
#ifdef GPH_HOSTEMU
  (void)y;
  return a / b;
#else
  const double q0 = a * y;
  return __builtin_fma(__builtin_fma(-q0, b, a), y, q0);
#endif
}
// first error of a locus into the launch's error word
#ifdef GPH_HOSTEMU
GPH_DEV void gph_raise(int *err, int code) { if (*err == 0) *err = code; }
#else
GPH_DEV void gph_raise(int *err, int code) { atomicMax(err, code); }
#endif

typedef GPH_LDS double lf64;
typedef GPH_LDS int16_t li16;
typedef GPH_LDS int32_t li32;
typedef GPH_LDS uint32_t lu32;
typedef GPH_LDS uint8_t lu8;
typedef GPH_LDS char lchar;

#define GPH_SMB ((lchar *)gph_sm)
// static-image accessors: `m` is a pointer to an array member of GphLds; after inlining the
// address is a compile-time constant + index.  Integer loads that steer control flow are
// made wave-uniform (v_readfirstlane) so that branches are scalar.
template <class T, int NN> GPH_DEV double gf64(T (GphLds::*m)[NN], int i) { return (gph_lds.*m)[GPH_IXW(i, NN, 800001)]; }
template <class T, int NN> GPH_DEV void sf64(T (GphLds::*m)[NN], int i, double v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(i, NN, 800002)] = v; GPH_EMU_ST(4); }
template <class T, int NN> GPH_DEV int gi16(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[GPH_IXW(i, NN, 800003)]); }
template <class T, int NN> GPH_DEV void si16(T (GphLds::*m)[NN], int i, int v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(i, NN, 800004)] = (T)v; GPH_EMU_ST(4); }
template <class T, int NN> GPH_DEV int gi32(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[GPH_IXW(i, NN, 800005)]); }
template <class T, int NN> GPH_DEV void si32(T (GphLds::*m)[NN], int i, int v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(i, NN, 800006)] = (T)v; GPH_EMU_ST(4); }
template <class T, int NN> GPH_DEV int gu8(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[GPH_IXW(i, NN, 800007)]); }
template <class T, int NN> GPH_DEV void su8(T (GphLds::*m)[NN], int i, int v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(i, NN, 800008)] = (T)v; GPH_EMU_ST(4); }
template <class T, int NN> GPH_DEV int gu8v(T (GphLds::*m)[NN], int i) { return (gph_lds.*m)[GPH_IXW(i, NN, 800009)]; }
// two-dimensional scratch arrays [2][..] (instance 0/1 of the pending stat deltas)
template <class T, int NN> GPH_DEV double gf64(T (GphLds::*m)[2][NN], int k, int i) { return (gph_lds.*m)[GPH_IXW(k, 2, 800010)][GPH_IXW(i, NN, 800010)]; }
template <class T, int NN> GPH_DEV void sf64(T (GphLds::*m)[2][NN], int k, int i, double v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(k, 2, 800011)][GPH_IXW(i, NN, 800011)] = v; GPH_EMU_ST(4); }
template <class T, int NN> GPH_DEV int gi16(T (GphLds::*m)[2][NN], int k, int i) { return RFL((gph_lds.*m)[GPH_IXW(k, 2, 800012)][GPH_IXW(i, NN, 800012)]); }
template <class T, int NN> GPH_DEV void si16(T (GphLds::*m)[2][NN], int k, int i, int v) { GPH_EMU_ST(3); (gph_lds.*m)[GPH_IXW(k, 2, 800013)][GPH_IXW(i, NN, 800013)] = (T)v; GPH_EMU_ST(4); }
// dynamic-part accessors (byte offset GPH_Q_* of the locus' sequence block): lane-varying, pruning only
// (checked build: the byte range of the access against the dynamic LDS of this launch, g_lay.dyn_bytes)
#define GPH_DYN(off, i, sz, w) GPH_IXW(((off) + (i) * (sz)) / (sz), (g_lay.dyn_bytes > 0 ? g_lay.dyn_bytes : 1 << 30) / (sz), 900000 + (w)) * (sz)
#ifdef GPH_BOUNDS
GPH_DEV double gf64(int off, int i) { return *(lf64 *)(GPH_SMB + GPH_DYN(off, i, 8, 1)); }
GPH_DEV void sf64(int off, int i, double v) { GPH_EMU_ST(5); *(lf64 *)(GPH_SMB + GPH_DYN(off, i, 8, 2)) = v; GPH_EMU_ST(6); }
#else
GPH_DEV double gf64(int off, int i) { return ((lf64 *)(GPH_SMB + off))[i]; }
GPH_DEV void sf64(int off, int i, double v) { GPH_EMU_ST(5); ((lf64 *)(GPH_SMB + off))[i] = v; GPH_EMU_ST(6); }
#endif
GPH_DEV int gi16(int off, int i) { return RFL(((li16 *)(GPH_SMB + off))[i]); }
GPH_DEV void si16(int off, int i, int v) { GPH_EMU_ST(5); ((li16 *)(GPH_SMB + off))[i] = (int16_t)v; GPH_EMU_ST(6); }
GPH_DEV int gi32(int off, int i) { return RFL(((li32 *)(GPH_SMB + off))[i]); }
GPH_DEV void si32(int off, int i, int v) { GPH_EMU_ST(5); ((li32 *)(GPH_SMB + off))[i] = v; GPH_EMU_ST(6); }
GPH_DEV int gu8(int off, int i) { return RFL(((lu8 *)(GPH_SMB + off))[i]); }
GPH_DEV void su8(int off, int i, int v) { GPH_EMU_ST(5); ((lu8 *)(GPH_SMB + off))[i] = (uint8_t)v; GPH_EMU_ST(6); }
// lane-varying byte/int loads (no readfirstlane): pruning only
#ifdef GPH_BOUNDS
GPH_DEV int gu8v(int off, int i) { return *(lu8 *)(GPH_SMB + GPH_DYN(off, i, 1, 3)); }
GPH_DEV int gu16v(int off, int i) { return *(GPH_LDS uint16_t *)(GPH_SMB + GPH_DYN(off, i, 2, 4)); }
#else
GPH_DEV int gu8v(int off, int i) { return ((lu8 *)(GPH_SMB + off))[i]; }
GPH_DEV int gu16v(int off, int i) { return ((GPH_LDS uint16_t *)(GPH_SMB + off))[i]; }
#endif
GPH_DEV int gu16(int off, int i) { return RFL(((GPH_LDS uint16_t *)(GPH_SMB + off))[i]); }
GPH_DEV int gi32v(int off, int i) { return ((li32 *)(GPH_SMB + off))[i]; }
// ---- the sequence block seen by the GENERIC (pattern, base) code paths (loci with more than GPH_WAVE phased patterns, and the
// one-lane host build): in the wave's dynamic LDS like everybody's, or -- a locus whose block outgrows the launch group's
// LDS budget ("huge": more than g_lay.huge_P patterns; the reference mallocs any P, LocusDataLikelihood.c:251) -- where it
// lies in HBM, 8 bytes per pattern for the root reduction's terms behind it.  g == null: LDS.  The lane-per-pattern paths
// (P <= GPH_WAVE: every hot loop) never come here: such a block always fits.
struct GphSeq { GPH_GLB char *g; };
GPH_DEV int sq_u8v(const GphSeq &S, int off, int i) { return S.g ? (int)((const GPH_GLB uint8_t *)(S.g + off))[i] : gu8v(off, i); }
GPH_DEV int sq_u16v(const GphSeq &S, int off, int i) { return S.g ? (int)((const GPH_GLB uint16_t *)(S.g + off))[i] : gu16v(off, i); }
GPH_DEV int sq_i32v(const GphSeq &S, int off, int i) { return S.g ? ((const GPH_GLB int32_t *)(S.g + off))[i] : gi32v(off, i); }
GPH_DEV double sq_f64(const GphSeq &S, int off, int i) { return S.g ? ((const GPH_GLB double *)(S.g + off))[i] : gf64(off, i); }
GPH_DEV void sq_sf64(const GphSeq &S, int off, int i, double v) { if (S.g) ((GPH_GLB double *)(S.g + off))[i] = v; else sf64(off, i, v); }
#define GPH_LEAFCODE_S(S, q_leaf, p, child) ((sq_u8v((S), (q_leaf), (p) * GPH_Q_NH(g_lay.n) + ((child) >> 1)) >> (((child) & 1) << 2)) & 15)
#define GPH_PATCOUNT_S(S, q_count, p) (g_lay.cnt16 ? sq_u16v((S), (q_count), (p)) : sq_i32v((S), (q_count), (p)))
// leaf code (4 bits) of leaf `child` for pattern p, and pattern p's count (16 or 32 bits): gph_types.h, sequence block
#define GPH_LEAFCODE(q_leaf, p, child) ((gu8v((q_leaf), (p) * GPH_Q_NH(g_lay.n) + ((child) >> 1)) >> (((child) & 1) << 2)) & 15)
#define GPH_PATCOUNT(q_count, p) (g_lay.cnt16 ? gu16v((q_count), (p)) : gi32v((q_count), (p)))
