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
#define GPH_LANE 0
#define GPH_NLANES 1
#define GPH_SYNC() ((void)0)
#define GPH_WAVE_FENCE() ((void)0)
#define RFL(x) (x)
extern thread_local char *gph_sm;
extern thread_local GphLds gph_lds;
extern GphLayout g_lay;
extern GphModel g_model;
extern GphGlobal *gph_G_emu;
typedef const GphGlobal gph_cglobal;
typedef const GphTauArgs gph_ctau;
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
#define RFL(x) __builtin_amdgcn_readfirstlane((int)(x))
extern __shared__ __attribute__((aligned(16))) char gph_sm[];   // dynamic part: sequence block + per-pattern terms
__shared__ GphLds gph_lds;   // static part: the locus image (gph_types.h); this header is included by one TU only
typedef __attribute__((address_space(4))) const GphKargs gph_ckargs;
typedef __attribute__((address_space(4))) const GphModel gph_cmodel;
typedef __attribute__((address_space(4))) const GphGlobal gph_cglobal;
typedef __attribute__((address_space(4))) const GphTauArgs gph_ctau;
// the model tables: GphCtxT<false> (the genealogy sweep) reads the copy in the kernel-argument segment, GphCtxT<true>
// (every kernel that runs after a decision the host has not seen) the chain state in HBM, through the pointer in the
// kernel-argument segment and the constant address space -- scalar loads either way
#define g_model (this->gmodel())
#define g_lay (((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->lay)
#define GPH_G ((gph_cglobal *)(((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->G))
#endif

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
template <class T, int NN> GPH_DEV double gf64(T (GphLds::*m)[NN], int i) { return (gph_lds.*m)[i]; }
template <class T, int NN> GPH_DEV void sf64(T (GphLds::*m)[NN], int i, double v) { (gph_lds.*m)[i] = v; }
template <class T, int NN> GPH_DEV int gi16(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[i]); }
template <class T, int NN> GPH_DEV void si16(T (GphLds::*m)[NN], int i, int v) { (gph_lds.*m)[i] = (T)v; }
template <class T, int NN> GPH_DEV int gi32(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[i]); }
template <class T, int NN> GPH_DEV void si32(T (GphLds::*m)[NN], int i, int v) { (gph_lds.*m)[i] = (T)v; }
template <class T, int NN> GPH_DEV int gu8(T (GphLds::*m)[NN], int i) { return RFL((gph_lds.*m)[i]); }
template <class T, int NN> GPH_DEV void su8(T (GphLds::*m)[NN], int i, int v) { (gph_lds.*m)[i] = (T)v; }
template <class T, int NN> GPH_DEV int gu8v(T (GphLds::*m)[NN], int i) { return (gph_lds.*m)[i]; }
// two-dimensional scratch arrays [2][..] (instance 0/1 of the pending stat deltas)
template <class T, int NN> GPH_DEV double gf64(T (GphLds::*m)[2][NN], int k, int i) { return (gph_lds.*m)[k][i]; }
template <class T, int NN> GPH_DEV void sf64(T (GphLds::*m)[2][NN], int k, int i, double v) { (gph_lds.*m)[k][i] = v; }
template <class T, int NN> GPH_DEV int gi16(T (GphLds::*m)[2][NN], int k, int i) { return RFL((gph_lds.*m)[k][i]); }
template <class T, int NN> GPH_DEV void si16(T (GphLds::*m)[2][NN], int k, int i, int v) { (gph_lds.*m)[k][i] = (T)v; }
// dynamic-part accessors (byte offset GPH_Q_* of the locus' sequence block): lane-varying, pruning only
GPH_DEV double gf64(int off, int i) { return ((lf64 *)(GPH_SMB + off))[i]; }
GPH_DEV void sf64(int off, int i, double v) { ((lf64 *)(GPH_SMB + off))[i] = v; }
GPH_DEV int gi16(int off, int i) { return RFL(((li16 *)(GPH_SMB + off))[i]); }
GPH_DEV void si16(int off, int i, int v) { ((li16 *)(GPH_SMB + off))[i] = (int16_t)v; }
GPH_DEV int gi32(int off, int i) { return RFL(((li32 *)(GPH_SMB + off))[i]); }
GPH_DEV void si32(int off, int i, int v) { ((li32 *)(GPH_SMB + off))[i] = v; }
GPH_DEV int gu8(int off, int i) { return RFL(((lu8 *)(GPH_SMB + off))[i]); }
GPH_DEV void su8(int off, int i, int v) { ((lu8 *)(GPH_SMB + off))[i] = (uint8_t)v; }
// lane-varying byte/int loads (no readfirstlane): pruning only
GPH_DEV int gu8v(int off, int i) { return ((lu8 *)(GPH_SMB + off))[i]; }
GPH_DEV int gu16v(int off, int i) { return ((GPH_LDS uint16_t *)(GPH_SMB + off))[i]; }
GPH_DEV int gu16(int off, int i) { return RFL(((GPH_LDS uint16_t *)(GPH_SMB + off))[i]); }
GPH_DEV int gi32v(int off, int i) { return ((li32 *)(GPH_SMB + off))[i]; }
