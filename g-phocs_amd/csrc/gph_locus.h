// gph_locus.h -- per-locus device functions (one wavefront per locus, state in LDS).
//
// Everything here operates on the LDS copy of ONE locus (gph_rt.h accessors)
// and on the __constant__ model/layout tables.  Control flow is wave-uniform;
// only prune_node()/root_terms() spread (pattern, base) pairs over the 64 lanes.
//
// Arithmetic contract (bit-faithful state evolution): every floating-point
// expression keeps the reference's operand order and association and the file
// is compiled with -ffp-contract=off.  Reference citations are file:line under
// the upstream src/ directory.
#pragma once
#include "gph_rt.h"

// Everything below (and gph_kernels.h) is the body of ONE struct: the per-locus functions are its members and
// the wave-uniform INTEGER scalars of the locus -- page scalars (root, list lengths, free-list head, migration
// count, generator state), evaluation counters and the SPR bookkeeping -- are its data members.  Every member
// function is inlined into its kernel, the object never leaves registers, and the scalars sit in the lanes of ONE
// vector register: a read is a v_readlane with a constant lane, a write a v_writelane -- no LDS round trip
// (~90 cycles plus a v_readfirstlane) every time the chain logic branches on one of them.
// load_scalars() after the page has been staged in, flush_scalars() before it is staged out.
template <bool GPH_GM> struct GphCtxT {
#ifndef GPH_HOSTEMU
  GPH_DEV gph_cmodel &gmodel() const
  {
#if GPH_BIG_BANDS
    return *(gph_cmodel *)&(((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->G->model);
#else
    if constexpr (GPH_GM) return *(gph_cmodel *)&(((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->G->model);
    else return ((gph_ckargs *)__builtin_amdgcn_kernarg_segment_ptr())->model;
#endif
  }
#endif
  GphPad<IS_COUNT + CN_COUNT + SI_COUNT> r_pad;     /* GPH_PADGET / GPH_PADSET, gph_rt.h */
#ifdef GPH_LOGSTEPS
  // decision-level transcript of the selected loci (test builds only; GphDev::slog*): where this locus's records go
  double *slog_ = nullptr;
  int32_t *slog_n_ = nullptr;
  int32_t slog_cap_ = 0;
  long long slog_gen_ = 0;
#define GPH_SLOG_OPEN(D, g) do { slog_ = nullptr; if ((D).slog_map && (D).slog_map[g] >= 0) { const int si_ = (D).slog_map[g]; \
      slog_ = (D).slog + (size_t)si_ * (D).slog_cap * 8; slog_n_ = (D).slog_n + si_; slog_cap_ = (D).slog_cap; } } while (0)
#define GPH_SLOG(kind, a, b, c, d, e, f) do { if (slog_ && GPH_LANE == 0) { const int k_ = (*slog_n_)++; if (k_ < slog_cap_) { double *r_ = slog_ + (size_t)k_ * 8; \
      r_[0] = (kind); r_[1] = (double)(a); r_[2] = (double)(b); r_[3] = (double)(c); r_[4] = (double)(d); r_[5] = (double)(e); r_[6] = (double)(f); r_[7] = 0.0; } } } while (0)
#else
#define GPH_SLOG_OPEN(D, g) ((void)0)
#define GPH_SLOG(kind, a, b, c, d, e, f) ((void)0)
#endif
  // node sets of the saved version (LocusDataLikelihood.c:75-104: recalcConditionals[], changedNodeIds[] /
  // changedCondIds[]) and the double buffer's current halves, bit = genealogy node: wave-uniform 64-bit scalars for
  // as long as a kernel works on the locus, held in lane pairs of the scalar pad (page words IS_DIRTY / IS_CBIT /
  // IS_SAVED): a read is two lane reads, no LDS round trip.  Marking a node, resetSaved and the conditional half of
  // revertToSaved are a few scalar instructions; the reference's two id lists are not kept at all (a list is only
  // ever walked to visit the members of the set).
#define m_dirty_get() pad64(IS_DIRTY0)
#define m_cbit_get() pad64(IS_CBIT0)
  // gph_nset: a set of genealogy nodes -- one 64-bit scalar up to 32 leaves, a pair beyond (GPH_BIG_TREE)
#if GPH_BIG_TREE
  struct gph_nset { uint64_t w[GPH_NSQ]; };
  GPH_DEV static gph_nset ns_none() { gph_nset r; for (int q = 0; q < GPH_NSQ; q++) r.w[q] = 0; return r; }
  GPH_DEV static bool ns_has(const gph_nset &s, int i)
  {
    uint64_t x = s.w[0];
    for (int q = 1; q < GPH_NSQ; q++) if ((i >> 6) == q) x = s.w[q];     /* (selects, not an indexed private array) */
    return ((x >> (i & 63)) & 1) != 0;
  }
  GPH_DEV static gph_nset ns_with(gph_nset s, int i) { for (int q = 0; q < GPH_NSQ; q++) if ((i >> 6) == q) s.w[q] |= (uint64_t)1 << (i & 63); return s; }
  GPH_DEV static gph_nset ns_flip(gph_nset s, int i) { for (int q = 0; q < GPH_NSQ; q++) if ((i >> 6) == q) s.w[q] ^= (uint64_t)1 << (i & 63); return s; }
  GPH_DEV static gph_nset ns_xor(gph_nset a, const gph_nset &b) { for (int q = 0; q < GPH_NSQ; q++) a.w[q] ^= b.w[q]; return a; }
  GPH_DEV static bool ns_any(const gph_nset &s) { uint64_t x = 0; for (int q = 0; q < GPH_NSQ; q++) x |= s.w[q]; return x != 0; }
#define NS_GET(k) ns_get_<(k)>()
#define NS_PUT(k, v) ns_put_<(k)>((v))
#else
  typedef uint64_t gph_nset;
  GPH_DEV static gph_nset ns_none() { return 0; }
  GPH_DEV static bool ns_has(gph_nset s, int i) { return ((s >> i) & 1) != 0; }
  GPH_DEV static gph_nset ns_with(gph_nset s, int i) { return s | ((uint64_t)1 << i); }
  GPH_DEV static gph_nset ns_flip(gph_nset s, int i) { return s ^ ((uint64_t)1 << i); }
  GPH_DEV static gph_nset ns_xor(gph_nset a, gph_nset b) { return a ^ b; }
  GPH_DEV static bool ns_any(gph_nset s) { return s != 0; }
#define NS_GET(k) pad64(k)
#define NS_PUT(k, v) setpad64((k), (v))
#endif

#undef GPH_FILE_ID
#define GPH_FILE_ID 1
// ---------------------------------------------------------------- accessors
#define AGE(i) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].age)
#define setAGE(i, v) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].age = (v))
#define SVAGE(i) (gph_lds.sv[GPH_IX((i), GPH_CAP_N)].age)
#define setSVAGE(i, v) (gph_lds.sv[GPH_IX((i), GPH_CAP_N)].age = (v))
#define EVT(e) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].time)
#define setEVT(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].time = (v))
#define MAGE(m) gf64(&GphLds::mig_age, GPH_IX((m), GPH_MAX_MIGS))
#define setMAGE(m, v) sf64(&GphLds::mig_age, GPH_IX((m), GPH_MAX_MIGS), (v))
#define COALS(p) gf64(&GphLds::coal, GPH_IX((p), GPH_CAP_K))
#define setCOALS(p, v) sf64(&GphLds::coal, GPH_IX((p), GPH_CAP_K), (v))
#define MIGST(b) gf64(&GphLds::migst, GPH_IX((b), GPH_CAP_B))
#define setMIGST(b, v) sf64(&GphLds::migst, GPH_IX((b), GPH_CAP_B), (v))
#define RBAGE(i) gf64(&GphLds::rb_age, GPH_IX((i), GPH_CAP_RB))
#define setRBAGE(i, v) sf64(&GphLds::rb_age, GPH_IX((i), GPH_CAP_RB), (v))
#define FS(k) gf64(&GphLds::fscal, (k))
#define setFS(k, v) sf64(&GphLds::fscal, (k), (v))
#define FATH(i) RFL((int)gph_lds.nd[GPH_IX((i), GPH_CAP_N)].father)
#define setFATH(i, v) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].father = (int16_t)(v))
#define LEFT(i) RFL((int)gph_lds.nd[GPH_IX((i), GPH_CAP_N)].left)
#define setLEFT(i, v) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].left = (int16_t)(v))
#define RGHT(i) RFL((int)gph_lds.nd[GPH_IX((i), GPH_CAP_N)].right)
#define setRGHT(i, v) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].right = (int16_t)(v))
#define NPOP(i) RFL((int)gph_lds.nd[GPH_IX((i), GPH_CAP_N)].npop)
#define setNPOP(i, v) (gph_lds.nd[GPH_IX((i), GPH_CAP_N)].npop = (int16_t)(v))
#define NEV(i) gi16(&GphLds::nev, GPH_IX((i), GPH_CAP_N))
#define setNEV(i, v) si16(&GphLds::nev, GPH_IX((i), GPH_CAP_N), (v))
#define SVF(i) RFL((int)gph_lds.sv[GPH_IX((i), GPH_CAP_N)].father)
#define SVL(i) RFL((int)gph_lds.sv[GPH_IX((i), GPH_CAP_N)].left)
#define SVR(i) RFL((int)gph_lds.sv[GPH_IX((i), GPH_CAP_N)].right)
#define ENEXT(e) RFL((int)gph_lds.ev[GPH_IX((e), GPH_CAP_E)].next)
#define setENEXT(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].next = (int16_t)(v))
#define EPREV(e) RFL((int)gph_lds.ev[GPH_IX((e), GPH_CAP_E)].prev)
#define setEPREV(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].prev = (int16_t)(v))
#define ENODE(e) RFL((int)gph_lds.ev[GPH_IX((e), GPH_CAP_E)].node)
#define setENODE(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].node = (int16_t)(v))
#define ENLIN(e) RFL((int)gph_lds.ev[GPH_IX((e), GPH_CAP_E)].nlin)
#define setENLIN(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].nlin = (uint8_t)(v))
#define ETYPE(e) RFL((int)gph_lds.ev[GPH_IX((e), GPH_CAP_E)].type)
#define setETYPE(e, v) (gph_lds.ev[GPH_IX((e), GPH_CAP_E)].type = (uint8_t)(v))
#define FIRSTEV(p) gi16(&GphLds::first, GPH_IX((p), GPH_CAP_K))
#define setFIRSTEV(p, v) si16(&GphLds::first, GPH_IX((p), GPH_CAP_K), (v))
#define MG(m, f) gi16(&GphLds::mig_i, GPH_IX((m), GPH_MAX_MIGS) * MG_COUNT + (f))
#define setMG(m, f, v) si16(&GphLds::mig_i, GPH_IX((m), GPH_MAX_MIGS) * MG_COUNT + (f), (v))
#define LIVING(i) gi16(&GphLds::living, GPH_IX((i), GPH_MAX_MIGS))
#define setLIVING(i, v) si16(&GphLds::living, GPH_IX((i), GPH_MAX_MIGS), (v))
#define NCOAL(p) gi16(&GphLds::ncoal, GPH_IX((p), GPH_CAP_K))
#define setNCOAL(p, v) si16(&GphLds::ncoal, GPH_IX((p), GPH_CAP_K), (v))
#define NMIGB(b) gi16(&GphLds::nmig, GPH_IX((b), GPH_CAP_B))
#define setNMIGB(b, v) si16(&GphLds::nmig, GPH_IX((b), GPH_CAP_B), (v))
#define RBI(k, i) gi16(&GphLds::rb_i, (k) * GPH_CAP_RB + GPH_IX((i), GPH_CAP_RB))
#define setRBI(k, i, v) si16(&GphLds::rb_i, (k) * GPH_CAP_RB + GPH_IX((i), GPH_CAP_RB), (v))
#define ISC(k) GPH_PADGET(k)
#define setISC(k, v) GPH_PADSET((k), (v))
#define CBIT(i) ((int)ns_has(NS_GET(IS_CBIT0), (i)))
// scratch
#define DEV(inst, i) RFL((int)gph_lds.s_dev[inst][GPH_IX((i), GPH_CAP_E)])
#define setDEV(inst, i, v) (gph_lds.s_dev[inst][GPH_IX((i), GPH_CAP_E)] = (gph_evid)(v))
#define DCOAL(inst, i) gf64(&GphLds::s_dcoal, (inst), GPH_IX((i), GPH_CAP_K))
#define setDCOAL(inst, i, v) sf64(&GphLds::s_dcoal, (inst), GPH_IX((i), GPH_CAP_K), (v))
#define DMIG(inst, i) gf64(&GphLds::s_dmig, (inst), GPH_IX((i), GPH_CAP_B))
#define setDMIG(inst, i, v) sf64(&GphLds::s_dmig, (inst), GPH_IX((i), GPH_CAP_B), (v))
#define DPOPS(inst, i) gi16(&GphLds::s_dpops, (inst), GPH_IX((i), GPH_CAP_K))
#define setDPOPS(inst, i, v) si16(&GphLds::s_dpops, (inst), GPH_IX((i), GPH_CAP_K), (v))
#define DBANDS(inst, i) gi16(&GphLds::s_dbands, (inst), GPH_IX((i), GPH_CAP_B))
#define setDBANDS(inst, i, v) si16(&GphLds::s_dbands, (inst), GPH_IX((i), GPH_CAP_B), (v))
#define DI(inst, k) RFL(gph_lds.s_di[inst][k])
#define setDI(inst, k, v) (gph_lds.s_di[inst][k] = (v))
#define SPRI(k) GPH_PADGET(IS_COUNT + CN_COUNT + (k))
#define setSPRI(k, v) GPH_PADSET(IS_COUNT + CN_COUNT + (k), (v))
#define SPRA(k, i) gi16(&GphLds::s_spri16, (k) * GPH_MAX_MIGS + GPH_IX((i), GPH_MAX_MIGS))
#define setSPRA(k, i, v) si16(&GphLds::s_spri16, (k) * GPH_MAX_MIGS + GPH_IX((i), GPH_MAX_MIGS), (v))
#define SPRAGE(i) gf64(&GphLds::s_sprf, GPH_IX((i), GPH_MAX_MIGS + 2))
#define setSPRAGE(i, v) sf64(&GphLds::s_sprf, GPH_IX((i), GPH_MAX_MIGS + 2), (v))
#define SPRLN(r) gf64(&GphLds::s_sprf, GPH_MAX_MIGS + (r))
#define setSPRLN(r, v) sf64(&GphLds::s_sprf, GPH_MAX_MIGS + (r), (v))
#define CNT(k) GPH_PADGET(IS_COUNT + (k))
#define setCNT(k, v) GPH_PADSET(IS_COUNT + (k), (v))

// ordered list of live migration bands (<= 16 entries of 4 bits): the reference keeps
// int live_mig_bands[MAX_MIG_BANDS] with swap-removal; the order decides which band a
// sampled migration picks (patch.c:1167-1169), so it is reproduced exactly -- in one
// 64-bit scalar instead of a private-memory array.
struct LiveList {
  uint64_t bits;
  int n;
};
#if GPH_BIG_BANDS
// more than 16 bands: the entries are bytes of the LDS image (s_live); `bits` is unused.  ONE list is live at a time (every
// function that keeps one -- recalc_stats, rubber_band, trace_lineage, check_gtree_structure -- finishes with it before
// another starts: none of them calls another while its list is in use)
GPH_DEV int ll_get(const LiveList &l, int i) { (void)l; return gu8(&GphLds::s_live, i); }
GPH_DEV void ll_set(LiveList &l, int i, int v) { (void)l; su8(&GphLds::s_live, i, v); }
GPH_DEV void ll_push(LiveList &l, int v) { ll_set(l, l.n, v); l.n++; }
GPH_DEV int ll_find(const LiveList &l, int v)
{
  for (int i = 0; i < l.n; i++) if (ll_get(l, i) == v) return i;
  return l.n;
}
#else
GPH_DEV int ll_get(const LiveList &l, int i) { return (int)((l.bits >> (4 * i)) & 15); }
GPH_DEV void ll_set(LiveList &l, int i, int v) { l.bits = (l.bits & ~((uint64_t)15 << (4 * i))) | ((uint64_t)v << (4 * i)); }
GPH_DEV void ll_push(LiveList &l, int v) { ll_set(l, l.n, v); l.n++; }
// first index i < n whose nibble equals v, else n -- branch-free (zero-nibble test; the lowest hit is exact;
// checked against the loop on 2e8 random lists, tools/verify_llfind.c)
GPH_DEV int ll_find(const LiveList &l, int v)
{
  const uint64_t x = l.bits ^ (0x1111111111111111ull * (uint64_t)v);
  uint64_t t = (x - 0x1111111111111111ull) & ~x & 0x8888888888888888ull;
  if (l.n < 16) t &= (((uint64_t)1 << (4 * l.n)) - 1);
  return t ? (int)(__builtin_ctzll(t) >> 2) : l.n;
}
#endif
GPH_DEV void ll_swap_remove(LiveList &l, int i) { l.n--; ll_set(l, i, ll_get(l, l.n)); }

// every field of one genealogy node with ONE LDS access
struct GphNodeS { double age; int father, left, right, npop; };
GPH_DEVHOT GphNodeS ld_node(int node)
{
  GphNodeS r;
  const gph_w4 w = gph_ld16(&gph_lds.nd[GPH_IX(node, GPH_CAP_N)]);
  union { double d; uint32_t u[2]; } t;
  t.u[0] = w.x; t.u[1] = w.y;
  const int w2 = RFL((int)w.z), w3 = RFL((int)w.w);
  r.age = t.d;
  r.father = (int)(int16_t)w2;
  r.left = w2 >> 16;
  r.right = (int)(int16_t)w3;
  r.npop = w3 >> 16;
  return r;
}
// every field of one event with ONE LDS access (ds_read_b128 of the GphEv record)
struct GphEvS { double time; int next, prev, node, nlin, type; };
GPH_DEVHOT GphEvS ld_ev(int ev)
{
  GphEvS r;
  const gph_w4 w = gph_ld16(&gph_lds.ev[GPH_IX(ev, GPH_CAP_E)]);
  union { double d; uint32_t u[2]; } t;
  t.u[0] = w.x; t.u[1] = w.y;
  const int w2 = RFL((int)w.z), w3 = RFL((int)w.w);
  r.time = t.d;
  r.next = (int)(int16_t)w2;
  r.prev = w2 >> 16;
  r.node = (int)(int16_t)w3;
  r.nlin = (int)(uint8_t)(w3 >> 16);
  r.type = (int)((uint32_t)w3 >> 24);
  return r;
}

#define gmin2(a, b) ((a) < (b) ? (a) : (b))
#define gmax2(a, b) ((a) > (b) ? (a) : (b))

// optional in-kernel cycle attribution (diagnostic builds only: -DGPH_STAMPS); the sums
// leave the kernel through OUT slots 14/15 and a side buffer, never through a result
#if defined(GPH_STAMPS) && !defined(GPH_HOSTEMU)
#define STAMP_BEGIN(k) long long stamp_t0_##k = __builtin_readcyclecounter()
#define STAMP_END(k) gph_lds.s_stamp[k] += (double)(__builtin_readcyclecounter() - stamp_t0_##k)
#else
#define STAMP_BEGIN(k) ((void)0)
#define STAMP_END(k) ((void)0)
#endif
// -DGPH_STAMPS=2: slots 2/3/4 attribute lik_compute's own phases instead of the chain functions;
// -DGPH_STAMPS=3: slots 2/3/4 = SPR accept path, SPR reject path, migration-node sweep
#if defined(GPH_STAMPS) && GPH_STAMPS == 2 && !defined(GPH_HOSTEMU)
#define STAMPA_BEGIN(k) ((void)0)
#define STAMPA_END(k) ((void)0)
#define STAMPB_BEGIN(k) STAMP_BEGIN(k)
#define STAMPB_END(k) STAMP_END(k)
#define STAMPC_BEGIN(k) ((void)0)
#define STAMPC_END(k) ((void)0)
#elif defined(GPH_STAMPS) && GPH_STAMPS == 3 && !defined(GPH_HOSTEMU)
#define STAMPA_BEGIN(k) ((void)0)
#define STAMPA_END(k) ((void)0)
#define STAMPB_BEGIN(k) ((void)0)
#define STAMPB_END(k) ((void)0)
#define STAMPC_BEGIN(k) STAMP_BEGIN(k)
#define STAMPC_END(k) STAMP_END(k)
#else
#define STAMPA_BEGIN(k) STAMP_BEGIN(k)
#define STAMPA_END(k) STAMP_END(k)
#define STAMPB_BEGIN(k) ((void)0)
#define STAMPB_END(k) ((void)0)
#define STAMPC_BEGIN(k) ((void)0)
#define STAMPC_END(k) ((void)0)
#endif
GPH_DEV void gph_fail(int code) { if (CNT(CN_ERROR) == 0) setCNT(CN_ERROR, code); }
GPH_DEV int gph_failed() { return CNT(CN_ERROR) != 0; }
GPH_DEV int gph_errcode() { return CNT(CN_ERROR); }

#define pad64(k) ((uint64_t)(uint32_t)ISC(k) | ((uint64_t)(uint32_t)ISC((k) + 1) << 32))
#define setpad64(k, v) do { const uint64_t pv64_ = (v); setISC((k), (int)(uint32_t)pv64_); setISC((k) + 1, (int)(uint32_t)(pv64_ >> 32)); } while (0)
#if GPH_BIG_TREE
/* word q of the set that starts at scalar K: pad scalars K + 2q, K + 2q + 1 (compile-time lanes) */
template <int K, int Q> GPH_DEV void ns_get_words_(gph_nset &r) { if constexpr (Q < GPH_NSQ) { r.w[Q] = pad64(K + 2 * Q); ns_get_words_<K, Q + 1>(r); } }
template <int K> GPH_DEV gph_nset ns_get_() { gph_nset r; ns_get_words_<K, 0>(r); return r; }
template <int K, int Q> GPH_DEV void ns_put_words_(const gph_nset &v) { if constexpr (Q < GPH_NSQ) { setpad64(K + 2 * Q, v.w[Q]); ns_put_words_<K, Q + 1>(v); } }
template <int K> GPH_DEV void ns_put_(const gph_nset &v) { ns_put_words_<K, 0>(v); }
#endif
GPH_DEV void load_scalars() { r_pad.load(gph_lds.iscal, IS_COUNT); }
GPH_DEV void flush_scalars() { r_pad.store(gph_lds.iscal, IS_COUNT); }

// exact quotients a / theta[pop], a / b with y = 1/b at hand, a / 3 (gph_quot, gph_rt.h)
GPH_DEV double gph_div_theta(double a, int pop) { return gph_quot(a, g_model.theta[pop], g_model.thetaInv[pop]); }
GPH_DEV double gph_div_by(double a, double b, double y) { return gph_quot(a, b, y); }
GPH_DEV double gph_div3(double a) { return gph_quot(a, 3.0, 1.0 / 3.0); }

// ---------------------------------------------------------------- RNG
// rndu, utils.c:498-513: unsigned 32-bit Wichmann-Hill without the sign fix-up
// The locus' generator state (RndCtx slot, utils.c:401) lives in registers while a kernel works on
// the locus: loaded from the page after stage-in, stored back before stage-out.
struct GphRng { uint32_t x, y, z; };
GPH_DEV void rng_load(GphRng &g) { g.x = (uint32_t)ISC(IS_RX); g.y = (uint32_t)ISC(IS_RY); g.z = (uint32_t)ISC(IS_RZ); }
GPH_DEV void rng_store(const GphRng &g) { setISC(IS_RX, (int)g.x); setISC(IS_RY, (int)g.y); setISC(IS_RZ, (int)g.z); }
GPH_DEV double l_rndu(GphRng &g)
{
  uint32_t x = g.x, y = g.y, z = g.z;
  double r;
  /* 171 * (x % 177) - 2 * (x / 177) = 171 * x - (171 * 177 + 2) * (x / 177) in arithmetic modulo 2^32 (which is what
   * the unsigned expression is): one quotient, two products and a difference per component */
  x = 171u * x - 30269u * (x / 177u);
  y = 172u * y - 30307u * (y / 176u);
  z = 170u * z - 30323u * (z / 178u);
  g.x = x;
  g.y = y;
  g.z = z;
#if defined(GPH_HOSTEMU) || !defined(__HIP_DEVICE_COMPILE__)   /* host form (and the host pass of hipcc) */
  r = x / 30269.0 + y / 30307.0 + z / 30323.0;
#else
  /* IEEE-exact quotients without the 14-instruction divide expansion: for a 32-bit integer x
   * and d in {30269, 30307, 30323}, q = fma(fma(-q0, d, x), 1/d, q0) with q0 = x * RN(1/d)
   * equals RN(x / d) for ALL 2^32 values of x (exhaustively verified, tools/verify_rng_div.c) */
  {
    /* the six constants come from the kernel-argument segment in one scalar load (see gph_math.h) */
    const gph_cdbl *RC = GPH_RNGC;
    const double rx = RC[0], ry = RC[1], rz = RC[2], mx = RC[3], my = RC[4], mz = RC[5];
    double xd = (double)x, yd = (double)y, zd = (double)z, q;
    q = xd * rx; double qx = __builtin_fma(__builtin_fma(-q, mx, xd), rx, q);
    q = yd * ry; double qy = __builtin_fma(__builtin_fma(-q, my, yd), ry, q);
    q = zd * rz; double qz = __builtin_fma(__builtin_fma(-q, mz, zd), rz, q);
    r = qx + qy + qz;
  }
  /* r - (int)r for 0 <= r < 2^19: both the subtraction and v_fract_f64's r - floor(r) are exact, i.e. the same value
   * (one instruction instead of convert, convert back, subtract) */
  return __builtin_amdgcn_fract(r);
#endif
  r = (r - (int)r);
  return r;
}
// ---- the sweep kernel's generator: the SAME stream, a batch of draws (GPH_RNG_BATCH) at a time.  The integer recurrences stay serial (scalar
// unit, one lane of three registers written per step); the expensive part of a draw -- three exact quotients, two sums and
// the fractional part, 15 fp64 vector instructions that used to run for ONE useful lane -- runs once per batch with a draw in
// every lane.  Handing a draw out is two lane reads.  The state written back to the page is the one after the last draw
// handed out (lane pos-1 of the batch), so a kernel leaves the generator exactly where the serial code leaves it.
#ifdef GPH_HOSTEMU
typedef GphRng GphRngB;
#else
// draws per batch: a batch is generated in full and the draws a kernel has not handed out when it ends are thrown away
// (and the state replayed to where it stopped): 64 per batch wasted 32 draws + 32 replay steps per kernel on average;
// measured 64 / 32 / 16 / 8: 16 and 32 are the fastest (-0.6 % sweep against 64), 8 pays more in refills than it saves
#ifndef GPH_RNG_BATCH
#define GPH_RNG_BATCH 16
#endif
struct GphRngB {
  int pos;               // draws of the batch handed out; GPH_RNG_BATCH = none left
  double u;              // per lane: draw `lane` of the batch
};
// The generator's integer state is touched twice per batch of draws, so it lives in lanes of the scalar pad, not in scalar
// registers that stay allocated (and get spilled) across the whole sweep: page scalars IS_RX / IS_RY / IS_RZ = the state
// the current batch STARTED from, CN_RX / CN_RY / CN_RZ = the state after its last draw (where the next batch starts).
GPH_DEV void rng_load(GphRngB &g)
{
  setCNT(CN_RX, ISC(IS_RX)); setCNT(CN_RY, ISC(IS_RY)); setCNT(CN_RZ, ISC(IS_RZ));
  g.pos = GPH_RNG_BATCH; g.u = 0.0;
}
// The state to leave behind is the one after the last draw handed out: the recurrences replayed from the batch's start
// for `pos` steps, once per kernel (scalar unit).  Keeping the per-lane states of the batch for this instead cost three
// vector registers for the whole kernel, and the sweep kernel is compiled for 64 of them.
GPH_DEV void rng_store(const GphRngB &g)
{
  uint32_t x = (uint32_t)CNT(CN_RX), y = (uint32_t)CNT(CN_RY), z = (uint32_t)CNT(CN_RZ);
  if (g.pos < GPH_RNG_BATCH) {
    x = (uint32_t)ISC(IS_RX); y = (uint32_t)ISC(IS_RY); z = (uint32_t)ISC(IS_RZ);
    for (int k = 0; k < g.pos; k++) {
      x = 171u * x - 30269u * (x / 177u);
      y = 172u * y - 30307u * (y / 176u);
      z = 170u * z - 30323u * (z / 178u);
    }
  }
  setISC(IS_RX, (int)x); setISC(IS_RY, (int)y); setISC(IS_RZ, (int)z);
}
#ifndef GPH_RNG_ROLLED
template <int K> GPH_DEV void rng_steps_(uint32_t &x, uint32_t &y, uint32_t &z, int &vx, int &vy, int &vz)
{
  if constexpr (K < GPH_RNG_BATCH) {
    x = 171u * x - 30269u * (x / 177u);
    y = 172u * y - 30307u * (y / 176u);
    z = 170u * z - 30323u * (z / 178u);
    asm("v_writelane_b32 %0, %3, %6\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6"
        : "+v"(vx), "+v"(vy), "+v"(vz) : "s"(x), "s"(y), "s"(z), "i"(K));
    rng_steps_<K + 1>(x, y, z, vx, vy, vz);
  }
}
#endif
#ifndef GPH_RNG_VECTOR_REFILL      /* the three recurrences on the scalar unit: the product form (see the measurement at the vector form below) */
GPH_DEV void rng_refill(GphRngB &g)
{
  uint32_t x = (uint32_t)CNT(CN_RX), y = (uint32_t)CNT(CN_RY), z = (uint32_t)CNT(CN_RZ);
  int vx = 0, vy = 0, vz = 0;
  setISC(IS_RX, (int)x); setISC(IS_RY, (int)y); setISC(IS_RZ, (int)z);
#ifndef GPH_RNG_ROLLED
  /* fully unrolled (round 5): the lane of every write is an immediate -- no s_mov m0 / s_nop / index add per draw, 21 instead
   * of 24 instructions per draw at 16 x the code per refill site: sweep -0.8 %, four interleaved A/B pairs, every pair in
   * favour (profiles/r05_ab_rng_unroll.txt) */
  rng_steps_<0>(x, y, z, vx, vy, vz);
#else
#pragma unroll 4
  for (int k = 0; k < GPH_RNG_BATCH; k++) {
    x = 171u * x - 30269u * (x / 177u);
    y = 172u * y - 30307u * (y / 176u);
    z = 170u * z - 30323u * (z / 178u);
    /* lane select through M0: a VOP3 instruction reads one scalar register besides it (constant-bus limit).  Nothing
     * else in this translation unit's code uses M0 (gfx9 DS instructions do not need it; checked in the ISA) */
    asm("s_mov_b32 m0, %6\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
        : "+v"(vx), "+v"(vy), "+v"(vz) : "s"(x), "s"(y), "s"(z), "s"(k) : "m0");
  }
#endif
  setCNT(CN_RX, (int)x); setCNT(CN_RY, (int)y); setCNT(CN_RZ, (int)z);
#if defined(__HIP_DEVICE_COMPILE__)       /* (the host pass of hipcc only parses this) */
  {
    const gph_cdbl *RC = GPH_RNGC;
    const double rx = RC[0], ry = RC[1], rz = RC[2], mx = RC[3], my = RC[4], mz = RC[5];
    double xd = (double)(uint32_t)vx, yd = (double)(uint32_t)vy, zd = (double)(uint32_t)vz, q;
    q = xd * rx; double qx = __builtin_fma(__builtin_fma(-q, mx, xd), rx, q);
    q = yd * ry; double qy = __builtin_fma(__builtin_fma(-q, my, yd), ry, q);
    q = zd * rz; double qz = __builtin_fma(__builtin_fma(-q, mz, zd), rz, q);
    g.u = __builtin_amdgcn_fract(qx + qy + qz);
  }
#endif
  g.pos = 0;
}
#else
// Round 5 experiment, NOT the product form: measured 3 % SLOWER (sweep 9.46 / 9.52 -> 9.77 / 9.78 ms, two
// interleaved A/B pairs on one box, same accept counters: profiles/r05_ab_rng_vector.txt) -- the vector pipe is the busier one
// (9 quarter-/full-rate vector instructions per draw = 28 SIMD-cycles replace 18 scalar ones + 12.6 SIMD-cycles of lane writes),
// seven live vector registers at every refill site add spills (9 -> 28), and twelve inlined sites add 750 static instructions.
// The recurrences run on the VECTOR unit, one component per row of 16 lanes (row 0 = x, 1 = y, 2 = z; row 3 runs z
// again and is ignored), every lane of a row redundantly.  A step is
//   q = x / d      by the 33-bit multiply-shift of Granlund & Montgomery: t = mulhi(x, M'), q = (t + ((x - t) >> 1)) >> 7
//                  with M' = floor(2^32 (256 - d) / d) + 1 -- equal to x / d for ALL 2^32 x and d = 177, 176, 178
//                  (exhaustively: tools/verify_rng_magic.c),
//   x = a x - m q  modulo 2^32 (= a (x % d) - c (x / d), utils.c:503-505, as in the scalar form above),
// and one v_cndmask_b32_dpp that shifts the row's history register one lane down (row_shl:1: lane i <- lane i + 1) and puts
// the new state into the row's last lane: after 16 steps lane j of a row holds the state after step j + 1, i.e. the integers of
// draw j.  9 vector instructions per draw where the scalar form had 18 scalar ones + s_mov m0 + s_nop + 3 v_writelane; the
// three quotients x / 30269.0 ... then run ONCE over the 48 lanes with per-row constants, and the y / z quotients come down
// to row 0 with four ds_bpermute_b32.  One asm statement: the DPP select reads the row-end mask from vcc, which must
// survive the 16 steps (no instruction in between writes it), and the compiler must not re-schedule a write of the history
// register next to its DPP read (2 wait states on gfx9: inside the block the previous write is 8 instructions away).
#define GPH_RNGV_STEP \
  "v_mul_hi_u32 %[q], %[st], %[mg]\n\t" \
  "v_mul_lo_u32 %[t], %[st], %[a]\n\t" \
  "v_sub_u32 %[st], %[st], %[q]\n\t" \
  "v_lshrrev_b32 %[st], 1, %[st]\n\t" \
  "v_add_u32 %[st], %[st], %[q]\n\t" \
  "v_lshrrev_b32 %[st], 7, %[st]\n\t" \
  "v_mul_lo_u32 %[st], %[st], %[nm]\n\t" \
  "v_add_u32 %[st], %[st], %[t]\n\t" \
  "v_cndmask_b32_dpp %[h], %[h], %[st], vcc row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define GPH_RNGV_STEP4 GPH_RNGV_STEP GPH_RNGV_STEP GPH_RNGV_STEP GPH_RNGV_STEP
GPH_DEV void rng_refill(GphRngB &g)
{
  static_assert(GPH_RNG_BATCH == 16, "one draw per lane of a 16-lane row");
  const uint32_t x0 = (uint32_t)CNT(CN_RX), y0 = (uint32_t)CNT(CN_RY), z0 = (uint32_t)CNT(CN_RZ);
  setISC(IS_RX, (int)x0); setISC(IS_RY, (int)y0); setISC(IS_RZ, (int)z0);
#if defined(__HIP_DEVICE_COMPILE__)       /* (the host pass of hipcc only parses this) */
  const int row = GPH_LANE >> 4;
  const bool r0 = row == 0, r1 = row == 1;
  uint32_t st = r0 ? x0 : r1 ? y0 : z0, h = 0, q, t;
  const uint32_t mg = r0 ? 0x724287f5u : r1 ? 0x745d1746u : 0x702e05c1u;      /* M' of 177, 176, 178 */
  const uint32_t a = r0 ? 171u : r1 ? 172u : 170u;
  const uint32_t nm = r0 ? 0u - 30269u : r1 ? 0u - 30307u : 0u - 30323u;      /* -(a d + c): 171 * 177 + 2 = 30269, ... */
  asm("s_mov_b32 vcc_lo, 0x80008000\n\t"
      "s_mov_b32 vcc_hi, 0x80008000\n\t"
      GPH_RNGV_STEP4 GPH_RNGV_STEP4 GPH_RNGV_STEP4 GPH_RNGV_STEP4
      : [st] "+v"(st), [h] "+v"(h), [q] "=&v"(q), [t] "=&v"(t) : [mg] "v"(mg), [a] "v"(a), [nm] "v"(nm) : "vcc");
  /* every lane of a row holds the state after the batch's last draw */
  setCNT(CN_RX, __builtin_amdgcn_readlane((int)st, 0)); setCNT(CN_RY, __builtin_amdgcn_readlane((int)st, 16));
  setCNT(CN_RZ, __builtin_amdgcn_readlane((int)st, 32));
  {
    const gph_cdbl *RC = GPH_RNGC;
    const double rr = r0 ? RC[0] : r1 ? RC[1] : RC[2], mm = r0 ? RC[3] : r1 ? RC[4] : RC[5];
    const double hd = (double)h, q0 = hd * rr;
    const double qq = __builtin_fma(__builtin_fma(-q0, mm, hd), rr, q0);      /* x / 30269.0 | y / 30307.0 | z / 30323.0, exact (l_rndu above) */
    union { double d; int32_t i[2]; } u, vy, vz;
    u.d = qq;
    const int ay = (GPH_LANE + 16) << 2, az = (GPH_LANE + 32) << 2;
    vy.i[0] = __builtin_amdgcn_ds_bpermute(ay, u.i[0]); vy.i[1] = __builtin_amdgcn_ds_bpermute(ay, u.i[1]);
    vz.i[0] = __builtin_amdgcn_ds_bpermute(az, u.i[0]); vz.i[1] = __builtin_amdgcn_ds_bpermute(az, u.i[1]);
    g.u = __builtin_amdgcn_fract(qq + vy.d + vz.d);      /* lanes 0 .. 15: draw `lane` (the other rows hold sums nobody reads) */
  }
#endif
  g.pos = 0;
}
#endif
GPH_DEV double l_rndu(GphRngB &g)
{
  if (g.pos >= GPH_RNG_BATCH) rng_refill(g);
  const double u = gph_readlane64(g.u, g.pos);
  g.pos++;
  return u;
}
#endif
// rndnormal, utils.c:459-472
template <class RNG> GPH_DEV double l_rndnormal(RNG &g)
{
  double u, v, s;
  int guard = 0;
  for (;;) {
    if (++guard > 100000) { gph_fail(89); return 0.0; }
    u = 2 * l_rndu(g) - 1;
    v = 2 * l_rndu(g) - 1;
    s = u * u + v * v;
    if (UNI(s > 0 && s < 1)) break;
  }
  s = sqrt(-2. * gph_log_u(s) / s);
  return u * s;
}
// rnd2normal8, utils.c:482-488 (kernel constants utils.c:427-431)
template <class RNG> GPH_DEV double l_rnd2normal8(RNG &g)
{
  const double m2s2 = 8.;
  double m2N = sqrt(m2s2 / (m2s2 + 1.));
  double s2N = sqrt(1. / (m2s2 + 1.));
  double z = m2N + l_rndnormal(g) * s2N;
  z = UNI(l_rndu(g) < 0.5) ? z : -z;
  return z;
}
// reflect, utils.c:333-398
GPH_DEV double l_reflect(double x, double a, double b)
{
  const double slack = 0.000000001;
  double xnew, di;
  a += slack;
  b -= slack;
  if (UNI(b <= a)) return (a + b) / 2.;
  if (UNI(x < b && x > a)) return x;
  xnew = x;
  if (UNI(xnew <= a)) xnew = 2. * a - xnew;
  di = 2. * (b - a);
  xnew = xnew - di * floor((xnew - a) / di);
  if (UNI(xnew >= b)) xnew = 2. * b - xnew;
  /* the reference loops here until the value is inside; landing exactly on a bound
   * ping-pongs forever there.  A wavefront must not hang: bail out after 64 folds. */
  int guard = 0;
  while (UNI(xnew <= a || xnew >= b)) {
    if (UNI(xnew >= b)) xnew = 2. * b - xnew;
    else xnew = 2 * a - xnew;
    if (++guard > 64) { gph_fail(90); return (a + b) / 2.; }
  }
  return xnew;
}

// ---------------------------------------------------------------- data likelihood
// copyNodeConditionals, LocusDataLikelihood.c:1889-1906: a node not yet marked in this proposal switches to the other
// half of its double buffer
GPH_DEV int lik_mark_cond(int node)
{
  const gph_nset d = NS_GET(IS_DIRTY0);
  if (CNT(CN_P) <= 0 || ns_has(d, node)) return 1;
  NS_PUT(IS_DIRTY0, ns_with(d, node));
  NS_PUT(IS_CBIT0, ns_flip(NS_GET(IS_CBIT0), node));
  return 0;
}
// copyNodeToSaved, LocusDataLikelihood.c:1864-1876
GPH_DEV void lik_save_node(int node, int recalc)
{
  if (recalc) lik_mark_cond(node);
  NS_PUT(IS_SAVED0, ns_with(NS_GET(IS_SAVED0), node));
  gph_lds.sv[node] = gph_lds.nd[node];   /* one 16-byte record: age, father, left, right */
}
// adjustGenNodeAge, LocusDataLikelihood.c:875-882
GPH_DEV void lik_adjust_age(int node, double age)
{
  lik_save_node(node, 1);
  setAGE(node, age);
}
// resetSaved, LocusDataLikelihood.c:852-864
GPH_DEV void lik_reset_saved()
{
  NS_PUT(IS_SAVED0, ns_none());
  NS_PUT(IS_DIRTY0, ns_none());
  setISC(IS_SV_ROOT, -1);
  setFS(FS_SV_DATALNL, FS(FS_DATALNL));
}
// revertToSaved, LocusDataLikelihood.c:768-841 (value semantics: a node's saved record and its previous conditional
// array are restored).  Every saved node on its own lane: one 16-byte copy each, all at once (the reference walks
// changedNodeIds[]; the copyAll case of mixing is the same thing with every node in the set).  Every node whose
// conditionals were recomputed goes back to the other half of its double buffer.
GPH_DEV void lik_revert()
{
  setFS(FS_DATALNL, FS(FS_SV_DATALNL));
  if (ISC(IS_SV_ROOT) >= 0) { setISC(IS_ROOT, ISC(IS_SV_ROOT)); setISC(IS_SV_ROOT, -1); }
  const gph_nset sv_ = NS_GET(IS_SAVED0);
  if (ns_any(sv_)) {
#if GPH_BIG_TREE
    GPH_EACH(k, g_lay.N)
#else
    GPH_EACH1(k, g_lay.N)
#endif
    {
      if (ns_has(sv_, k)) {
        GphNode r = gph_lds.sv[k];
        r.npop = gph_lds.nd[k].npop;     /* nodePops is not part of the saved version */
        gph_lds.nd[k] = r;
      }
    }
  }
  NS_PUT(IS_CBIT0, ns_xor(NS_GET(IS_CBIT0), NS_GET(IS_DIRTY0)));
  NS_PUT(IS_DIRTY0, ns_none());
  NS_PUT(IS_SAVED0, ns_none());
}

// computeEdgeConditionalJC, LocusDataLikelihood.c:1831-1848
GPH_DEV double edge_prob(double len)
{
  if (len < 1e-100) return 0.0;
  return ((1 - gph_exp_u(-4 * len / 3.0)) / 4.0);
}

// same, argument differs per lane (table words gathered through the vector L1)
GPH_DEV double edge_prob_v(double len)
{
  if (len < 1e-100) return 0.0;
  return ((1 - gph_exp(gph_div3(-4 * len))) / 4.0);
}

// The fp64 conditional arrays [2][n-1][P][4] of the locus stay in global memory: they are
// touched lane-parallel and fully coalesced (4P consecutive doubles per node), the working
// set of all resident waves fits the L2 / Infinity Cache, and keeping them out of LDS is
// what lets 16+ loci be resident per CU instead of ~6 (the chain logic is latency-bound,
// so resident loci per CU is the throughput lever).  The wave's base pointer sits in two
// LDS scratch words.
GPH_DEV gdbl *cond_base()
{
  uint64_t lo = (uint32_t)RFL(gph_lds.s_condptr[0]), hi = (uint32_t)RFL(gph_lds.s_condptr[1]);
  return (gdbl *)(uintptr_t)(lo | (hi << 32));
}
// the sequence block as the generic (pattern, base) paths see it: dynamic LDS, or HBM for a locus whose block outgrows it
GPH_DEV GphSeq seq_ref()
{
  GphSeq S; S.g = nullptr;
  if (CNT(CN_HUGE)) {
    const uint64_t lo = (uint32_t)CNT(CN_SEQLO), hi = (uint32_t)CNT(CN_SEQHI);
    S.g = (GPH_GLB char *)(uintptr_t)(lo | (hi << 32));
  }
  return S;
}
GPH_DEV void set_cond_base(const void *p)
{
  uint64_t v = (uint64_t)(uintptr_t)p;
  gph_lds.s_condptr[0] = (uint32_t)v;
  gph_lds.s_condptr[1] = (uint32_t)(v >> 32);
}
// index (in doubles) of the conditional array (buffer `bit`) of internal node `node`
GPH_DEV int cond_off(int node, int bit)
{
  int P = CNT(CN_P);
  return ((bit * (g_lay.n - 1) + (node - g_lay.n)) * P) * 4;
}

// one child's factor for (pattern p, base a): computeSubtreeConditionals_new,
// LocusDataLikelihood.c:1650-1673.  A leaf child is a base code (one-hot / N).
template <class CP>
GPH_DEV double child_factor(int child, CP cnd, int p, int a, double pe, double qe, int q_leaf = GPH_Q_LEAF, const GphSeq SQ = GphSeq{nullptr})
{
  double s0, s1, s2, s3, sa, S, Sp;
  if (child < g_lay.n) {
    int code = GPH_LEAFCODE_S(SQ, q_leaf, p, child);
    s0 = (code == 4 || code == 0) ? 1.0 : 0.0;
    s1 = (code == 4 || code == 1) ? 1.0 : 0.0;
    s2 = (code == 4 || code == 2) ? 1.0 : 0.0;
    s3 = (code == 4 || code == 3) ? 1.0 : 0.0;
  } else {
    s0 = cnd[4 * p + 0];
    s1 = cnd[4 * p + 1];
    s2 = cnd[4 * p + 2];
    s3 = cnd[4 * p + 3];
  }
  sa = a == 0 ? s0 : a == 1 ? s1 : a == 2 ? s2 : s3;
  S = 0.0;
  S += s0;
  S += s1;
  S += s2;
  S += s3;
  if (S >= 4) return 1.0; /* missing data below this edge: factor not applied (x*1.0 == x) */
  Sp = S * pe;
  return (Sp + sa * qe);
}

// recompute the conditionals of one internal node: lanes = (pattern, base) pairs
// inner loops of computeConditionalJC_new, LocusDataLikelihood.c:1596-1633
GPH_DEV void prune_node(int node)
{
  int l = LEFT(node), r = RGHT(node), P = CNT(CN_P);
  double mut = FS(FS_MUTRATE); /* locus mutation rate (1 under CONST, GPhoCS.c:1141) */
  double pl = edge_prob(mut * (AGE(node) - AGE(l)));
  double ql = 1 - 4.0 * pl;
  double pr = edge_prob(mut * (AGE(node) - AGE(r)));
  double qr = 1 - 4.0 * pr;
  gdbl *cb = cond_base();
  gdbl *pc = cb + cond_off(node, CBIT(node));
  const gdbl *lc = cb + (l >= g_lay.n ? cond_off(l, CBIT(l)) : 0);
  const gdbl *rc = cb + (r >= g_lay.n ? cond_off(r, CBIT(r)) : 0);
  const GphSeq SQ = seq_ref();
  int idx;
  for (idx = GPH_LANE; idx < 4 * P; idx += GPH_NLANES) {
    int p = idx >> 2, a = idx & 3;
    double v = 1.0, f;
    f = child_factor(l, lc, p, a, pl, ql, GPH_Q_LEAF, SQ);
    v *= f;
    f = child_factor(r, rc, p, a, pr, qr, GPH_Q_LEAF, SQ);
    v *= f;
    pc[idx] = v;
  }
  GPH_SYNC();
}

// GPH_LANE_NODES: the wave programs that keep one genealogy node per lane (2n - 1 <= 63): device builds up to 32 leaves
#if !defined(GPH_HOSTEMU) && !GPH_BIG_TREE
#define GPH_LANE_NODES 1
#else
#define GPH_LANE_NODES 0
#endif
// GPH_DEVFORMS: the device forms of the lane-parallel functions are compiled -- every device build, and the host build with the
// 64-lane micro-wave (gph_emu64.h), where they run next to the one-lane host forms
#if !defined(GPH_HOSTEMU) || defined(GPH_EMU64)
#define GPH_DEVFORMS 1
#else
#define GPH_DEVFORMS 0
#endif
#if GPH_DEVFORMS
// the next lane's value (lane i <- lane i + 1, the last lane gets 0): two v_mov_b32_dpp wave_shl:1 -- VALU register moves,
// against two ds_bpermute_b32 (an LDS-pipe instruction and a crossbar round trip each; tools/probe/dpp_probe.cpp has the
// direction check on gfx950)
GPH_DEV double dpp_next64(double v)
{
  union { double d; int32_t i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x130, 0xf, 0xf, true);
  u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x130, 0xf, 0xf, true);
  return u.d;
}
// the further phases of an unphased pattern (LocusDataLikelihood.c:466-479): the rows after pattern `lane` of the root's
// conditionals sit in the NEXT lanes' registers q0..q3 (rows of one pattern are adjacent, all below P <= 64); every round
// moves the four values one lane down and the lanes that still have a phase add them in the reference's order (phase
// by phase, base by base)
GPH_DEVHOT double add_phases(double prob, int ph, double q0, double q1, double q2, double q3)
{
  for (int k = 1; __ballot(ph > k) != 0; k++) {
    q0 = dpp_next64(q0); q1 = dpp_next64(q1); q2 = dpp_next64(q2); q3 = dpp_next64(q3);
    if (ph > k) {
      prob += q0;
      prob += q1;
      prob += q2;
      prob += q3;
    }
  }
  return prob;
}
// sum of term[0..P-1] in lane order, P <= 64 (lanes >= P and lanes without a term hold +0.0: x + 0.0 == x bit for
// bit, and the running sum is never -0.0): two lane reads with a constant lane + one add per pattern, no mask
// bookkeeping, an exit test every eight patterns.  (Measured alternative: every lane fetching term[k] through the LDS
// crossbar -- ds_bpermute with the lane in the offset field, one add per pattern, a third of the vector instructions
// -- is 2 % SLOWER: the permutes' round trip sits on the evaluation's critical path.)
GPH_DEVHOT double ordered_sum64(double term, int P)
{
  double s = 0.0;
#define GPH_ADD8(b) s += gph_readlane64(term, (b) + 0); s += gph_readlane64(term, (b) + 1); s += gph_readlane64(term, (b) + 2); s += gph_readlane64(term, (b) + 3); \
                    s += gph_readlane64(term, (b) + 4); s += gph_readlane64(term, (b) + 5); s += gph_readlane64(term, (b) + 6); s += gph_readlane64(term, (b) + 7);
  GPH_ADD8(0)
  if (P > 8) { GPH_ADD8(8)
  if (P > 16) { GPH_ADD8(16)
  if (P > 24) { GPH_ADD8(24)
  if (P > 32) { GPH_ADD8(32)
  if (P > 40) { GPH_ADD8(40)
  if (P > 48) { GPH_ADD8(48)
  if (P > 56) { GPH_ADD8(56) } } } } } } }
#undef GPH_ADD8
  return s;
}
#endif
// the same sum with the terms handed over through LDS: every lane stores its term once, then every lane reads term k
// at a uniform address (a broadcast read, two terms per 16-byte read) and adds it -- ONE vector instruction per pattern
// instead of three (two lane reads + the add), at the price of one LDS read per two patterns.  The reads are
// independent of the running sum and are issued ahead of it.  q_terms: 8 bytes per pattern (rounded up to eight patterns)
// of the dynamic LDS behind the sequence block, for the loci that have them inside the launch group's allocation -- the
// host sizes that so that the terms do not cost a resident workgroup (g_lay.lds_sum, g_lay.dyn_bytes, gph_engine_load_loci).
// Measured: -4 % vector instructions per sweep wavefront, -1.0 % sweep time.
#if GPH_DEVFORMS
GPH_DEVHOT double ordered_sum64_lds(double term, int P, int q_terms)
{
  lf64 *t = (lf64 *)(GPH_SMB + q_terms);
  if (GPH_LANE < ((P + 7) & ~7)) t[GPH_LANE] = term;     /* the terms the sum below reads: P rounded up to its groups of eight */
  GPH_WAVE_FENCE();
  double s = 0.0;
#define GPH_ADD8L(b) s += t[(b) + 0]; s += t[(b) + 1]; s += t[(b) + 2]; s += t[(b) + 3]; s += t[(b) + 4]; s += t[(b) + 5]; s += t[(b) + 6]; s += t[(b) + 7];
  GPH_ADD8L(0)
  if (P > 8) { GPH_ADD8L(8)
  if (P > 16) { GPH_ADD8L(16)
  if (P > 24) { GPH_ADD8L(24)
  if (P > 32) { GPH_ADD8L(32)
  if (P > 40) { GPH_ADD8L(40)
  if (P > 48) { GPH_ADD8L(48)
  if (P > 56) { GPH_ADD8L(56) } } } } } } }
#undef GPH_ADD8L
  GPH_WAVE_FENCE();
  return s;
}
#endif
// root reduction of the generic (pattern, base) paths, LocusDataLikelihood.c:466-479: per unphased pattern
// log(sum over phases and bases / (4 phases)) * count, summed in pattern order.  Device: 64 patterns at a time -- a lane per
// pattern computes its term (a pattern that is a further phase of another one, or lies beyond P, contributes +0.0: x + 0.0
// == x bit for bit, and the running sum is never -0.0), the 64 terms are added to the running sum in lane order: the
// additions of the serial loop.  (Rounds 1-4 kept a per-pattern terms array in LDS and walked it with two LDS round trips
// per pattern; a locus whose block lies in HBM would pay two memory round trips per pattern for that.)  Host: the serial
// loop, terms behind the block.
template <class CP>
GPH_DEVHOT double root_sum_generic(CP rc, int P, int q_phases, int q_count, int q_terms, const GphSeq SQ)
{
  double lnl = 0.0;
#ifdef GPH_HOSTEMU
#ifdef GPH_EMU64
  if (!gph_emu::in_wave())        /* (inside a micro-wave: the device form below) */
#endif
  {
  for (int p = 0; p < P; p++) {
    const int ph = GPH_PHASES(sq_u16v(SQ, q_phases, p));
    if (ph > 0) {
      const int nc = 4 * ph;
      double prob = 0.0;
      for (int c = 0; c < nc; c++) prob += rc[p * 4 + c];
      sq_sf64(SQ, q_terms, p, gph_log(prob / nc) * GPH_PATCOUNT_S(SQ, q_count, p));
    }
  }
  for (int p = 0; p < P; p++)
    if (sq_u16v(SQ, q_phases, p) > 0) lnl += sq_f64(SQ, q_terms, p);
  return lnl;
  }
#endif
#if GPH_DEVFORMS
  (void)q_terms;
  for (int p0 = 0; p0 < P; p0 += GPH_WAVE) {
    const int p = p0 + GPH_LANE;
    double term = 0.0;
    const int ph = p < P ? GPH_PHASES(sq_u16v(SQ, q_phases, p)) : 0;
    if (ph > 0) {
      const int nc = 4 * ph;
      double prob = 0.0;
      for (int c = 0; c < nc; c++) prob += rc[p * 4 + c];
      term = gph_log(prob / nc) * GPH_PATCOUNT_S(SQ, q_count, p);
    }
    const int cnt = P - p0 < GPH_WAVE ? P - p0 : GPH_WAVE;
    for (int k = 0; k < cnt; k++) lnl += gph_readlane64(term, k);
  }
#endif
  return lnl;
}
// prune_node() with every tree scalar already in (scalar) registers
template <class DP>
GPH_DEV void prune_node_r(int node, int l, int r, double pl, double pr, int cbn, int cbl, int cbr, int P, DP cb,
                          int q_leaf = GPH_Q_LEAF, const GphSeq SQ = GphSeq{nullptr})
{
  double ql = 1 - 4.0 * pl;
  double qr = 1 - 4.0 * pr;
  const int nint = g_lay.n - 1;
  DP pc = cb + ((cbn * nint + (node - g_lay.n)) * P) * 4;
  DP lc = cb + (l >= g_lay.n ? ((cbl * nint + (l - g_lay.n)) * P) * 4 : 0);
  DP rc = cb + (r >= g_lay.n ? ((cbr * nint + (r - g_lay.n)) * P) * 4 : 0);
  for (int idx = GPH_LANE; idx < 4 * P; idx += GPH_NLANES) {
    int p = idx >> 2, a = idx & 3;
    double v = 1.0, f;
    f = child_factor(l, lc, p, a, pl, ql, q_leaf, SQ);
    v *= f;
    f = child_factor(r, rc, p, a, pr, qr, q_leaf, SQ);
    v *= f;
    pc[idx] = v;
  }
  GPH_WAVE_FENCE();
}

#if GPH_DEVFORMS
// ---- lanes = patterns (P <= 64): one lane owns pattern `lane` and its 4 base entries.
// The 4 conditionals a lane just produced stay in its registers (q0..q3): when the next node
// processed is the parent (the usual case: a dirty path is a chain), that child is not re-read
// from memory -- no store->load round trip on the critical path.
#ifdef GPH_HOSTEMU
struct alignas(16) gph_d2 { double x, y; };      /* (g++ has no ext_vector_type) */
#else
typedef double gph_d2 __attribute__((ext_vector_type(2)));
#endif
typedef GPH_GLB gph_d2 gdbl2;

// factors of one child for the 4 bases of pattern `lane` (computeSubtreeConditionals_new,
// LocusDataLikelihood.c:1650-1673; same operations in the same order as child_factor())
template <class CP, class CP2>
GPH_DEVHOT void child_factor4(int child, CP cnd, bool fwd, double q0, double q1, double q2, double q3,
                              double pe, double qe, int lc, double &f0, double &f1, double &f2, double &f3,
                              int q_leaf)
{
  /* lc = min(lane, P - 1): lanes beyond the last pattern repeat its (unconditional, in-range) loads -- their results
   * are never stored or summed, and no execution mask / fill value is needed around the loads */
  if (child < g_lay.n) {
    /* leaf: one-hot (or N).  S = 1 exactly, so S*pe = pe and sa*qe is qe or 0: bit-identical shortcut */
    const int code = GPH_LEAFCODE(q_leaf, lc, child);
    const double hit = pe + qe;
    const double other = code == 4 ? 1.0 : pe;   /* N: every base gets 1.0 (no code matches below) */
    f0 = code == 0 ? hit : other;
    f1 = code == 1 ? hit : other;
    f2 = code == 2 ? hit : other;
    f3 = code == 3 ? hit : other;
    return;
  }
  double s0 = q0, s1 = q1, s2 = q2, s3 = q3;
  if (!fwd) {
    CP2 c2 = (CP2)(cnd + 4 * lc);
    const gph_d2 a = c2[0], b = c2[1];
    s0 = a.x; s1 = a.y; s2 = b.x; s3 = b.y;
  }
  double S = s0;     /* 0.0 + s0: conditionals are never -0.0 */
  S += s1;
  S += s2;
  S += s3;
  /* The skip of LocusDataLikelihood.c:1660-1663 (a son whose four conditionals add up to 4 -- a subtree of N only -- leaves
   * the parent's product alone: factor 1.0) needs NO select here.  Such a son's conditionals are four exact 1.0 (leaves
   * carry 1.0 for N, products of 1.0 are 1.0), so S = 4 exactly, S * pe = 4 pe exactly, s * qe = qe, and
   * fl(4 pe + qe) with qe = fl(1 - 4 pe) is 1.0: 1 - 4 pe = qe + d with |d| <= ulp(qe) / 2 <= 2^-54, and 1 - d rounds to
   * 1.0 for every such d (below 1 the spacing is 2^-53 and a tie goes to the even 1.0; above it 2^-52).  The general
   * formula therefore gives the reference's factor bit for bit; the two compares and eight selects per node step that
   * spelt the skip out were 5 % of the kernel's vector instructions (tools/bbcount.sh).  [pe in [0, 1/4]; -ffp-contract=off] */
  const double Sp = S * pe;
  f0 = Sp + s0 * qe;
  f1 = Sp + s1 * qe;
  f2 = Sp + s2 * qe;
  f3 = Sp + s3 * qe;
}

// factors of a child that is NOT in the registers: a leaf (base code) or an internal node's array at cb + off
template <class DP, class DP2>
GPH_DEVHOT void child_generic4(int child, DP cb, int off, double pe, double qe, int lc, double &f0, double &f1, double &f2,
                               double &f3, int q_leaf)
{
  child_factor4<DP, DP2>(child, cb + off, false, 0.0, 0.0, 0.0, 0.0, pe, qe, lc, f0, f1, f2, f3, q_leaf);
}
// factors of the child whose conditionals are still in the registers, in place
GPH_DEVHOT void child_inplace4(double &s0, double &s1, double &s2, double &s3, double pe, double qe)
{
  double S = s0;
  S += s1;
  S += s2;
  S += s3;
  const double Sp = S * pe;          /* (no select for a son of N only: see child_factor4) */
  s0 = Sp + s0 * qe;
  s1 = Sp + s1 * qe;
  s2 = Sp + s2 * qe;
  s3 = Sp + s3 * qe;
}

// recompute node `node`; on entry q* hold node `prev`'s conditionals (prev < 0: nothing), on exit
// this node's.  po / lo / ro = offsets (in doubles) of the node's and its children's current arrays.
// Three straight code paths -- left child in the registers, right child in the registers, neither -- so that the
// forwarded conditionals are used where they are (no register copies to pick an operand); f_right * f_left is
// f_left * f_right bit for bit (IEEE multiplication commutes).
template <class DP, class DP2>
GPH_DEVHOT void prune_node_q(int l, int r, double pl, double pr, int po, int lo, int ro,
                             int P, int lc, DP cb, int prev,
                             double &q0, double &q1, double &q2, double &q3, int q_leaf = GPH_Q_LEAF)
{
  const double ql = 1 - 4.0 * pl;
  const double qr = 1 - 4.0 * pr;
  const int lane = GPH_LANE;
  const bool act = lane < P;
#ifdef GPH_BOUNDS
  /* the three arrays of the step inside the locus's [2][n-1][P][4] doubles (checked build only) */
  /* (a LEAF child has no array: its offset is whatever the caller's lane arithmetic gave and is never used) */
  { const int ext_ = 2 * (g_lay.n - 1) * P * 4 - 4 * P + 1; po = GPH_IX(po, ext_); if (l >= g_lay.n) lo = GPH_IX(lo, ext_); if (r >= g_lay.n) ro = GPH_IX(ro, ext_); }
#endif
  DP pc = cb + po;
  /* a child recomputed earlier in this evaluation is re-read after its stores: memory operations of one
   * wavefront are performed in order, only the compiler must not reorder them (no instruction) */
  GPH_WAVE_FENCE();
  double g0, g1, g2, g3;
#ifndef GPH_NOMASK_PRUNE
  /* round 6: the step's loads, fp64 arithmetic and store under EXEC = the lanes that own a pattern.  The lanes beyond P used
   * to compute on repeated data and throw the result away -- no instruction less, but 64 instead of P lanes of fp64 units
   * switching in a kernel the chip clocks down (2.27 GHz under k_sweep, 2.41 under the evaluate kernels): masked, -0.6 % sweep
   * time in two interleaved A/B pairs (profiles/r06_ab_mask_prune.txt).  No cross-lane operation inside. */
  if (act) {
#endif
  if (l == prev) {
    child_inplace4(q0, q1, q2, q3, pl, ql);
    child_generic4<DP, DP2>(r, cb, ro, pr, qr, lc, g0, g1, g2, g3, q_leaf);
  } else if (r == prev) {
    child_inplace4(q0, q1, q2, q3, pr, qr);
    child_generic4<DP, DP2>(l, cb, lo, pl, ql, lc, g0, g1, g2, g3, q_leaf);
  } else {
    child_generic4<DP, DP2>(l, cb, lo, pl, ql, lc, q0, q1, q2, q3, q_leaf);
    child_generic4<DP, DP2>(r, cb, ro, pr, qr, lc, g0, g1, g2, g3, q_leaf);
  }
  q0 = q0 * g0;
  q1 = q1 * g1;
  q2 = q2 * g2;
  q3 = q3 * g3;
#ifndef GPH_NOMASK_PRUNE
  {
#else
  if (act) {
#endif
    DP2 o2 = (DP2)(pc + 4 * lane);
    gph_d2 a = {q0, q1}, b = {q2, q3};
    o2[0] = a;
    o2[1] = b;
  }
#ifndef GPH_NOMASK_PRUNE
  }
#endif
}

#endif   /* GPH_DEVFORMS */
// GPH_EMU_LANES: the host build with the micro-wave AND one node per lane (up to 32 leaves): the lane-per-node device forms run
// on it; with the big-tree capacities the micro-wave runs the list-driven device forms instead
#if defined(GPH_EMU64) && !GPH_BIG_TREE
#define GPH_EMU_LANES 1
#else
#define GPH_EMU_LANES 0
#endif
#if GPH_LANE_NODES || GPH_EMU_LANES
#ifdef GPH_EMU64
#define GPH_LIK_COMPUTE_LANES lik_compute_lanes      /* next to the host form; lik_compute() below dispatches */
#else
#define GPH_LIK_COMPUTE_LANES lik_compute
#endif
// computeLocusDataLikelihood, LocusDataLikelihood.c:426-483.  Device form: the genealogy
// (father/left/right/age, one node per lane) and the dirty / current-buffer sets (64-bit
// masks) are pulled into registers once; "which nodes must be recomputed" is a ballot
// fix-point over the tree instead of the recursion of computeConditionalJC_new (:1559-1636),
// nodes are processed as soon as their recomputed children are done (any such order gives
// bit-identical conditionals), copyNodeConditionals bookkeeping is applied to the masks and
// written back once, and the per-pattern terms are added in pattern order through v_readlane.
GPH_DEVHOT double GPH_LIK_COMPUTE_LANES(int useOld, bool warm = false)
{
  const int n = g_lay.n, N = g_lay.N, lane = GPH_LANE;
  useOld = RFL(useOld);
  const int P = CNT(CN_P);
  if (P == 0) return 0.0;
  const int q_phases = CNT(CN_QPH), q_count = CNT(CN_QCNT), q_terms = CNT(CN_QTERMS);
  (void)q_terms;
  STAMPB_BEGIN(2);
  const bool isnode = lane < N;
  GphNode me = {0.0, -1, -1, -1, -1};
  if (isnode) me = gph_lds.nd[lane];    /* the lane's node record: one 16-byte read */
  const int le = me.left, ri = me.right;
  const double ag = me.age;
  uint64_t dirty = m_dirty_get(), cbit = m_cbit_get();
  const uint64_t internal = (((uint64_t)1 << N) - 1) & ~(((uint64_t)1 << n) - 1);
  uint64_t need, todo;
  if (!useOld) { cbit ^= internal & ~dirty; dirty |= internal; }
  setFS(FS_SV_DATALNL, FS(FS_DATALNL));
  if (!useOld) {
    need = internal;
  } else {
    need = dirty;
    /* le / ri of a lane that is not an internal node are -1 or a leaf's: whatever their shifts give is masked by `internal`,
     * a scalar AND of the ballot instead of two more per-lane conditions in it */
    const int le6 = le & 63, ri6 = ri & 63;
    for (int it = 0; it <= N; it++) {
      const uint64_t nn = need | (__ballot((((need >> le6) | (need >> ri6)) & 1) != 0) & internal);
      if (nn == need) break;
      need = nn;
    }
    setCNT(CN_EVALS, CNT(CN_EVALS) + 1);
  }
  const int root = ISC(IS_ROOT);
  if (!((need >> root) & 1)) { if (useOld) setCNT(CN_EMPTY, CNT(CN_EMPTY) + 1); return FS(FS_DATALNL); }
  todo = need & internal;
  const int nord = __builtin_popcountll(todo);
  gdbl *cb = cond_base();
  const double mut = FS(FS_MUTRATE);
  /* edge transition probabilities (computeEdgeConditionalJC, LocusDataLikelihood.c:1831-1848) of every
   * edge below a node that is recomputed: one lane per child node, all edges in ONE vector exp */
  double pe = 0.0;
  {
    const int fa = me.father;
    if (fa >= 0 && ((todo >> fa) & 1)) pe = edge_prob_v(mut * (gph_lds.nd[fa].age - ag));
  }
  const bool wide = P > GPH_WAVE;   /* more than one pattern per lane: generic (pattern, base) mapping */
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  int prev = -1;
  const int lc = lane < P ? lane : P - 1;
  /* copyNodeConditionals (LocusDataLikelihood.c:1889) of every node that is going to be recomputed, all at once:
   * a node not yet dirty in this proposal switches to its other array */
  if (useOld) { const uint64_t flip = todo & ~dirty; dirty |= flip; cbit ^= flip; }
  setpad64(IS_DIRTY0, dirty);
  setpad64(IS_CBIT0, cbit);
  /* offset (in doubles) of the lane's node's current array */
  const int coff = (((int)((cbit >> lane) & 1)) * (n - 1) + (lane - n)) * P * 4;
  const int fal = me.father;
  /* conditionals written by an earlier evaluation of this wave must have landed before they are re-read */
  GPH_WAVE_FENCE();
  if (warm && !wide) {
    /* (only kernels that evaluate ONE proposal per locus ask for this; in the sweep kernel, whose conditionals are
     * warm in L2, the same prefetch costs 4 % -- measured)
     * a kernel that evaluates ONE proposal per locus finds the conditionals it reads (the children of recomputed
     * nodes that are not recomputed themselves) cold in HBM, one dependent miss per node of the path.  Touch
     * them all with a single load first -- 8 lanes per array, one 128-byte line each -- so the misses overlap. */
    uint64_t sm = __ballot(isnode && lane >= n && !((todo >> lane) & 1) && fal >= 0 && ((todo >> fal) & 1));
    const int last = 32 * P - 8, mine = (lane & 7) * 128;
    int off = -1;
    for (int j = 0; sm != 0 && j < 8; j++) {
      const int sn = __builtin_ctzll(sm);
      sm &= sm - 1;
      const int c = __builtin_amdgcn_readlane(coff, sn);
      if ((lane >> 3) == j && mine < last + 128) off = c * 8 + (mine < last ? mine : last);
    }
    if (off >= 0) (void)*(const volatile GPH_GLB int *)((const GPH_GLB char *)cb + off);
  }
  STAMPB_END(2);
  /* the two mappings are separate loops (the choice is per locus): the per-pattern one keeps the conditionals of the
   * node just computed in registers from step to step, and its loop must stay simple enough for that */
  bool bad = false;
  GphSeq SQ; SQ.g = nullptr;
  if (wide) {
    SQ = seq_ref();
    for (int guard = 0; todo != 0; guard++) {
      bool rdy = isnode && ((todo >> lane) & 1) && (le < n || !((todo >> le) & 1)) && (ri < n || !((todo >> ri) & 1));
      uint64_t rmask = __ballot(rdy);
      if (rmask == 0 || guard > N) { gph_fail(100); return FS(FS_DATALNL); }
      while (rmask) {
        const int node = __builtin_ctzll(rmask);
        const uint64_t bit = (uint64_t)1 << node;
        rmask &= rmask - 1;
        const int l = __builtin_amdgcn_readlane(le, node), r = __builtin_amdgcn_readlane(ri, node);
        STAMP_BEGIN(7);
        prune_node_r<gdbl *>(node, l, r, gph_readlane64(pe, l), gph_readlane64(pe, r), (int)((cbit >> node) & 1),
                             (int)((cbit >> l) & 1), (int)((cbit >> r) & 1), P, cb, GPH_Q_LEAF, SQ);
        STAMP_END(7);
        todo &= ~bit;
      }
    }
  } else {
    /* one node per step.  Usual case: a dirty path is a chain, the next node is the father of the one just
     * computed (whose conditionals are still in registers) -- three lane reads and a bit test.  Otherwise
     * (start, or the father waits for its other subtree) any node whose recomputed children are done. */
    /* ONE exit: every step takes a node off `todo`, so the loop ends whatever the tree looks like; a step that finds no
     * ready node (a corrupted tree) takes the lowest one and the failure is reported behind the loop.  (As a `break` or a
     * `return` the check was a second exit, and the structurizer -- this loop contains the lane-masked store -- paid for
     * it with an exit flag set, inverted and tested in EVERY step.) */
    while (todo != 0) {
      int node = -1, l = 0, r = 0;
      if (prev >= 0) {
        const int f = __builtin_amdgcn_readlane(fal, prev);
        if (f >= 0) {
          l = __builtin_amdgcn_readlane(le, f);
          r = __builtin_amdgcn_readlane(ri, f);
          const int sib = l == prev ? r : l;
          if (!((todo >> sib) & 1)) node = f;        /* (a leaf is never in `todo`: no separate test for sib < n) */
        }
      }
      if (node < 0) {
        /* lanes in `todo` are internal nodes: their le / ri are node ids, and a leaf's bit of `todo` is 0 */
        bool rdy = ((todo >> lane) & 1) && !((todo >> (le & 63)) & 1) && !((todo >> (ri & 63)) & 1);
        uint64_t rmask = __ballot(rdy);
        if (rmask == 0) { bad = true; rmask = todo; }
        node = __builtin_ctzll(rmask);
        l = __builtin_amdgcn_readlane(le, node);
        r = __builtin_amdgcn_readlane(ri, node);
      }
      STAMP_BEGIN(7);
      /* the two edge probabilities of the step: lane l's and lane r's value of `pe` in EVERY lane -- through the LDS crossbar
       * (ds_bpermute_b32 with a uniform lane: no storage, no vector-pipe cycles) instead of four lane reads with the lane in a
       * scalar register, 8.3 vector-pipe cycles each (profiles/r04_class_probe.txt) */
#if !defined(GPH_PE_READLANE)      /* -0.4 % sweep time, two interleaved A/B pairs (profiles/r05_ab_pe_bpermute.txt); the offsets of the three arrays the same way: +0.3 %, not kept */
      const double pl_ = gph_bcast64(pe, l), pr_ = gph_bcast64(pe, r);
#else
      const double pl_ = gph_readlane64(pe, l), pr_ = gph_readlane64(pe, r);
#endif
#if defined(GPH_PE_BPERMUTE) && GPH_PE_BPERMUTE >= 2
      prune_node_q<gdbl *, gdbl2 *>(l, r, pl_, pr_, gph_bcast32(coff, node), gph_bcast32(coff, l), gph_bcast32(coff, r), P, lc, cb, prev,
                                    q0, q1, q2, q3);
#else
      prune_node_q<gdbl *, gdbl2 *>(l, r, pl_, pr_, __builtin_amdgcn_readlane(coff, node),
                                    __builtin_amdgcn_readlane(coff, l), __builtin_amdgcn_readlane(coff, r), P, lc, cb, prev,
                                    q0, q1, q2, q3);
#endif
      STAMP_END(7);
      todo &= ~((uint64_t)1 << node);
      prev = node;
    }
  }
  if (bad) { gph_fail(100); return FS(FS_DATALNL); }
  setCNT(CN_NODES, CNT(CN_NODES) + nord);
  STAMPB_BEGIN(4);
  /* root reduction, LocusDataLikelihood.c:466-479: per unphased pattern
   * log(sum over phases and bases / (4*phases)) * count, summed in pattern order */
  double lnl = 0.0;
  const gdbl *rc = cb + (((int)((cbit >> root) & 1) * (n - 1) + (root - n)) * P) * 4;
  if (!wide) {
    /* the root was computed last: its conditionals for pattern `lane` are q0..q3; only the further
     * phases of an unphased pattern (the following rows) come from memory */
    double term = 0.0;
    const int ph = lane < P ? gu16v(q_phases, lane) : 0;
    double prob = q0;   /* 0.0 + q0 */
    prob += q1;
    prob += q2;
    prob += q3;
    prob = add_phases(prob, ph, q0, q1, q2, q3);
    /* (the ballot sits in front of the branch: every lane takes part, the lanes without a term with `false` -- the mask the
     * branch's active lanes alone would give; wave-uniform code around every cross-lane operation is what gph_emu64.h checks) */
    const bool pow2_ = __ballot(ph > 0 && (ph & (ph - 1)) != 0) == 0;
    if (ph > 0) {
      const int nc = 4 * ph;
      /* phase counts are powers of two upstream (2^hets): the division is an exact exponent shift */
      double avg;
      if (pow2_) avg = __builtin_ldexp(prob, -(2 + __builtin_ctz(ph)));
      else avg = prob / nc;
      term = gph_log(avg) * GPH_PATCOUNT(q_count, lane);
    }
    if (CNT(CN_SUMLDS)) lnl = ordered_sum64_lds(term, P, q_terms);
    else lnl = ordered_sum64(term, P);
  } else {
    GPH_WAVE_FENCE();
    lnl = root_sum_generic(rc, P, q_phases, q_count, q_terms, SQ);
  }
  setFS(FS_DATALNL, lnl);
  if (!useOld) setCNT(CN_NODES0, CNT(CN_NODES0) + nord);     /* (the algorithmic-byte count covers useOld evaluations: out_common) */
  STAMPB_END(4);
  return lnl;
}
#endif
#if !GPH_LANE_NODES
#ifdef GPH_EMU64
#define GPH_LIK_COMPUTE_LIST lik_compute_list
#else
#define GPH_LIK_COMPUTE_LIST lik_compute
#endif
// computeLocusDataLikelihood, LocusDataLikelihood.c:426-483, with the recursion of
// computeConditionalJC_new (:1559-1636) replaced by: mark the ancestors of every
// dirty node, list the marked internal nodes parent-before-child, process the
// list backwards.  Same set of recomputed nodes, children always before parents.
GPH_DEVHOT double GPH_LIK_COMPUTE_LIST(int useOld, bool warm = false)
{
  const int n = g_lay.n, N = g_lay.N;
  (void)warm;
  useOld = RFL(useOld);
  int P = CNT(CN_P), i, node, k, sp, nord;
  gph_nset need = ns_none();
  double lnl;
  if (P == 0) return 0.0;
  const int q_phases = CNT(CN_QPH), q_count = CNT(CN_QCNT), q_terms = CNT(CN_QTERMS);
  if (!useOld)
    for (node = n; node < N; node++) lik_mark_cond(node);
  setFS(FS_SV_DATALNL, FS(FS_DATALNL));
  if (!useOld) {
    for (node = n; node < N; node++) need = ns_with(need, node);
  } else {
    for (i = 0; i < N; i++) {
      if (!ns_has(NS_GET(IS_DIRTY0), i)) continue;
      node = i;
      int guard = 0;
      while (node >= 0 && !ns_has(need, node)) {
        need = ns_with(need, node);
        node = FATH(node);
        if (++guard > N) { gph_fail(99); return FS(FS_DATALNL); }
      }
    }
  }
  if (useOld) setCNT(CN_EVALS, CNT(CN_EVALS) + 1);
  node = ISC(IS_ROOT);
  if (!ns_has(need, node)) { if (useOld) setCNT(CN_EMPTY, CNT(CN_EMPTY) + 1); return FS(FS_DATALNL); }
  /* pre-order list of needed internal nodes */
  nord = 0;
  sp = 0;
  si16(&GphLds::s_stack, sp++, node);
  while (sp > 0) {
    node = gi16(&GphLds::s_stack, --sp);
    if (nord >= N) { gph_fail(100); return FS(FS_DATALNL); }
    si16(&GphLds::s_ord, nord++, node);
    k = LEFT(node);
    if (k >= n && ns_has(need, k)) si16(&GphLds::s_stack, sp++, k);
    k = RGHT(node);
    if (k >= n && ns_has(need, k)) si16(&GphLds::s_stack, sp++, k);
  }
#if GPH_DEVFORMS && GPH_BIG_TREE
#ifdef GPH_EMU64
  if (P <= GPH_WAVE && gph_emu::in_wave()) {
#else
  if (P <= GPH_WAVE) {
#endif
    /* the big-tree build on the device, at most one pattern per lane: the ORDER of the recomputation comes from the
     * list above (there is no lane per node to schedule it by ballots), everything else is the vector form of the
     * smaller builds -- every edge probability of the evaluation in one vector exp (lanes = child nodes, two rounds),
     * lanes = patterns with the four base entries of a pattern in the lane's registers, the conditionals of the node
     * just computed forwarded in registers when the next node is its parent, per-pattern logarithms on the lanes,
     * the ordered pattern sum.  Same operations in the same order as prune_node / the loop below. */
    const int lane = GPH_LANE, nint = n - 1;
    gdbl *cb = cond_base();
    const double mut = FS(FS_MUTRATE);
    for (int c = lane; c < N; c += GPH_NLANES) {
      const int fa = gph_lds.nd[c].father;
      double pe = 0.0;
      if (fa >= 0 && ns_has(need, fa)) pe = edge_prob_v(mut * (gph_lds.nd[fa].age - gph_lds.nd[c].age));
      gph_lds.s_pe[c] = pe;
    }
    GPH_WAVE_FENCE();
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    int prev = -1;
    const int lc = lane < P ? lane : P - 1;
    for (i = nord - 1; i >= 0; i--) {
      node = gi16(&GphLds::s_ord, i);
      if (useOld) lik_mark_cond(node);
      const int l = LEFT(node), r = RGHT(node);
      const double pl = gph_lds.s_pe[l], pr = gph_lds.s_pe[r];
      const gph_nset cb_ = NS_GET(IS_CBIT0);
      const int po = (((int)ns_has(cb_, node) * nint + (node - n)) * P) * 4;
      const int lo = l >= n ? (((int)ns_has(cb_, l) * nint + (l - n)) * P) * 4 : 0;
      const int ro = r >= n ? (((int)ns_has(cb_, r) * nint + (r - n)) * P) * 4 : 0;
      prune_node_q<gdbl *, gdbl2 *>(l, r, pl, pr, po, lo, ro, P, lc, cb, prev, q0, q1, q2, q3);
      prev = node;
    }
    setCNT(CN_NODES, CNT(CN_NODES) + nord);
    /* root reduction (LocusDataLikelihood.c:466-479): the root was computed last, its conditionals of pattern `lane`
     * are q0..q3; further phases come from the next lanes' registers, in the reference's order */
    double term = 0.0;
    const int ph = lane < P ? gu16v(q_phases, lane) : 0;
    double prob = q0;
    prob += q1;
    prob += q2;
    prob += q3;
    prob = add_phases(prob, ph, q0, q1, q2, q3);
    const bool pow2_ = __ballot(ph > 0 && (ph & (ph - 1)) != 0) == 0;      /* (in front of the branch: see the lane-per-node form) */
    if (ph > 0) {
      const int nc = 4 * ph;
      double avg;
      if (pow2_) avg = __builtin_ldexp(prob, -(2 + __builtin_ctz(ph)));
      else avg = prob / nc;
      term = gph_log(avg) * GPH_PATCOUNT(q_count, lane);
    }
    lnl = ordered_sum64(term, P);
    setFS(FS_DATALNL, lnl);
    if (!useOld) setCNT(CN_NODES0, CNT(CN_NODES0) + nord);
    return lnl;
  }
#endif
  for (i = nord - 1; i >= 0; i--) {
    node = gi16(&GphLds::s_ord, i);
    if (useOld) lik_mark_cond(node);
    { STAMP_BEGIN(7); prune_node(node); STAMP_END(7); }
  }
  setCNT(CN_NODES, CNT(CN_NODES) + nord);
  /* root reduction, LocusDataLikelihood.c:466-479: per unphased pattern
   * gph_log_u(sum over phases and bases / (4*phases)) * count, summed in pattern order */
  {
    const gdbl *rc = cond_base() + cond_off(ISC(IS_ROOT), CBIT(ISC(IS_ROOT)));
    lnl = root_sum_generic(rc, P, q_phases, q_count, q_terms, seq_ref());
  }
  setFS(FS_DATALNL, lnl);
  if (!useOld) setCNT(CN_NODES0, CNT(CN_NODES0) + nord);
  return lnl;
}

#endif
#ifdef GPH_EMU64
// the host build with the micro-wave: the DEVICE form on 64 emulated lanes (gph_emu64.h) -- or, with GPH_EMU64=0 in the
// environment, the one-lane list form: the two must agree bit for bit on every golden
GPH_DEVHOT double lik_compute(int useOld, bool warm = false)
{
#if GPH_EMU_LANES
  if (!gph_emu::enabled() || g_lay.N > gph_emu::W) return lik_compute_list(useOld, warm);
  double r = 0.0;
  gph_emu::run([&] { const double v = lik_compute_lanes(useOld, warm); if (GPH_LANE == 0) r = v; });
  return r;
#else
  /* big-tree capacities: the list-driven form ITSELF on the micro-wave -- its wave-uniform part on every lane, its device block
   * (edge probabilities on a lane per child node, a lane per pattern, register forwarding) as on the gfx950 builds g / h / b / n */
  if (!gph_emu::enabled()) return lik_compute_list(useOld, warm);
  double r = 0.0;
  gph_emu::run([&] { const double v = lik_compute_list(useOld, warm); if (GPH_LANE == 0) r = v; });
  return r;
#endif
}
#endif

// ---------------------------------------------------------------- stateless full evaluation
// The value computeLocusDataLikelihood(useOld = 0) (LocusDataLikelihood.c:426-483) returns for a locus at
// mutation rate `rate`, WITHOUT touching the locus: its node records (GphNode[N]) sit at byte offset o_nd and
// its sequence block at o_seq of the dynamic LDS, the conditionals go to a single-buffer scratch array
// [n-1][P][4] (dynamic LDS at o_scr, or global memory when gscr != NULL).  Same arithmetic, same order of
// operations as lik_compute(0): the serial scan of UpdateLocusRate (kb_lrate_scan) uses it to decide, the
// accepted loci are then recomputed in place by lik_compute(0) and must reproduce the value bit for bit.
#if GPH_LANE_NODES
typedef GPH_LDS gph_d2 ld2;
template <class DP, class DP2>
GPH_DEVHOT double lik_private_t(int o_nd, int o_seq, int P, int root, double rate, DP scr)
{
  const int n = g_lay.n, N = g_lay.N, lane = GPH_LANE;
  const int q_leaf = o_seq + GPH_Q_LEAF, q_phases = o_seq + GPH_Q_PHASES(P, n), q_count = o_seq + GPH_Q_COUNT(P, n),
            q_terms = o_seq + GPH_Q_TERMS(P, n, g_lay.cnt16);
  const bool isnode = lane < N;
#ifdef GPH_LRSTAMP
  const uint64_t st0 = __builtin_readcyclecounter();
#endif
  typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));
  union { gu32x4 v; GphNode n; } nu;
  GphNode me = {0.0, -1, -1, -1, -1};
  if (isnode) { nu.v = ((GPH_LDS gu32x4 *)(GPH_SMB + o_nd))[lane]; me = nu.n; }   /* the lane's node record: one 16-byte read */
  const int le = me.left, ri = me.right;
  const double ag = me.age;
  const uint64_t internal = (((uint64_t)1 << N) - 1) & ~(((uint64_t)1 << n) - 1);
  uint64_t todo = internal;
  double pe = 0.0;
  {
    const int fa = me.father;
    if (isnode && fa >= 0) pe = edge_prob_v(rate * (((lf64 *)(GPH_SMB + o_nd))[2 * fa] - ag));   /* GphNode::age of the father */
  }
  const bool wide = P > GPH_WAVE;
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  int prev = -1;
  GPH_WAVE_FENCE();
#ifdef GPH_LRSTAMP
  pe = RFLD(pe) * 0.0 + pe;
  const uint64_t st1 = __builtin_readcyclecounter();
  gph_lds.s_cntf[1] += (double)(st1 - st0);
#endif
  for (int guard = 0; todo != 0; guard++) {
    bool rdy = isnode && ((todo >> lane) & 1) && (le < n || !((todo >> le) & 1)) && (ri < n || !((todo >> ri) & 1));
    uint64_t rmask = __ballot(rdy);
    if (rmask == 0 || guard > N) { gph_fail(100); return 0.0; }
    if (wide) {
      while (rmask) {
        const int node = __builtin_ctzll(rmask);
        rmask &= rmask - 1;
        const int l = __builtin_amdgcn_readlane(le, node), r = __builtin_amdgcn_readlane(ri, node);
        prune_node_r<DP>(node, l, r, gph_readlane64(pe, l), gph_readlane64(pe, r), 0, 0, 0, P, scr, q_leaf);
        todo &= ~((uint64_t)1 << node);
      }
    } else {
      const uint64_t pm = __ballot(isnode && (le == prev || ri == prev)) & rmask;
      const int node = __builtin_ctzll(pm ? pm : rmask);
      const uint64_t bit = (uint64_t)1 << node;
      const int l = __builtin_amdgcn_readlane(le, node), r = __builtin_amdgcn_readlane(ri, node);
      prune_node_q<DP, DP2>(l, r, gph_readlane64(pe, l), gph_readlane64(pe, r), (node - n) * P * 4, l >= n ? (l - n) * P * 4 : 0,
                            r >= n ? (r - n) * P * 4 : 0, P, lane < P ? lane : P - 1, scr, prev, q0, q1, q2, q3, q_leaf);
      todo &= ~bit;
      prev = node;
    }
  }
#ifdef GPH_LRSTAMP
  q0 = RFLD(q0) * 0.0 + q0;
  const uint64_t st2 = __builtin_readcyclecounter();
  gph_lds.s_cntf[2] += (double)(st2 - st1);
#endif
  /* root reduction, LocusDataLikelihood.c:466-479 (see lik_compute) */
  double lnl = 0.0;
  DP rc = scr + ((root - n) * P) * 4;
  if (!wide) {
    double term = 0.0;
    const int ph = lane < P ? gu16v(q_phases, lane) : 0;
    double prob = 0.0;
    prob += q0;
    prob += q1;
    prob += q2;
    prob += q3;
    prob = add_phases(prob, ph, q0, q1, q2, q3);
    if (ph > 0) {
      const int nc = 4 * ph;
      double avg;
      if (__ballot(ph > 0 && (ph & (ph - 1)) != 0) == 0) avg = __builtin_ldexp(prob, -(2 + __builtin_ctz(ph)));
      else avg = prob / nc;
      term = gph_log(avg) * GPH_PATCOUNT(q_count, lane);
    }
    lnl = ordered_sum64(term, P);
  } else {
    GPH_WAVE_FENCE();
    for (int p = lane; p < P; p += GPH_NLANES) {
      int ph = GPH_PHASES(gu16v(q_phases, p));
      if (ph > 0) {
        int nc = 4 * ph;
        double prob = 0.0;
        for (int c = 0; c < nc; c++) prob += rc[p * 4 + c];
        sf64(q_terms, p, gph_log(prob / nc) * GPH_PATCOUNT(q_count, p));
      }
    }
    GPH_SYNC();
    for (int p = 0; p < P; p++)
      if (gu16(q_phases, p) > 0) lnl += gf64(q_terms, p);
  }
#ifdef GPH_LRSTAMP
  lnl = RFLD(lnl);
  gph_lds.s_cntf[3] += (double)(__builtin_readcyclecounter() - st2);
  gph_lds.s_cntf[4] += 1.0;
#endif
  return lnl;
}
GPH_DEVHOT double lik_private(int o_nd, int o_seq, int P, int root, double rate, int o_scr, gdbl *gscr)
{
  P = RFL(P); root = RFL(root); rate = RFLD(rate);
  if (P == 0) return 0.0;
  if (gscr) return lik_private_t<gdbl *, gdbl2 *>(o_nd, o_seq, P, root, rate, gscr);
  return lik_private_t<lf64 *, ld2 *>(o_nd, o_seq, P, root, rate, (lf64 *)(GPH_SMB + o_scr));
}
#else
GPH_DEV double lik_private(int o_nd, int o_seq, int P, int root, double rate, int o_scr, gdbl *gscr)
{
  const int n = g_lay.n, N = g_lay.N;
  if (P == 0) return 0.0;
  const int q_leaf = o_seq + GPH_Q_LEAF, q_phases = o_seq + GPH_Q_PHASES(P, n), q_count = o_seq + GPH_Q_COUNT(P, n);
  const GphNode *nds = (const GphNode *)((char *)gph_sm + o_nd);
  double *scr = gscr ? (double *)gscr : (double *)(gph_sm + o_scr);
  int ord[GPH_CAP_N], st[GPH_CAP_N], nord = 0, sp = 0, i, p;
  st[sp++] = root;
  while (sp > 0) {
    int node = st[--sp];
    if (nord >= N) { gph_fail(100); return 0.0; }
    ord[nord++] = node;
    if (nds[node].left >= n) st[sp++] = nds[node].left;
    if (nds[node].right >= n) st[sp++] = nds[node].right;
  }
  for (i = nord - 1; i >= 0; i--) {
    const int node = ord[i], l = nds[node].left, r = nds[node].right;
    const double pl = edge_prob(rate * (nds[node].age - nds[l].age));
    const double pr = edge_prob(rate * (nds[node].age - nds[r].age));
    prune_node_r<double *>(node, l, r, pl, pr, 0, 0, 0, P, scr, q_leaf);
  }
  double lnl = 0.0;
  const double *rc = scr + ((root - n) * P) * 4;
  for (p = 0; p < P; p++) {
    int ph = GPH_PHASES(gu16v(q_phases, p));
    if (ph > 0) {
      int nc = 4 * ph, c;
      double prob = 0.0;
      for (c = 0; c < nc; c++) prob += rc[p * 4 + c];
      lnl += gph_log(prob / nc) * GPH_PATCOUNT(q_count, p);
    }
  }
  return lnl;
}
#endif

// scaleAllNodeAges, LocusDataLikelihood.c:895-917
GPH_DEV double lik_scale_ages(double factor)
{
  int node;
  double old = FS(FS_DATALNL);
  for (node = 0; node < g_lay.N; node++) lik_adjust_age(node, factor * AGE(node));
  lik_compute(1);
  return FS(FS_DATALNL) - old;
}

// executeGenSPR, LocusDataLikelihood.c:931-1012
// The node sets of the saved version (recomputed / current half / saved record) are read from the scalar pad ONCE, updated in
// scalar registers for the up to five nodes an SPR saves, and written back once (every copyNodeToSaved /
// copyNodeConditionals used to be its own read-modify-write of the pad: a dozen lane reads / writes per node)
GPH_DEV int lik_spr(int subtreeRoot, int target, double age)
{
  int targetFather = FATH(target);
  int father = FATH(subtreeRoot);
  const GphNodeS F_ = ld_node(father);
  int grandpa = F_.father;
  int sibling = F_.left + F_.right - subtreeRoot;
  gph_nset dirty = NS_GET(IS_DIRTY0), cbit = NS_GET(IS_CBIT0), saved = NS_GET(IS_SAVED0);
  const bool hasP = CNT(CN_P) > 0;
  int ret = 0;
#define SPR_MARK(node) do { if (hasP && !ns_has(dirty, (node))) { dirty = ns_with(dirty, (node)); cbit = ns_flip(cbit, (node)); } } while (0)
#define SPR_SAVE(node, recalc) do { if (recalc) SPR_MARK(node); saved = ns_with(saved, (node)); gph_lds.sv[node] = gph_lds.nd[node]; } while (0)
  SPR_SAVE(father, 1);              /* adjustGenNodeAge(father, age) */
  setAGE(father, age);
  if (!(target == sibling || target == father)) {
    SPR_SAVE(sibling, 0);
    setFATH(sibling, grandpa);
    if (grandpa >= 0) {
      SPR_SAVE(grandpa, 1);
      if (LEFT(grandpa) == father) setLEFT(grandpa, sibling);
      else setRGHT(grandpa, sibling);
    }
    setFATH(father, targetFather);
    setLEFT(father, subtreeRoot);
    setRGHT(father, target);
    if (target != grandpa) SPR_SAVE(target, 0);
    setFATH(target, father);
    if (targetFather < 0) {
      setISC(IS_SV_ROOT, target);
      setISC(IS_ROOT, father);
      ret = 1;
    } else {
      if (targetFather == sibling) SPR_MARK(targetFather);
      else if (targetFather != grandpa) SPR_SAVE(targetFather, 1);
      if (LEFT(targetFather) == target) setLEFT(targetFather, father);
      else setRGHT(targetFather, father);
      if (grandpa < 0) {
        setISC(IS_SV_ROOT, father);
        setISC(IS_ROOT, sibling);
        ret = 2;
      }
    }
  }
#undef SPR_MARK
#undef SPR_SAVE
  NS_PUT(IS_DIRTY0, dirty);
  NS_PUT(IS_CBIT0, cbit);
  NS_PUT(IS_SAVED0, saved);
  return ret;
}

// ---------------------------------------------------------------- migration-node lookups
// findLastMig / findFirstMig, patch.c:374-414
GPH_DEV int find_last_mig(int node, double age)
{
  int i, mig, last = -1, nm = ISC(IS_NUM_MIGS);
  for (i = 0; i < nm; i++) {
    mig = LIVING(i);
    if (MG(mig, MG_BRANCH) != node) continue;
    if ((age < 0 || MAGE(mig) < age) && (last < 0 || MAGE(mig) > MAGE(last))) last = mig;
  }
  return last;
}
GPH_DEV int find_first_mig(int node, double age)
{
  int i, mig, first = -1, nm = ISC(IS_NUM_MIGS);
  for (i = 0; i < nm; i++) {
    mig = LIVING(i);
    if (MG(mig, MG_BRANCH) != node) continue;
    if (MAGE(mig) > age && (first < 0 || MAGE(mig) < MAGE(first))) first = mig;
  }
  return first;
}
// findFirstMig(inode, -1), findLastMig(left, -1), findLastMig(right, -1) (patch.c:374-414) in ONE pass over the
// locus's migration nodes: the three searches of UpdateGB_InternalNode's bounds (GPhoCS.c:2334-2352) filter on three
// different branches, so every migration node feeds at most one of them and the results are the ones of three passes
GPH_DEV void mig_bounds(int inode, int left, int right, int &first_up, int &last_l, int &last_r)
{
  int i, mig, br, nm = ISC(IS_NUM_MIGS);
  /* three scalars updated by selects: with one conditional store per result the compiler sinks the three stores into ONE
   * through a selected address, which turns the results into a stack array -- scratch memory in the bounds of every node */
  int fu = -1, ll = -1, lr = -1;
  for (i = 0; i < nm; i++) {
    mig = LIVING(i);
    br = MG(mig, MG_BRANCH);
    const bool up = br == inode, lf = !up && br == left, rt = !up && !lf && br == right;
    if (!(up || lf || rt)) continue;
    const int cur = up ? fu : lf ? ll : lr;
    const double a = MAGE(mig);
    bool take = cur < 0;
    if (!take) take = up ? a < MAGE(cur) : a > MAGE(cur);
    if (up && !(a > -1)) take = false;
    if (take) { fu = up ? mig : fu; ll = lf ? mig : ll; lr = rt ? mig : lr; }
  }
  first_up = fu; last_l = ll; last_r = lr;
}
// getEdgesForTimePop, patch.c:526-571 (targets written to s_targets, increasing node id).
// Device form: one lane per genealogy node evaluates the membership test, a ballot yields the
// candidate set in node order (the order decides which edge a sampled coalescence picks).
GPH_DEV int edges_for_time_pop(double time, int pop, int exc)
{
  int node, mig, pop1, num = 0, f;
  if (UNI(g_model.popAge[pop] > time + 0.0000001)) return 0;
#ifdef GPH_EMU64
  /* the host build with the micro-wave: the device form below on 64 emulated lanes (it recurses once: inside the wave) */
  if (gph_emu::enabled() && (GPH_BIG_TREE || g_lay.N <= gph_emu::W)) {
    int r = 0;
    gph_emu::run([&] { const int v = edges_for_time_pop(time, pop, exc); if (GPH_LANE == 0) r = v; });
    return r;
  }
#endif
#if GPH_LANE_NODES || GPH_EMU_LANES
#ifdef GPH_EMU64
  if (gph_emu::in_wave())
#endif
  {
    const int lane = GPH_LANE;
    bool in = false;
    if (lane < g_lay.N && lane != exc) {
      f = gph_lds.nd[lane].father;
      in = !(gph_lds.nd[lane].age > time) && !(f >= 0 && gph_lds.nd[f].age <= time);
      if (in && pop != g_lay.rootPop) {
        /* findLastMig(node = lane, time), patch.c:374-391 */
        int last = -1, nm = ISC(IS_NUM_MIGS);
        for (int i = 0; i < nm; i++) {
          mig = LIVING(i);
          if (MG(mig, MG_BRANCH) != lane) continue;
          if ((time < 0 || MAGE(mig) < time) && (last < 0 || MAGE(mig) > MAGE(last))) last = mig;
        }
        pop1 = (last >= 0) ? (int)gph_lds.mig_i[last * MG_COUNT + MG_SPOP] : (int)gph_lds.nd[lane].npop;
        in = ((g_model.isAnc[pop] >> pop1) & 1) != 0;
      }
    }
    uint64_t m = __ballot(in);
    num = __builtin_popcountll(m);
    if (in) gph_lds.s_targets[__builtin_popcountll(m & (((uint64_t)1 << lane) - 1))] = (int16_t)lane;
    GPH_SYNC();
    return num;
  }
#endif
#if GPH_BIG_TREE && GPH_DEVFORMS
  /* the big-tree build on the device: the same membership test with a lane per node, 64 nodes a round; the candidates
   * of a round go behind those of the rounds before it (node order, as the reference lists them) */
#ifdef GPH_EMU64
  if (gph_emu::in_wave())
#endif
  {
    const int lane = GPH_LANE;
    for (int base = 0; base < g_lay.N; base += GPH_NLANES) {
      const int nd_ = base + lane;
      bool in = false;
      if (nd_ < g_lay.N && nd_ != exc) {
        f = gph_lds.nd[nd_].father;
        in = !(gph_lds.nd[nd_].age > time) && !(f >= 0 && gph_lds.nd[f].age <= time);
        if (in && pop != g_lay.rootPop) {
          int last = -1, nm = ISC(IS_NUM_MIGS);
          for (int i = 0; i < nm; i++) {
            mig = LIVING(i);
            if (MG(mig, MG_BRANCH) != nd_) continue;
            if ((time < 0 || MAGE(mig) < time) && (last < 0 || MAGE(mig) > MAGE(last))) last = mig;
          }
          pop1 = (last >= 0) ? (int)gph_lds.mig_i[last * MG_COUNT + MG_SPOP] : (int)gph_lds.nd[nd_].npop;
          in = ((g_model.isAnc[pop] >> pop1) & 1) != 0;
        }
      }
      const uint64_t m = __ballot(in);
      if (in) gph_lds.s_targets[num + __builtin_popcountll(m & (((uint64_t)1 << lane) - 1))] = (int16_t)nd_;
      num += __builtin_popcountll(m);
    }
    GPH_SYNC();
    return num;
  }
#endif
#ifdef GPH_HOSTEMU
  for (node = 0; node < g_lay.N; node++) {
    f = FATH(node);
    if (node == exc || UNI(AGE(node) > time) || (f >= 0 && UNI(AGE(f) <= time))) continue;
    if (pop == g_lay.rootPop) { si16(&GphLds::s_targets, num++, node); continue; }
    mig = find_last_mig(node, time);
    pop1 = (mig >= 0) ? MG(mig, MG_SPOP) : NPOP(node);
    if ((g_model.isAnc[pop] >> pop1) & 1) si16(&GphLds::s_targets, num++, node);
  }
  return num;
#endif
}

// ---------------------------------------------------------------- event chain
// removeEvent, patch.c:1666-1700.  pop = the population whose chain holds the event: upstream finds it, when the event
// heads its chain, by walking to the chain's END_CHAIN event (:1680-1686); every caller here knows it (the host build
// of the tests still walks and checks the caller's claim on every golden)
GPH_DEV void remove_event(int ev, int pop)
{
  const GphEvS R = ld_ev(ev);
  int nx = R.next, pv = R.prev;
  setEVT(nx, EVT(nx) + R.time);
  setEPREV(nx, pv);
  if (pv < 0) {
#ifdef GPH_HOSTEMU
    int guard = 0;
    for (pv = nx; ETYPE(pv) != GPH_END_CHAIN; pv = ENEXT(pv)) { if (++guard > GPH_CAP_E || ENEXT(pv) < 0) { gph_fail(91); return; } }
    if (ENODE(pv) != pop) { gph_fail(91); return; }
#endif
    setFIRSTEV(pop, nx);
  } else {
    setENEXT(pv, nx);
  }
  nx = ISC(IS_FREE);
  setENEXT(ev, nx);
  setEPREV(nx, ev);
  setISC(IS_FREE, ev);
  setEVT(ev, 0);
  setENLIN(ev, 0);
  setENODE(ev, -1);
}
// createEventBefore, patch.c:1707-1742
// R = the record of `ev` (the caller has just read it)
GPH_DEVHOT int create_event_before(int pop, int ev, double elapsed, const GphEvS &R)
{
  int pv = R.prev, nw = ISC(IS_FREE);
  const int fnext = ENEXT(nw);
  setISC(IS_FREE, fnext);
  if (fnext < 0) { gph_fail(15); }
  setENEXT(nw, ev);
  setEPREV(nw, pv);
  setENLIN(nw, R.nlin);
  setEVT(nw, elapsed);
  setETYPE(nw, GPH_DUMMY);
  setEPREV(ev, nw);
  setEVT(ev, R.time - elapsed);
  if (pv < 0) setFIRSTEV(pop, nw);
  else setENEXT(pv, nw);
  return nw;
}
GPH_DEV int create_event_before(int pop, int ev, double elapsed) { return create_event_before(pop, ev, elapsed, ld_ev(ev)); }
// createEvent, patch.c:1753-1802
GPH_DEVHOT int create_event(int pop, double age)
{
  int ev;
  pop = RFL(pop);
  double dt = age - g_model.popAge[pop];
  if (UNI(dt < 0)) return -1;
  if (pop != g_lay.rootPop && UNI(age > g_model.popAge[g_model.popFather[pop]] + 0.000001)) return -1;
  int guard = 0;
  ev = FIRSTEV(pop);
  GphEvS R = ld_ev(ev);
  while (R.type != GPH_END_CHAIN && UNI(R.time < dt)) {
    dt -= R.time;
    if (++guard > GPH_CAP_E || R.next < 0) { gph_fail(92); return -1; }
    ev = R.next;
    R = ld_ev(ev);
  }
  if (UNI(R.time < dt)) {
    if (UNI(R.time < dt - 0.000001)) { gph_fail(18); return -1; }
    dt = R.time;
  }
  return create_event_before(pop, ev, dt, R);
}

// recalcStats, patch.c:2387-2513.  The reference also patches the global totals
// here; the engine instead re-reduces the per-locus statistics after the kernel.
GPH_DEVHOT double recalc_stats(int pop)
{
  pop = RFL(pop);
  int n, id, b, ev, nc = 0;
  LiveList live = {0, 0};
  double t, delta = 0.0, cs = 0.0;
  int guard = 0;
  ev = FIRSTEV(pop);
  n = ENLIN(ev);
  int nxt;
  for (; ev >= 0; ev = nxt) {
    if (++guard > GPH_CAP_E) { gph_fail(93); return 0.0; }
    const GphEvS R = ld_ev(ev);
    nxt = R.next;
    setENLIN(ev, n);
    id = R.node;
    t = R.time;
    cs += n * (n - 1) * t;
    for (b = 0; b < live.n; b++) sf64(&GphLds::s_chkmig, ll_get(live, b), gf64(&GphLds::s_chkmig, ll_get(live, b)) + n * t);
    switch (R.type) {
    case GPH_SAMPLES_START: n += g_model.samplesPerPop[pop]; break;
    case GPH_COAL: nc++; n--; break;
    case GPH_IN_MIG: {
      int bb = MG(id, MG_BAND);
      si16(&GphLds::s_chknm, bb, gi16(&GphLds::s_chknm, bb) + 1);
      n--;
      break;
    }
    case GPH_OUT_MIG: n++; break;
    case GPH_MIG_BAND_START:
      ll_push(live, id);
      si16(&GphLds::s_chknm, id, 0);
      sf64(&GphLds::s_chkmig, id, 0.0);
      break;
    case GPH_MIG_BAND_END:
      delta -= (gf64(&GphLds::s_chkmig, id) - MIGST(id)) * g_model.migRate[id];
      setMIGST(id, gf64(&GphLds::s_chkmig, id));
      setNMIGB(id, gi16(&GphLds::s_chknm, id));
      b = ll_find(live, id);
      if (b == live.n) { gph_fail(25); return 0.0; }
      ll_swap_remove(live, b);
      break;
    case GPH_DUMMY:
    case GPH_END_CHAIN: break;
    default: gph_fail(26); return 0.0;
    }
  }
  if (live.n != 0) { gph_fail(27); return 0.0; }
  delta -= gph_div_theta(cs - COALS(pop), pop);
  setCOALS(pop, cs);
  setNCOAL(pop, nc);
  return delta;
}

// computeGenetreeStats, patch.c:2330-2354
GPH_DEV void compute_genetree_stats()
{
  int i, pop;
  for (i = 0; i < g_lay.K; i++) {
    pop = g_model.postOrder[i];
    if (pop >= g_lay.Kc)
      setENLIN(FIRSTEV(pop), ENLIN(g_model.popSon0[pop]) + ENLIN(g_model.popSon1[pop]));
    else
      setENLIN(FIRSTEV(pop), 0);
    recalc_stats(pop);
  }
}

// gtreeLnLikelihood, patch.c:2702-2738 (no admixture)
GPH_DEV double gtree_lnl()
{
  int pop, b;
  double lnLd = 0, theta, rate;
  for (pop = 0; pop < g_lay.K; pop++) {
    theta = g_model.theta[pop];
    lnLd += NCOAL(pop) * g_model.logTwoTheta[pop] - COALS(pop) / (theta);
  }
  for (b = 0; b < g_lay.B; b++) {
    rate = g_model.migRate[b];
    if (rate > 0.0) lnLd += NMIGB(b) * g_model.logMigRate[b] - MIGST(b) * rate;
  }
  return lnLd;
}

// constructEventChain, patch.c:1961-2125
GPH_DEV void construct_event_chain()
{
  const int K = g_lay.K, E = g_lay.E;
  int i, pop, mig, node, ev, b;
  double age;
  for (pop = 0; pop < K; pop++) {
    setETYPE(pop, GPH_END_CHAIN);
    setENEXT(pop, -1);
    setEPREV(pop, -1);
    setENODE(pop, pop);
    setENLIN(pop, 0);
    if (pop == g_lay.rootPop) setEVT(pop, GPH_OLDAGE - g_model.popAge[g_lay.rootPop]);
    else setEVT(pop, g_model.popAge[g_model.popFather[pop]] - g_model.popAge[pop]);
    setFIRSTEV(pop, pop);
  }
  setISC(IS_FREE, K);
  setEPREV(K, -1);
  setENEXT(E - 1, -1);
  for (ev = K; ev < E - 1; ev++) { setENEXT(ev, ev + 1); setEPREV(ev + 1, ev); }
  for (b = 0; b < g_lay.B; b++) {
    pop = g_model.bandTgt[b];
    ev = create_event(pop, g_model.bandStart[b]);
    if (ev < 0) { gph_fail(20); return; }
    setETYPE(ev, GPH_MIG_BAND_START);
    setENODE(ev, b);
    ev = create_event(pop, g_model.bandEnd[b]);
    if (ev < 0) { gph_fail(21); return; }
    setETYPE(ev, GPH_MIG_BAND_END);
    setENODE(ev, b);
  }
  for (pop = 0; pop < g_lay.Kc; pop++) {
    ev = create_event(pop, g_model.sampleAge[pop]);
    if (ev < 0) { gph_fail(21); return; }
    setETYPE(ev, GPH_SAMPLES_START);
  }
  for (i = 0; i < ISC(IS_NUM_MIGS); i++) {
    mig = LIVING(i);
    age = MAGE(mig);
    ev = create_event(MG(mig, MG_TPOP), age);
    if (ev < 0) { gph_fail(22); return; }
    setETYPE(ev, GPH_IN_MIG);
    setENODE(ev, mig);
    setMG(mig, MG_TEV, ev);
    ev = create_event(MG(mig, MG_SPOP), age);
    if (ev < 0) { gph_fail(23); return; }
    setETYPE(ev, GPH_OUT_MIG);
    setENODE(ev, mig);
    setMG(mig, MG_SEV, ev);
  }
  for (node = g_lay.n; node < g_lay.N; node++) {
    ev = create_event(NPOP(node), AGE(node));
    if (ev < 0) { gph_fail(24); return; }
    setETYPE(ev, GPH_COAL);
    setENODE(ev, node);
    setNEV(node, ev);
  }
}

// ---------------------------------------------------------------- considerEventMove
// computeMigStatsDelta, patch.c:1838-1864
GPH_DEV void mig_stats_delta(int inst, double bottom_age, int bottom_pop, double top_age, int dlin)
{
  int b, nb = 0;
  double dt, lo, hi;
  /* the bands whose target population is bottom_pop or above it, in band order (the filter of patch.c:1846 as one
   * table word per population) */
  for (int w = 0; w < GPH_BANDW; w++)
    for (uint32_t over = g_model.bandsOver[bottom_pop][w]; over != 0; over &= over - 1) {
      b = 32 * w + __builtin_ctz(over);
      hi = gmin2(g_model.bandEnd[b], top_age);
      lo = gmax2(g_model.bandStart[b], bottom_age);
      dt = hi - lo;
      if (dt <= 0) continue;
      setDBANDS(inst, nb, b);
      setDMIG(inst, nb, dlin * dt);
      nb++;
    }
  setDI(inst, DI_NBANDS, nb);
}
// computeCoalStatsDelta, patch.c:1878-1927
GPH_DEV void coal_stats_delta(int inst, int bottom_event, int bottom_pop, int top_event, int dlin)
{
  int pop = bottom_pop, ev = bottom_event, np = 1, ne = 0;
  double acc = 0;
  int guard = 0;
  setDPOPS(inst, 0, pop);
  while (ev >= 0) {
    if (++guard > 2 * GPH_CAP_E) { gph_fail(94); break; }
    const GphEvS R = ld_ev(ev);
    acc += dlin * (dlin - 1 + 2 * R.nlin) * R.time;
    setDEV(inst, ne, ev);
    ne++;
    if (ev == top_event) break;
    ev = R.next;
    if (ev < 0) {
      if (g_model.popFather[pop] < 0) { gph_fail(19); break; }
      setDCOAL(inst, np - 1, acc);
      acc = 0;
      pop = g_model.popFather[pop];
      ev = FIRSTEV(pop);
      setDPOPS(inst, np, pop);
      np++;
    }
  }
  setDCOAL(inst, np - 1, acc);
  setDI(inst, DI_NPOPS, np);
  setDI(inst, DI_NEV, ne);
}
// computeDeltaLnLd, patch.c:1516-1532
GPH_DEV double delta_lnld(int inst)
{
  int i, np = DI(inst, DI_NPOPS), nb = DI(inst, DI_NBANDS);
  double r = 0;
  for (i = 0; i < np; i++) r -= gph_div_theta(DCOAL(inst, i), DPOPS(inst, i));
  for (i = 0; i < nb; i++) r -= DMIG(inst, i) * g_model.migRate[DBANDS(inst, i)];
  return r;
}
// considerEventMove, patch.c:1434-1507
GPH_DEVHOT double consider_event_move(int inst, int event_id, int source_pop, double original_age,
                                     int target_pop, double new_age)
{
  int new_event, bottom_event, top_event, bottom_pop, dlin;
  double top_age, bottom_age, r;
  inst = RFL(inst); event_id = RFL(event_id); source_pop = RFL(source_pop); target_pop = RFL(target_pop);
  const int ev_type = ETYPE(event_id);     /* (read once: creating the new event does not change the old one's type) */
  new_event = create_event(target_pop, new_age);
  if (new_event < 0) { gph_fail(13); return 0.0; }
  GPH_SLOG(2, event_id, source_pop, target_pop, original_age, new_age, new_event);   /* patch.c:1451-1454 */
  setDI(inst, DI_ORIG, event_id);
  setDI(inst, DI_UPD, new_event);
  setDI(inst, DI_SRCPOP, source_pop);
  setDI(inst, DI_TGTPOP, target_pop);
  if (new_age > original_age) {
    dlin = (ev_type == GPH_OUT_MIG) ? (-1) : (1);
    bottom_event = ENEXT(event_id);
    top_event = new_event;
    bottom_pop = source_pop;
    top_age = new_age;
    bottom_age = original_age;
  } else {
    dlin = (ev_type == GPH_OUT_MIG) ? (1) : (-1);
    bottom_event = ENEXT(new_event);
    top_event = event_id;
    bottom_pop = target_pop;
    top_age = original_age;
    bottom_age = new_age;
  }
  setDI(inst, DI_DLIN, dlin);
  coal_stats_delta(inst, bottom_event, bottom_pop, top_event, dlin);
  mig_stats_delta(inst, bottom_age, bottom_pop, top_age, dlin);
  r = delta_lnld(inst);
  if (ev_type == GPH_COAL && source_pop != target_pop)
    r += gph_log_u(g_model.theta[source_pop] / g_model.theta[target_pop]);
  return r;
}
GPH_DEV void delta_clear(int inst)
{
  setDI(inst, DI_NPOPS, 0);
  setDI(inst, DI_NBANDS, 0);
  setDI(inst, DI_NEV, 0);
  setDI(inst, DI_DLIN, 0);
  setDI(inst, DI_ORIG, -1);
  setDI(inst, DI_UPD, -1);
}
// acceptEventChainChanges, patch.c:1540-1633
GPH_DEV void accept_event_chain_changes(int inst)
{
  int i, pop, b, ue, oe, dlin = DI(inst, DI_DLIN);
  /* the listed populations / bands / events are distinct: one lane per list entry */
  (void)pop; (void)b;
  GPH_EACH1(k, DI(inst, DI_NPOPS)) { const int q = gph_lds.s_dpops[inst][k]; gph_lds.coal[q] = gph_lds.coal[q] + gph_lds.s_dcoal[inst][k]; }
  GPH_EACH1(k, DI(inst, DI_NBANDS)) { const int q = gph_lds.s_dbands[inst][k]; gph_lds.migst[q] = gph_lds.migst[q] + gph_lds.s_dmig[inst][k]; }
  oe = DI(inst, DI_ORIG);
  i = DI(inst, DI_NEV) - 1;
  if (i >= 0 && DEV(inst, i) == oe) i--;
  GPH_EACH(k, i + 1) { const int q = gph_lds.s_dev[inst][k]; gph_lds.ev[q].nlin = (uint8_t)(gph_lds.ev[q].nlin + dlin); }
  ue = DI(inst, DI_UPD);
  if (ue >= 0) {
    const GphEvS O_ = ld_ev(oe);       /* node and type of the original event: one LDS round trip */
    setENODE(ue, O_.node);
    setETYPE(ue, O_.type);
    switch (O_.type) {
    case GPH_COAL: setNEV(O_.node, ue); break;
    case GPH_OUT_MIG: setMG(O_.node, MG_SEV, ue); break;
    case GPH_IN_MIG: setMG(O_.node, MG_TEV, ue); break;
    default: gph_fail(14); break;
    }
    remove_event(oe, DI(inst, DI_SRCPOP));
  }
  delta_clear(inst);
}
// rejectEventChainChanges, patch.c:1639-1661
GPH_DEV void reject_event_chain_changes(int inst)
{
  if (DI(inst, DI_UPD) >= 0) remove_event(DI(inst, DI_UPD), DI(inst, DI_TGTPOP));
  delta_clear(inst);
}

// ---------------------------------------------------------------- rubber band
// rubberBand, patch.c:596-801.  age0 = the age the population's chain starts at (comb->age of the reference): the
// model's popAge[pop] when the proposal is evaluated; the commit passes the age the evaluation saw (GphTauFin), because
// it may run after the model has moved on
GPH_DEVHOT double rubber_band(int pop, double age0, double static_point, double moving_point, double factor, int post,
                             int *out_num_events)
{
  int i, ev, b, node_id, num_lins, count_events = 0, flag, ty;
  LiveList live = {0, 0};
  pop = RFL(pop); post = RFL(post);
  double age, dt, mig_rate = 0.0, mig_delta, coal_delta = 0.0, lnLd = 0.0, age1;
  double fm1 = factor - 1.0;
  double start_time = gmin2(static_point, moving_point);
  double end_time = gmax2(static_point, moving_point);
  if (pop == g_lay.rootPop) { start_time = moving_point; end_time = GPH_OLDAGE; }
  ev = FIRSTEV(pop);
  age = age0;
  flag = (age >= start_time);
  int guard = 0;
  while (UNI(age < end_time)) {
    if (ev == -1) { gph_fail(11); break; }
    if (++guard > GPH_CAP_E) { gph_fail(95); break; }
    const GphEvS R = ld_ev(ev);      /* every field of the interval with ONE LDS round trip (they were five) */
    dt = gmin2(R.time, end_time - age);
    age += dt;
    if (!flag && UNI(age > start_time)) { flag = 1; dt = age - start_time; }
    if (flag) {
      dt *= fm1;
      num_lins = R.nlin;
      mig_delta = dt * num_lins;
      coal_delta += mig_delta * (num_lins - 1);
      lnLd -= mig_delta * mig_rate;
      if (post) {
        setEVT(ev, R.time + dt);
        for (b = 0; b < live.n; b++) setMIGST(ll_get(live, b), MIGST(ll_get(live, b)) + mig_delta);
      }
    }
    ty = R.type;
    if (UNI(age >= end_time) && ty != GPH_SAMPLES_START) break;
    node_id = R.node;
    switch (ty) {
    case GPH_COAL:
      if (flag) {
        count_events++;
        if (!post) {
          age1 = AGE(node_id);
          age1 += (age1 - static_point) * fm1;
          lik_adjust_age(node_id, age1);
        }
      }
      break;
    case GPH_SAMPLES_START:
      if (flag && g_model.sampleAge[pop] > 0) {
        if (static_point < moving_point && !post) {
          age1 = g_model.sampleAge[pop];
          age1 += (age1 - static_point) * fm1;
          for (i = 0; i < g_lay.n; i++)
            if (NPOP(i) == pop) lik_adjust_age(i, age1);
        }
      }
      break;
    case GPH_IN_MIG:
      if (flag && post) setMAGE(node_id, MAGE(node_id) + (MAGE(node_id) - static_point) * fm1);
      break;
    case GPH_MIG_BAND_START:
      mig_rate += g_model.migRate[node_id];
      ll_push(live, node_id);
      break;
    case GPH_MIG_BAND_END:
      mig_rate -= g_model.migRate[node_id];
      i = ll_find(live, node_id);
      if (i == live.n) { gph_fail(4); return 0.0; }
      ll_swap_remove(live, i);
      break;
    case GPH_END_CHAIN: age = end_time; break;
    default: break;
    }
    ev = R.next;
  }
  if (post) setCOALS(pop, COALS(pop) + coal_delta);
  lnLd -= coal_delta / (g_model.theta[pop]);
  *out_num_events += count_events;
  return lnLd;
}

// rubberBandRipple, patch.c:815-869
GPH_DEV double rubber_band_ripple(int do_or_redo)
{
  int i, pop, nw, orig, nmoved = ISC(IS_RB_NUM);
  gph_popmask affected = 0;
  double delta = 0.0;
  if (nmoved == 0) return 0.0;
  for (i = 0; i < nmoved; i++) {
    pop = RBI(2, i);
    orig = RBI(0, i);
    affected |= (gph_popmask)1 << pop;
    if (do_or_redo) {
      nw = create_event(pop, RBAGE(i));
      setRBI(1, i, nw);
      if (nw < 0) { gph_fail(5); return 0.0; }
      setETYPE(nw, ETYPE(orig));
      setENODE(nw, ENODE(orig));
      setETYPE(orig, GPH_DUMMY);
    } else {
      nw = RBI(1, i);
      if (nw < 0) continue;      /* (the "do" pass failed on this entry -- Fatal Error 0005, a chain already corrupted: nothing to undo) */
      setETYPE(orig, ETYPE(nw));
      if (FIRSTEV(pop) == nw) setENLIN(ENEXT(nw), ENLIN(nw));
      remove_event(nw, pop);
    }
  }
  for (pop = 0; pop < g_lay.K; pop++)
    if ((affected >> pop) & 1) delta += recalc_stats(pop);
  if (!do_or_redo) setISC(IS_RB_NUM, 0);
  return delta;
}

// traceLineage's consistency check when a walk enters the parent population (patch.c:1053: fabs(age / popAge - 1) >
// 0.01 is fatal): |age - popAge| > 0.01 popAge -- the same test without the fp64 division (14 instructions, once per
// population a walk crosses); a valid chain is off by rounding errors, nowhere near the threshold where the two forms differ
// (pop_age is the age of the ANCESTRAL population the walk enters: a split time, positive in every model the front end and
// gph_engine_set_model accept -- tau-initial 0 is refused upstream too; for pop_age == 0 the reference's quotient is inf or
// NaN and never "> 0.01", this form flags any age != 0: both are outside what a run can reach)
GPH_DEV bool pop_age_off(double age, double pop_age) { return UNI(fabs(age - pop_age) > 0.01 * pop_age); }

// the migration bands a lineage in population `pop` at time `age` is exposed to (patch.c:934-944, 1285-1294: a scan of all
// bands for target == pop and start < age < end, strict at the start of a walk, start <= age after a migration event),
// in increasing band order: only the bands whose target IS `pop` are looked at (GphModel.bandsInto) -- most populations
// have none, and the scan of every band cost a scalar load and a dozen instructions per band at the start of every walk
GPH_DEV void live_bands_into(int pop, double age, bool strict, LiveList &live, double &mig_rate)
{
  for (int w = 0; w < GPH_BANDW; w++) {
    for (uint32_t m = g_model.bandsInto[pop][w]; m != 0; m &= m - 1) {
      const int b = 32 * w + __builtin_ctz(m);
      const double st = g_model.bandStart[b];
      if ((strict ? st < age : st <= age) && g_model.bandEnd[b] > age) {
        mig_rate += g_model.migRate[b];
        ll_push(live, b);
      }
    }
  }
}

// ---------------------------------------------------------------- traceLineage
// traceLineage, patch.c:886-1331.  RECONNECT == 0: walk the existing edge above
// `node`, removing one lineage; RECONNECT == 1: re-sample its path from the prior
// what the fused pruning walk (trace_pair) hands to the prior-sampling walk when the two part: where the sampling walk stands
// and its running values -- on their common prefix the two walks add the same increments to the same start values
struct GphWalkResume { int ev, pop, nev, have_u; LiveList live; double age, mig_rate, dcoal, lnld, u; };
template <int RECONNECT, class RNG, bool RESUME = false>
GPH_DEVHOT int trace_lineage(int node, RNG &rng, const GphWalkResume *rs = nullptr)
{
  const int inst = RECONNECT;
  node = RFL(node);
  int i, pop, ev, node_id, b = -1, mig_source, proceed;
  LiveList live = {0, 0};
  int target, num_targets, nev = 0;
  double age, t = 0, event_sample, rate = 0.0, mig_rate, theta, lnld = 0.0, thinv, dcoal;
  int have_u = 0;
  double u0 = 0.0;
  (void)have_u; (void)u0;
  if constexpr (RESUME) {
    /* the walk is under way: trace_pair has initialised both delta instances and walked the common prefix */
    pop = rs->pop; ev = rs->ev; nev = rs->nev; live = rs->live; age = rs->age; mig_rate = rs->mig_rate; lnld = rs->lnld; dcoal = rs->dcoal;
    have_u = rs->have_u; u0 = rs->u;
    theta = g_model.theta[pop];
    thinv = g_model.thetaInv[pop];
    mig_source = -1;
    proceed = 1;
  } else {
  pop = NPOP(node);
  if (node < g_lay.n) {
    ev = FIRSTEV(pop);
    GphEvS S0 = ld_ev(ev);
    /* (bounded like every chain walk: a broken link or a cycle must not hang the wavefront -- Fatal Error 0101 as for a chain
     * without its SAMPLES_START; found by tests/test_native_comm.py: a host-build rank spun here on a chain broken for the test) */
    for (int guard_ = 0; S0.type != GPH_SAMPLES_START && S0.type != GPH_END_CHAIN && S0.next >= 0 && guard_ < GPH_CAP_E; guard_++) { ev = S0.next; S0 = ld_ev(ev); }
    if (S0.type != GPH_SAMPLES_START) { gph_fail(101); setDI(inst, DI_NEV, 0); setSPRLN(RECONNECT, 0.0); return RECONNECT ? -1 : 0; }
    ev = S0.next;
  } else {
    ev = ENEXT(NEV(node));
  }
  theta = g_model.theta[pop];
  thinv = g_model.thetaInv[pop];
  age = AGE(node);
  if (!RECONNECT) {
    setSPRI(SI_NOLD, 0);
    if (node != ISC(IS_ROOT)) setSPRI(SI_FEV_OLD, NEV(FATH(node)));
  } else {
    setSPRI(SI_NNEW, 0);
  }
  setDI(inst, DI_NPOPS, g_lay.K);
  setDI(inst, DI_NBANDS, g_lay.B);
  /* one lane per population / band */
  GPH_EACH1(k, g_lay.K) { gph_lds.s_dpops[inst][k] = (int16_t)k; gph_lds.s_dcoal[inst][k] = 0.0; }
  GPH_EACH1(k, g_lay.B) { gph_lds.s_dbands[inst][k] = (int16_t)k; gph_lds.s_dmig[inst][k] = 0.0; }
  mig_rate = 0.0;
  live_bands_into(pop, age, true, live, mig_rate);
  mig_source = -1;
  proceed = 1;
  /* per step every field of the current interval is read ONCE into registers (the LDS image is only
   * re-read when the walk steps to another event); the coalescence-statistic delta of the population
   * being crossed accumulates in a register and is written back when the walk leaves it -- same values,
   * same order of additions */
  dcoal = DCOAL(inst, pop);
  }
  const int fev_old = RECONNECT ? -1 : SPRI(SI_FEV_OLD);
  if constexpr (RECONNECT != 0) {
    /* The prior-sampling walk as two nested loops: a TIGHT inner loop over the intervals the lineage passes through
     * (three of four), with one exit for "an event falls inside this interval", and the rare work -- creating the
     * migration events, or the coalescence that ends the walk -- outside it.  Same operations in the same order as
     * the single loop below (which the pruning walk still uses); the point is what the register allocator makes of
     * it: the single loop carried a dozen scalars across six merging paths and paid for the merges with copies
     * (a quarter of its instructions). */
    int status = 0;          /* 1 coalesced, -1 end of the root chain (no event sampled), 2 no migration slot left */
    for (;;) {
      double et = 0.0;
      int nlin = 0, evnext = -1;
      bool through;
      /* ONE exit, at the bottom: the rare stops raise `status` and fall through to it.  (Measured alternatives: the
       * rare arrivals handled after the step instead of before it, or outside this loop altogether -- the second
       * leaves a copy-free inner loop and pays for it at every exit and re-entry: +2 % sweep time.  Requesting the
       * NEXT event's record while the current interval is worked on (software pipelining of the one LDS round trip
       * per interval): +0.9 % -- the other wavefronts already cover that latency, the four extra loop-carried
       * registers do not come free.) */
      do {
        through = false;
        if (nev >= GPH_CAP_E) { gph_fail(96); status = 3; }
        else if (ev < 0) {
          if (g_model.popFather[pop] < 0) status = -1;
          else {
            setDCOAL(inst, pop, dcoal);
            pop = g_model.popFather[pop];
            dcoal = DCOAL(inst, pop);
            theta = g_model.theta[pop];
            thinv = g_model.thetaInv[pop];
            ev = FIRSTEV(pop);
            mig_rate = 0.0;
            if (pop_age_off(age, g_model.popAge[pop])) { gph_fail(8); status = 3; }
            else age = g_model.popAge[pop];
          }
        }
        if (status == 0) {
          const GphEvS R = ld_ev(ev);
          nlin = R.nlin;
          et = R.time;
          evnext = R.next;
          rate = mig_rate + gph_div_by(2 * nlin, theta, thinv);
          if (UNI(rate <= 0)) {
            through = true;
          } else {
            double u;
            if constexpr (RESUME) { if (have_u) { u = u0; have_u = 0; } else u = l_rndu(rng); }   /* (the draw trace_pair's test of this interval consumed) */
            else u = l_rndu(rng);
            /* t = -(1/rate) log(u) is only USED when it falls inside the interval.  -log(u) >= y + y^2/2 for
             * y = 1 - u in (0, 1]: when that bound clears rate*et with a margin far above the rounding errors of
             * either side (each a few 1e-16 relative), t >= et is certain and neither the logarithm nor the
             * reciprocal is evaluated -- about three of four draws of a walk pass through their interval */
            const double y = 1.0 - u;
            through = UNI(y + 0.5 * y * y >= (rate * et) * (1.0 + 1e-9));
            if (!through) {
              t = -(1 / rate) * gph_log_u(u);
              through = UNI(t >= et);
            }
          }
          if (through) {
            t = et;
            age += t;
            dcoal += 2 * nlin * t;
            for (i = 0; i < live.n; i++) setDMIG(inst, ll_get(live, i), DMIG(inst, ll_get(live, i)) + t);
            setDEV(inst, nev, ev);
            nev++;
            lnld -= rate * t;
            if (R.type == GPH_MIG_BAND_START) {
              mig_rate += g_model.migRate[R.node];
              ll_push(live, R.node);
            } else if (R.type == GPH_MIG_BAND_END) {
              mig_rate -= g_model.migRate[R.node];
              if (live.n == 1) mig_rate = 0.0;
              i = ll_find(live, R.node);
              if (i < live.n) ll_swap_remove(live, i);
            }
            ev = evnext;
          }
        }
      } while (through);
      if (status) break;
      /* an event at age + t, inside the interval of `ev` */
      age += t;
      event_sample = rate * l_rndu(rng);
      if (UNI(event_sample < mig_rate)) {
        int k = SPRI(SI_NNEW);
        if (GPH_MAX_MIGS <= ISC(IS_NUM_MIGS) + k - SPRI(SI_NOLD)) {
          setCNT(CN_NOTENOUGH, CNT(CN_NOTENOUGH) + 1);
          status = 2;
          break;
        }
        for (i = 0; event_sample >= 0 && i < live.n; i++) event_sample -= g_model.migRate[ll_get(live, i)];
        if (i <= 0) { gph_fail(9); status = 3; break; }
        b = ll_get(live, i - 1);
        setSPRA(SA_NEWBAND, k, b);
        if (g_model.bandTgt[b] != pop) { gph_fail(9); status = 3; break; }
        setSPRAGE(k, age);
        const int nw = create_event_before(pop, ev, t);   /* the new interval: same lineage count, type DUMMY */
        setSPRA(SA_NEWIN, k, nw);
        mig_source = create_event(g_model.bandSrc[b], age);
        setSPRA(SA_NEWOUT, k, mig_source);
        if (mig_source < 0) { gph_fail(10); status = 3; break; }
        setSPRI(SI_NNEW, k + 1);
        dcoal += 2 * nlin * t;
        for (i = 0; i < live.n; i++) setDMIG(inst, ll_get(live, i), DMIG(inst, ll_get(live, i)) + t);
        setDEV(inst, nev, nw);
        nev++;
        lnld -= rate * t;
        lnld += g_model.logMigRate[b];
        setDCOAL(inst, pop, dcoal);
        pop = g_model.bandSrc[b];
        dcoal = DCOAL(inst, pop);
        theta = g_model.theta[pop];
        thinv = g_model.thetaInv[pop];
        mig_rate = 0.0;
        live.n = 0;
        live_bands_into(pop, age, false, live, mig_rate);
        ev = ENEXT(mig_source);
        mig_source = -1;
      } else {
        num_targets = edges_for_time_pop((age - t) + et / 2, pop, node);
        if (num_targets != nlin) { gph_fail(11); status = 3; break; }
        i = (int)((event_sample - mig_rate) * theta / 2);
        target = gi16(&GphLds::s_targets, i);
        lik_spr(node, target, age);
        setSPRI(SI_FPOP_NEW, pop);
        setSPRI(SI_TARGET, target);
        const int nw = create_event_before(pop, ev, t);
        setSPRI(SI_FEV_NEW, nw);
        dcoal += 2 * nlin * t;
        for (i = 0; i < live.n; i++) setDMIG(inst, ll_get(live, i), DMIG(inst, ll_get(live, i)) + t);
        setDEV(inst, nev, nw);
        nev++;
        lnld -= rate * t;
        status = 1;
        break;
      }
    }
    if (status == -1 || status == 2) { setDCOAL(inst, pop, dcoal); setDI(inst, DI_NEV, nev); setSPRLN(RECONNECT, lnld); return -1; }
    setDCOAL(inst, pop, dcoal);
    lnld += g_model.logTwoTheta[pop];
    setDI(inst, DI_NEV, nev);
    setSPRLN(RECONNECT, lnld);
    return 0;
  }
  /* RECONNECT == 0: the pruning walk along the existing edge (no sampling, no new events) */
  while (proceed) {
    if (nev >= GPH_CAP_E) { gph_fail(96); break; }   /* every step lists one event: a corrupted chain cannot spin */
    if (ev < 0) {
      if (g_model.popFather[pop] < 0) { gph_fail(6); break; }
      setDCOAL(inst, pop, dcoal);
      pop = g_model.popFather[pop];
      dcoal = DCOAL(inst, pop);
      theta = g_model.theta[pop];
      thinv = g_model.thetaInv[pop];
      ev = FIRSTEV(pop);
      mig_rate = 0.0;
      if (pop_age_off(age, g_model.popAge[pop])) { gph_fail(8); break; }
      age = g_model.popAge[pop];
    }
    const GphEvS R = ld_ev(ev);
    node_id = R.node;
    const int nlin = R.nlin - 1;
    setENLIN(ev, nlin);
    t = R.time;
    age += t;
    proceed = (ev != fev_old);
    if (R.type == GPH_IN_MIG) {
      if (MG(node_id, MG_BRANCH) == node) {
        int k = SPRI(SI_NOLD);
        b = MG(node_id, MG_BAND);
        mig_source = MG(node_id, MG_SEV);
        setSPRA(SA_OLD, k, node_id);
        setSPRI(SI_NOLD, k + 1);
      }
    }
    dcoal += 2 * nlin * t;
    for (i = 0; i < live.n; i++) setDMIG(inst, ll_get(live, i), DMIG(inst, ll_get(live, i)) + t);
    setDEV(inst, nev, ev);
    nev++;
    lnld -= (mig_rate + gph_div_by(2 * nlin, theta, thinv)) * t;
    if (mig_source >= 0) {
      lnld += g_model.logMigRate[b];
      setDCOAL(inst, pop, dcoal);
      pop = g_model.bandSrc[b];
      dcoal = DCOAL(inst, pop);
      theta = g_model.theta[pop];
      thinv = g_model.thetaInv[pop];
      mig_rate = 0.0;
      live.n = 0;
      live_bands_into(pop, age, false, live, mig_rate);
      ev = ENEXT(mig_source);
      mig_source = -1;
    } else {
      if (R.type == GPH_MIG_BAND_START) {
        mig_rate += g_model.migRate[node_id];
        ll_push(live, node_id);
      } else if (R.type == GPH_MIG_BAND_END) {
        mig_rate -= g_model.migRate[node_id];
        if (live.n == 1) mig_rate = 0.0;
        i = ll_find(live, node_id);
        if (i < live.n) ll_swap_remove(live, i);
      }
      ev = R.next;
    }
  }
  setDCOAL(inst, pop, dcoal);
  lnld += g_model.logTwoTheta[pop];
  setDI(inst, DI_NEV, nev);
  setSPRLN(RECONNECT, lnld);
  return 0;
}

// The two lineage walks of an SPR proposal (GPhoCS.c:2659-2670: traceLineage(gen, node, 0), then traceLineage(gen, node, 1))
// with their COMMON PREFIX walked once.  Both start at the node's event with the same age, population and live bands; the
// pruning walk follows the existing edge and takes one lineage off every interval, the prior-sampling walk then passes
// through the same intervals -- with the decremented lineage counts -- until an event falls inside one.  As long as both are
// on the same interval they add the SAME increments (2 n t, (m + 2 n / theta) t, t per live band) to the same start values, so
// the pruning walk carries the other one along: per interval it runs the sampling walk's test (the draw and the pass-through
// bound, in the sampling walk's order -- the pruning walk draws nothing) and writes its increments to both delta instances.
// Where they part -- an event sampled inside the interval (the test's draw is handed over), the pruning walk's own migration
// event, or the end of the old edge -- the sampling walk's state is parked (LDS, over the candidate-edge list that is not in
// use yet) and trace_lineage<1> resumes from it once the pruning walk has finished: every interval from there on it reads as
// it always did, after the pruning walk's decrements.  Measured on the benchmark's shape: 2.5 of 4.7 intervals are common.
#if !GPH_BIG_BANDS
#define GPH_WRS_OFF ((8 - offsetof(GphLds, s_targets) % 8) % 8)       /* to the next 8-byte boundary inside s_targets */
#define GPH_WRS_BASE ((lchar *)gph_lds.s_targets + GPH_WRS_OFF)
static_assert(sizeof(((GphLds *)0)->s_targets) >= 56 + GPH_WRS_OFF, "the parked walk state (56 bytes) overlays s_targets");
GPH_DEV void walk_park(const GphWalkResume &r)
{
  lf64 *d = (lf64 *)GPH_WRS_BASE;
  d[0] = r.age; d[1] = r.mig_rate; d[2] = r.dcoal; d[3] = r.lnld; d[4] = r.u;
  *(GPH_LDS uint64_t *)(GPH_WRS_BASE + 40) = r.live.bits;
  li16 *h = (li16 *)(GPH_WRS_BASE + 48);
  h[0] = (int16_t)r.ev; h[1] = (int16_t)r.pop; h[2] = (int16_t)r.nev; h[3] = (int16_t)(r.live.n | (r.have_u << 8));
}
GPH_DEV void walk_unpark(GphWalkResume &r)
{
  lf64 *d = (lf64 *)GPH_WRS_BASE;
  r.age = d[0]; r.mig_rate = d[1]; r.dcoal = d[2]; r.lnld = d[3]; r.u = d[4];
  const uint64_t bits = *(GPH_LDS uint64_t *)(GPH_WRS_BASE + 40);
  r.live.bits = (uint64_t)(uint32_t)RFL((int)(uint32_t)bits) | ((uint64_t)(uint32_t)RFL((int)(uint32_t)(bits >> 32)) << 32);
  li16 *h = (li16 *)(GPH_WRS_BASE + 48);
  r.ev = RFL((int)h[0]); r.pop = RFL((int)h[1]); r.nev = RFL((int)h[2]);
  const int w = RFL((int)h[3]);
  r.live.n = w & 255; r.have_u = (w >> 8) & 1;
}
template <class RNG> GPH_DEVHOT int trace_pair(int node, RNG &rng)
{
  node = RFL(node);
  int i, pop, ev, node_id, b = -1, mig_source = -1, proceed = 1, nev = 0;
  LiveList live = {0, 0};
  double age, t, mig_rate, theta, thinv, lnld = 0.0, dcoal;
  pop = NPOP(node);
  if (node < g_lay.n) {
    ev = FIRSTEV(pop);
    GphEvS S0 = ld_ev(ev);
    for (int guard_ = 0; S0.type != GPH_SAMPLES_START && S0.type != GPH_END_CHAIN && S0.next >= 0 && guard_ < GPH_CAP_E; guard_++) { ev = S0.next; S0 = ld_ev(ev); }
    if (S0.type != GPH_SAMPLES_START) { gph_fail(101); setDI(0, DI_NEV, 0); setSPRLN(0, 0.0); setDI(1, DI_NEV, 0); setSPRLN(1, 0.0); return -1; }
    ev = S0.next;
  } else {
    ev = ENEXT(NEV(node));
  }
  theta = g_model.theta[pop];
  thinv = g_model.thetaInv[pop];
  age = AGE(node);
  setSPRI(SI_NOLD, 0);
  if (node != ISC(IS_ROOT)) setSPRI(SI_FEV_OLD, NEV(FATH(node)));
  setSPRI(SI_NNEW, 0);
  setDI(0, DI_NPOPS, g_lay.K); setDI(0, DI_NBANDS, g_lay.B);
  setDI(1, DI_NPOPS, g_lay.K); setDI(1, DI_NBANDS, g_lay.B);
  GPH_EACH1(k, g_lay.K) { gph_lds.s_dpops[0][k] = (int16_t)k; gph_lds.s_dcoal[0][k] = 0.0; gph_lds.s_dpops[1][k] = (int16_t)k; gph_lds.s_dcoal[1][k] = 0.0; }
  GPH_EACH1(k, g_lay.B) { gph_lds.s_dbands[0][k] = (int16_t)k; gph_lds.s_dmig[0][k] = 0.0; gph_lds.s_dbands[1][k] = (int16_t)k; gph_lds.s_dmig[1][k] = 0.0; }
  mig_rate = 0.0;
  live_bands_into(pop, age, true, live, mig_rate);
  const int fev_old = SPRI(SI_FEV_OLD);
  dcoal = DCOAL(0, pop);
  int both = 1;                 /* the prior-sampling walk is still on this walk's interval */
  GphWalkResume rs;
  /* ONE exit.  The three consistency checks of a step (a chain that ends in the root population, a population entered at
   * the wrong age, no slot left in the event list for the next step) raise `bad`, which the loop condition tests together
   * with `proceed`; the step that raised it finishes on what it has (LDS reads and writes only -- an out-of-range LDS
   * access is dropped by the hardware) and the failure is reported behind the loop.  As `break`s the checks were three more
   * exits whose live-out values met the normal exit's behind the loop: two dozen register copies at the head of EVERY
   * interval; as `return`s from inside the loop they still cost exit flags cleared and tested in every interval
   * (tools/bbcount.sh, round 4). */
  int bad = 0;                  /* the code of the failed check */
  int at_end = 0;               /* the step just done was the old edge's last */
#define GPH_WALK_ABORT() do { setDI(0, DI_NEV, 0); setSPRLN(0, 0.0); setDI(1, DI_NEV, 0); setSPRLN(1, 0.0); return -1; } while (0)
  while (proceed && !bad) {
    if (ev < 0 && g_model.popFather[pop] < 0) bad = 6;
    else if (ev < 0) {
      setDCOAL(0, pop, dcoal);
      if (both) setDCOAL(1, pop, dcoal);
      pop = g_model.popFather[pop];
      dcoal = DCOAL(0, pop);
      theta = g_model.theta[pop];
      thinv = g_model.thetaInv[pop];
      ev = FIRSTEV(pop);
      mig_rate = 0.0;
      if (pop_age_off(age, g_model.popAge[pop])) bad = 8;
      age = g_model.popAge[pop];
    }
    const GphEvS R = ld_ev(ev);
    node_id = R.node;
    const int nlin = R.nlin - 1;
    /* m + 2 (n - 1) / theta of this interval: the sampling walk's event rate and the factor of this walk's own likelihood
     * term below (one evaluation for both: the compiler does not merge the two across the branch) */
    const double rate = mig_rate + gph_div_by(2 * nlin, theta, thinv);
    if (both) {
      /* the sampling walk's step for this interval (trace_lineage<1>'s inner loop), on the lineage count it would read */
      if (!UNI(rate <= 0)) {
        const double u = l_rndu(rng);
        const double y = 1.0 - u;
        bool through = UNI(y + 0.5 * y * y >= (rate * R.time) * (1.0 + 1e-9));
        if (!through) through = UNI(-(1 / rate) * gph_log_u(u) >= R.time);
        if (!through) {
          /* an event falls inside this interval: the sampling walk takes over HERE, with the draw */
          rs.ev = ev; rs.pop = pop; rs.nev = nev; rs.have_u = 1; rs.live = live; rs.age = age; rs.mig_rate = mig_rate; rs.dcoal = dcoal; rs.lnld = lnld; rs.u = u;
          walk_park(rs);
          both = 0;
        }
      }
    }
    setENLIN(ev, nlin);
    t = R.time;
    age += t;
    at_end = (ev == fev_old);
    proceed = !at_end;
    if (R.type == GPH_IN_MIG) {
      if (MG(node_id, MG_BRANCH) == node) {
        int k = SPRI(SI_NOLD);
        b = MG(node_id, MG_BAND);
        mig_source = MG(node_id, MG_SEV);
        setSPRA(SA_OLD, k, node_id);
        setSPRI(SI_NOLD, k + 1);
      }
    }
    dcoal += 2 * nlin * t;
    for (i = 0; i < live.n; i++) {
      const double v = DMIG(0, ll_get(live, i)) + t;
      setDMIG(0, ll_get(live, i), v);
      if (both) setDMIG(1, ll_get(live, i), v);
    }
    /* the event goes on both walks' lists: unconditionally -- once the walks have parted, the sampling walk (which resumes
     * only after this loop) lists its own events from rs.nev on, over whatever this walk left there */
    setDEV(0, nev, ev);
    setDEV(1, nev, ev);
    nev++;
    if (nev >= GPH_CAP_E) proceed = 0;             /* the next step would not have a slot in the event list: see behind the loop */
    lnld -= rate * t;
    if (mig_source >= 0) {
      if (both) {
        /* the old edge leaves through a migration event: the sampling walk passes it and stays in this population */
        rs.ev = R.next; rs.pop = pop; rs.nev = nev; rs.have_u = 0; rs.live = live; rs.age = age; rs.mig_rate = mig_rate; rs.dcoal = dcoal; rs.lnld = lnld; rs.u = 0.0;
        walk_park(rs);
        both = 0;
      }
      lnld += g_model.logMigRate[b];
      setDCOAL(0, pop, dcoal);
      pop = g_model.bandSrc[b];
      dcoal = DCOAL(0, pop);
      theta = g_model.theta[pop];
      thinv = g_model.thetaInv[pop];
      mig_rate = 0.0;
      live.n = 0;
      live_bands_into(pop, age, false, live, mig_rate);
      ev = ENEXT(mig_source);
      mig_source = -1;
    } else {
      if (R.type == GPH_MIG_BAND_START) {
        mig_rate += g_model.migRate[node_id];
        ll_push(live, node_id);
      } else if (R.type == GPH_MIG_BAND_END) {
        mig_rate -= g_model.migRate[node_id];
        if (live.n == 1) mig_rate = 0.0;
        i = ll_find(live, node_id);
        if (i < live.n) ll_swap_remove(live, i);
      }
      ev = R.next;
    }
  }
  if (!bad && !at_end) bad = 96;        /* stopped by the capacity of the event list, not by the end of the old edge */
  if (bad) { gph_fail(bad); GPH_WALK_ABORT(); }
  setDCOAL(0, pop, dcoal);
  setDI(0, DI_NEV, nev);
  setSPRLN(0, lnld + g_model.logTwoTheta[pop]);
  if (gph_failed()) { setDI(1, DI_NEV, 0); setSPRLN(1, 0.0); return -1; }
  if (both) {
    /* the old edge ended first: the sampling walk goes on from the next event with this walk's values */
    rs.ev = ev; rs.pop = pop; rs.nev = nev; rs.have_u = 0; rs.live = live; rs.age = age; rs.mig_rate = mig_rate; rs.dcoal = dcoal; rs.lnld = lnld; rs.u = 0.0;
  } else {
    walk_unpark(rs);
  }
  return trace_lineage<1, RNG, true>(node, rng, &rs);
#undef GPH_WALK_ABORT
}
#endif

// replaceMigNodes, patch.c:1343-1420
GPH_DEV void replace_mig_nodes(int node)
{
  int i, j, mig = 0, b, nold = SPRI(SI_NOLD), nnew = SPRI(SI_NNEW), nm;
  int mx = nold > nnew ? nold : nnew;
  for (i = 0; i < mx; i++) {
    if (i < nold) {
      mig = SPRA(SA_OLD, i);
      remove_event(MG(mig, MG_SEV), MG(mig, MG_SPOP));
      remove_event(MG(mig, MG_TEV), MG(mig, MG_TPOP));
      b = MG(mig, MG_BAND);
      setNMIGB(b, NMIGB(b) - 1);
    } else {
      for (mig = 0; mig < GPH_MAX_MIGS; mig++) if (MG(mig, MG_BAND) < 0) break;
      if (mig == GPH_MAX_MIGS) { gph_fail(12); return; }
      nm = ISC(IS_NUM_MIGS);
      setLIVING(nm, mig);
      setISC(IS_NUM_MIGS, nm + 1);
      setMG(mig, MG_BRANCH, node);
    }
    if (i < nnew) {
      setMG(mig, MG_SEV, SPRA(SA_NEWOUT, i));
      setMG(mig, MG_TEV, SPRA(SA_NEWIN, i));
      setMAGE(mig, SPRAGE(i));
      b = SPRA(SA_NEWBAND, i);
      setMG(mig, MG_BAND, b);
      setMG(mig, MG_SPOP, g_model.bandSrc[b]);
      setMG(mig, MG_TPOP, g_model.bandTgt[b]);
      setETYPE(SPRA(SA_NEWOUT, i), GPH_OUT_MIG);
      setENODE(SPRA(SA_NEWOUT, i), mig);
      setETYPE(SPRA(SA_NEWIN, i), GPH_IN_MIG);
      setENODE(SPRA(SA_NEWIN, i), mig);
      setNMIGB(b, NMIGB(b) + 1);
    } else {
      setMG(mig, MG_BAND, -1);
      nm = ISC(IS_NUM_MIGS);
      for (j = 0; j < nm; j++) {
        if (LIVING(j) == mig) { setLIVING(j, LIVING(nm - 1)); setISC(IS_NUM_MIGS, nm - 1); break; }
      }
    }
  }
}

// synchronizeEvents, patch.c:3548-3633
GPH_DEV int synchronize_events()
{
  int i, pop, ev, id, res = 1;
  double realAge = 0.0, age, PREC = 0.0000001, et;
  for (i = 0; i < g_lay.K; i++) {
    pop = g_model.postOrder[i];
    int guard = 0;
    ev = FIRSTEV(pop);
    age = g_model.popAge[pop];
    int nxt_;
    for (; ev >= 0; ev = nxt_) {
      if (++guard > GPH_CAP_E) { gph_fail(97); return 0; }
      const GphEvS R = ld_ev(ev);    /* one LDS round trip per event (this pass walks every chain of every locus once per iteration) */
      nxt_ = R.next;
      id = R.node;
      age += R.time;
      switch (R.type) {
      case GPH_SAMPLES_START: realAge = g_model.sampleAge[pop]; break;
      case GPH_COAL: realAge = AGE(id); break;
      case GPH_IN_MIG:
      case GPH_OUT_MIG: realAge = MAGE(id); break;
      case GPH_MIG_BAND_START: realAge = g_model.bandStart[id]; break;
      case GPH_MIG_BAND_END: realAge = g_model.bandEnd[id]; break;
      case GPH_END_CHAIN:
        if (pop != g_lay.rootPop) realAge = g_model.popAge[g_model.popFather[pop]];
        else realAge = age;
        break;
      default: realAge = age; break;
      }
      if (fabs(realAge - age) > PREC) res = 0;
      et = R.time + (realAge - age);
      if (et < -PREC) res = 0;
      else if (et < 0.0) et = 0.0;
      setEVT(ev, et);
      age = realAge;
    }
  }
  return res;
}

// the state-mutating part of checkGtreeStructure (patch.c:2978-3380): statistics are
// recomputed from the chain and overwrite the stored ones; returns 0 on inconsistency
GPH_DEV int check_gtree_structure()
{
  int i, n, pop, b, ev, id, res = 1, nc;
  LiveList live = {0, 0};
  double age, dt, PREC = 0.0000000001, cs;
  /* lineages entering each population: LDS work list (s_targets is free here) */
  for (pop = 0; pop < g_lay.K; pop++) si16(&GphLds::s_targets, pop, 0);
  for (i = 0; i < g_lay.K; i++) {
    pop = g_model.postOrder[i];
    cs = 0.0;
    nc = 0;
    n = gi16(&GphLds::s_targets, pop);
    age = g_model.popAge[pop];
    live.n = 0;
    int guard = 0;
    for (ev = FIRSTEV(pop); ev >= 0; ev = ENEXT(ev)) {
      if (++guard > GPH_CAP_E) { gph_fail(98); return 0; }
      if (ENLIN(ev) != n) res = 0;
      if (ENEXT(ev) >= 0 && ev != EPREV(ENEXT(ev))) res = 0;
      id = ENODE(ev);
      dt = EVT(ev);
      age += dt;
      cs += n * (n - 1) * dt;
      for (b = 0; b < live.n; b++) sf64(&GphLds::s_chkmig, ll_get(live, b), gf64(&GphLds::s_chkmig, ll_get(live, b)) + n * dt);
      switch (ETYPE(ev)) {
      case GPH_SAMPLES_START:
        n += g_model.samplesPerPop[pop];
        if (fabs(g_model.sampleAge[pop] - age) > PREC) res = 0;
        break;
      case GPH_COAL:
        nc++;
        n--;
        if (fabs(AGE(id) - age) > PREC) res = 0;
        if (NPOP(id) != pop || NEV(id) != ev) res = 0;
        break;
      case GPH_IN_MIG: {
        int bb = MG(id, MG_BAND);
        si16(&GphLds::s_chknm, bb, gi16(&GphLds::s_chknm, bb) + 1);
        n--;
        if (fabs(MAGE(id) - age) > PREC || MG(id, MG_TEV) != ev) res = 0;
        break;
      }
      case GPH_OUT_MIG:
        n++;
        if (fabs(MAGE(id) - age) > PREC || MG(id, MG_SEV) != ev) res = 0;
        break;
      case GPH_MIG_BAND_START:
        ll_push(live, id);
        si16(&GphLds::s_chknm, id, 0);
        sf64(&GphLds::s_chkmig, id, 0.0);
        if (fabs(g_model.bandStart[id] - age) > PREC) res = 0;
        break;
      case GPH_MIG_BAND_END:
        b = ll_find(live, id);
        if (b == live.n) res = 0;
        else ll_swap_remove(live, b);
        if (fabs(g_model.bandEnd[id] - age) > PREC) res = 0;
        break;
      case GPH_END_CHAIN:
        if (id != pop || live.n != 0 || ENEXT(ev) >= 0) res = 0;
        if (pop != g_lay.rootPop) {
          si16(&GphLds::s_targets, g_model.popFather[pop], gi16(&GphLds::s_targets, g_model.popFather[pop]) + n);
          if (fabs(g_model.popAge[g_model.popFather[pop]] - age) > PREC) res = 0;
        }
        break;
      default: res = 0; break;
      }
    }
    sf64(&GphLds::s_chkcoal, pop, cs);
    si16(&GphLds::s_chknc, pop, nc);
  }
  for (pop = 0; pop < g_lay.K; pop++) {
    if (fabs(gf64(&GphLds::s_chkcoal, pop) - COALS(pop)) > PREC) res = 0;
    setCOALS(pop, gf64(&GphLds::s_chkcoal, pop));
    if (gi16(&GphLds::s_chknc, pop) != NCOAL(pop)) res = 0;
  }
  for (b = 0; b < g_lay.B; b++) {
    if (fabs(gf64(&GphLds::s_chkmig, b) - MIGST(b)) > PREC) res = 0;
    setMIGST(b, gf64(&GphLds::s_chkmig, b));
    if (gi16(&GphLds::s_chknm, b) != NMIGB(b)) res = 0;
  }
  return res;
}
