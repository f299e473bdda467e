// gph_kernels.h -- per-locus kernel bodies: stage the locus into LDS, run the
// proposal(s), stage it back.  One 64-lane wavefront (= one workgroup) per locus.
//
// Each kb_* function is the body of the per-locus loop of one reference
// proposal function (GPhoCS.c); the `#pragma omp parallel for` over loci there
// (MultiCoreUtils.h:8-21) is the grid here.  Cross-locus `omp atomic`
// accumulations become per-locus output slots reduced by a fixed-shape tree.
#pragma once
#include "gph_rt.h"

struct GphDev {            // device pointers (passed by value to every kernel)
  char *pages;             // L * page_bytes
  char *shadow;            // L * page_bytes
  char *cond;              // conditionals, per-locus byte offset cond_off[g]
  const uint64_t *cond_off;
  const char *seq;         // leaf codes, phases, counts; per-locus byte offset seq_off[g]
  const uint64_t *seq_off;
  const int32_t *P;        // per locus (slot g): P[2 g] phased patterns, P[2 g + 1] unphased patterns (those with a phase count: the U of the algorithmic-byte formula)
  const int32_t *orig;     // original (input-order) local index of the locus stored at slot g
  double *out;             // L * GPH_OUT_SLOTS
  double *stats;           // L * (2K+2B): coal_stats, num_coals, mig_stats, num_migs of every locus, compact
  int32_t L;               // loci on this device
  int32_t Ltot;            // loci over all devices (dataSetup.numLoci)
  int64_t locus_begin;     // global index of this device's first locus
  int32_t *err;            // sticky error code of the kernels that are not followed by a reduction (commit / revert)
  // decision-level transcript (GPH_LOGSTEPS builds only -- the host build of the tests and the control build
  // libgphocs_hip_plain.so; every other build carries the three words and never reads them): per-proposal records
  // of the selected loci, upstream's -DLOG_STEPS (GPhoCS.c:2363-2401, 2540-2577, 2654-2718; patch.c:1451)
  const int32_t *slog_map; // per slot: index of its record buffer, -1 = not logged
  double *slog;            // [selected][slog_cap][8]: kind, six values, spare
  int32_t *slog_n;         // [selected] records written (may exceed slog_cap: the excess is dropped)
  int32_t slog_cap, pad_;
};

#include "gph_locus.h"   // opens struct GphCtx; closed at the end of this file

// Loci are stored sorted by decreasing number of phased patterns (the longest wavefronts start first).  ONE
// dispatch covers every locus with at most one pattern per lane (P <= 64); the rare pattern-rich ones form a second
// one with its own LDS allocation (per-pattern terms array, generic mapping).  More, finer P-buckets were measured
// and dropped: every dispatch costs about one wavefront lifetime of tail.

#undef GPH_FILE_ID
#define GPH_FILE_ID 2
// ---------------------------------------------------------------- staging
GPH_DEV void copy16_g2l(int lds_off, const char *src, int bytes) { gph_copy16_in(GPH_SMB + lds_off, src, bytes >> 4); }
// HBM page <-> page part of the static LDS image: identical layout, one coalesced copy
GPH_DEV void page_in(const char *page) { gph_copy16_in(GPH_LDSP(&gph_lds), page, g_lay.page_bytes >> 4); }
GPH_DEV void page_out(char *page) { gph_copy16_out(page, GPH_LDSP(&gph_lds), g_lay.page_bytes >> 4); }

GPH_DEV void scratch_init(const GphDev &D, int g, int P, uint64_t cond_off)
{
  int k;
  for (k = 0; k < CN_COUNT; k++) setCNT(k, 0);
  setCNT(CN_P, P);
  setCNT(CN_QPH, GPH_Q_PHASES(P, g_lay.n)); setCNT(CN_QCNT, GPH_Q_COUNT(P, g_lay.n)); setCNT(CN_QTERMS, GPH_Q_TERMS(P, g_lay.n, g_lay.cnt16));
  setCNT(CN_SUMLDS, g_lay.lds_sum && GPH_Q_TERMS(P, g_lay.n, g_lay.cnt16) + 8 * ((P + 7) & ~7) <= g_lay.dyn_bytes);
  if (P > g_lay.huge_P) {       /* the block stays in HBM: its address for the generic paths (seq_ref) */
    const uint64_t a_ = (uint64_t)(uintptr_t)(D.seq + D.seq_off[g]);
    setCNT(CN_HUGE, 1); setCNT(CN_SEQLO, (int)(uint32_t)a_); setCNT(CN_SEQHI, (int)(uint32_t)(a_ >> 32));
  }
  sf64(&GphLds::s_cntf, 0, 0.0);
#if defined(GPH_STAMPS) || defined(GPH_HOSTEMU)
  for (k = 0; k < 8; k++) gph_lds.s_stamp[k] = 0.0;
#endif
  set_cond_base(D.cond + cond_off);
  delta_clear(0);
  delta_clear(1);
}

// load page (+ optionally the read-only sequence block: leaf codes, phases, counts).
// Conditionals are never staged: kernels read/write them in place (see cond_base()).
// All global loads of the stage-in are ISSUED before the first one is waited for: the generic copy loops
// (load 1 KB, wait, write LDS, repeat) serialised 5-7 memory round trips at the head of every wavefront, a quarter
// of the lifetime of a wavefront of the short kernels (tau / mixing evaluate, commit).
GPH_DEV void stage_in_copy(const GphDev &D, int g, const char *pages, int withSeq, int &P_, uint64_t &co_)
{
  constexpr int PCH = (int)((offsetof(GphLds, s_dcoal) + 1023) / 1024);   /* 1 KB chunks of the page part */
  constexpr int SCH = 2;                                                   /* first 2 KB of the sequence block */
  uint64_t o0 = 0, o1 = 0;
  if (withSeq) { o0 = D.seq_off[g]; o1 = D.seq_off[g + 1]; }
  P_ = D.P[2 * g];               /* per-locus table entries: scalar loads, in flight with everything else */
  co_ = D.cond_off[g];
  /* (a locus whose block outgrows the launch group's LDS reads it where it lies: nothing to stage) */
  gph_copy16_in2<PCH, SCH>(GPH_LDSP(&gph_lds), pages + (size_t)g * g_lay.page_bytes, g_lay.page_bytes >> 4, GPH_SMB, D.seq + o0,
                           P_ > g_lay.huge_P ? 0 : (int)(o1 - o0) >> 4);
  GPH_SYNC();
}
GPH_DEV void stage_in(const GphDev &D, int g, const char *pages, int withSeq)
{
  int P_; uint64_t co_;
  stage_in_copy(D, g, pages, withSeq, P_, co_);
  load_scalars();
  scratch_init(D, g, P_, co_);
}
// ---- the EVALUATED state of a global proposal (UpdateTau / UpdateSampleAge / mixing) waits in the shadow page for the
// decision.  An evaluation changes the node records (ages), the scalars and lists behind the saved copies -- and the event
// pool ONLY when its ripple created events (IS_RB_NUM != 0: a migration event or a band time moved in a distant population,
// rare).  Round 6: the shadow page is SPARSE -- without a ripple only the node records and the tail behind the saved copies
// are written (the event pool, 45 % of the page, and the saved copies, which only a revert reads, are those of the main page);
// whoever takes the evaluated state merges the two.  2.1 KB less written per locus and evaluation (variant s).
GPH_DEV void stage_out_evaluated(const GphDev &D, int g)
{
  flush_scalars();
  GPH_SYNC();
  char *sh = D.shadow + (size_t)g * g_lay.page_bytes;
  if (ISC(IS_RB_NUM) != 0) { page_out(sh); return; }
  gph_copy16_out(sh + g_lay.o_nd, GPH_LDSP(&gph_lds.nd), g_lay.N);
  gph_copy16_out(sh + g_lay.o_mig_age, GPH_LDSP(&gph_lds.mig_age), (g_lay.page_bytes - g_lay.o_mig_age) >> 4);
}
GPH_DEV void stage_in_evaluated(const GphDev &D, int g, int withSeq)
{
  int P_; uint64_t co_;
  stage_in_copy(D, g, D.pages, withSeq, P_, co_);
  const char *sh = D.shadow + (size_t)g * g_lay.page_bytes;
  const int32_t *is = (const int32_t *)(sh + g_lay.o_iscal);
  if (RFL(is[IS_RB_NUM]) != 0) {
    page_in(sh);
  } else {
    gph_copy16_in(GPH_LDSP(&gph_lds.nd), sh + g_lay.o_nd, g_lay.N);
    gph_copy16_in(GPH_LDSP(&gph_lds.mig_age), sh + g_lay.o_mig_age, (g_lay.page_bytes - g_lay.o_mig_age) >> 4);
  }
  GPH_SYNC();
  load_scalars();
  scratch_init(D, g, P_, co_);
}
GPH_DEV void stage_out(const GphDev &D, int g, char *pages, int unused)
{
  (void)unused;
  flush_scalars();
  GPH_SYNC();
  page_out(pages + (size_t)g * g_lay.page_bytes);
  if (pages == D.pages) {
    /* compact copy of the sufficient statistics for the totals reduction (computeTotalStats) */
    const int K = g_lay.K, B = g_lay.B, C = 2 * K + 2 * B;
    for (int c = GPH_LANE; c < C; c += GPH_NLANES) {
      double v;
      if (c < K) v = gph_lds.coal[c];
      else if (c < 2 * K) v = (double)gph_lds.ncoal[c - K];
      else if (c < 2 * K + B) v = gph_lds.migst[c - 2 * K];
      else v = (double)gph_lds.nmig[c - 2 * K - B];
      D.stats[(size_t)g * C + c] = v;
    }
  }
}
// algorithmic bytes of the launch's useOld evaluations: 96 R P + 20 N + 8 U + 8 per evaluation that recomputed
// anything -- derived ONCE here from the counters (recomputed nodes, evaluations, the empty ones) instead of being summed in
// LDS by every evaluation: the same integers, a dozen instructions less per evaluation
GPH_DEV double eval_bytes(int U)
{
  const int P = CNT(CN_P), full = CNT(CN_EVALS) - CNT(CN_EMPTY);
  if (P <= 0 || full <= 0) return 0.0;
  /* U (the unphased patterns of the locus) comes from the per-locus table: counting the phase words here was 70 of the 3 300
   * instructions of a tau evaluation -- the measurement on the hot path again (tools/bbcount.sh, round 5) */
  return (double)(96ll * (CNT(CN_NODES) - CNT(CN_NODES0)) * P + (long long)(20 * g_lay.N + 8 * U + 8) * full);
}
GPH_DEV void out_common(const GphDev &D, int g)
{
  const double bytes_ = eval_bytes(D.P[2 * g + 1]);
  if (GPH_LANE == 0) {
    double *o = D.out + (size_t)g * GPH_OUT_SLOTS;
    o[8] = CNT(CN_EVALS);
    o[9] = CNT(CN_NODES);
    o[10] = bytes_;
    /* max over loci = the FIRST failing locus (smallest global index) with its own code: (2^30 - locus) 2^14 + code, exact in
     * a double (GphGlobal::error / error_locus, gph_global.h: gg_count; the reference names the locus: GPhoCS.c:660-676) */
    o[11] = gph_errcode() != 0 ? (double)((((long long)1 << 30) - (long long)(D.orig[g] + D.locus_begin)) * 16384 + (gph_errcode() & 16383)) : 0.0;
    o[13] = CNT(CN_NOTENOUGH);
    if (gph_errcode() != 0) {
      gph_raise(D.err, gph_errcode());
    }
  }
}
#define OUT(g, k, v) do { if (GPH_LANE == 0) D.out[(size_t)(g) * GPH_OUT_SLOTS + (k)] = (v); } while (0)

// ---------------------------------------------------------------- init
// Coalescence1Pop, patch.c:279-358, iterative over the population post-order (the
// reference recursion visits sons[0], sons[1], then the population itself).  The
// living-lineage array is packed exactly as the recursion packs it: a cursor
// advances past each finished population's SURVIVING lineages, so an ancestral
// population's list is the contiguous run [base(son0), cursor).
GPH_DEV void random_gtree(GphRng &rng)
{
  const int n = g_lay.n;
  int pi, pop, nextId = n, num, node1, node2, choice, a, b, base, cur = 0;
  double t, T;
  for (pi = 0; pi < g_lay.K; pi++) {
    pop = g_model.postOrder[pi];
    if (pop < g_lay.Kc) {
      node1 = pop > 0 ? g_model.cumSamples[pop - 1] : 0;
      num = g_model.cumSamples[pop] - node1;
      base = cur;
      for (node2 = 0; node2 < num; ++node2) {
        si16(&GphLds::s_targets, base + node2, node1 + node2);
        setNPOP(node1 + node2, pop);
        setNEV(node1 + node2, -1);
        setLEFT(node1 + node2, -1);
        setRGHT(node1 + node2, -1);
        setFATH(node1 + node2, -1);
        setAGE(node1 + node2, g_model.sampleAge[pop]);
      }
    } else {
      base = gi16(&GphLds::s_dpops, 0, g_model.popSon0[pop]);   /* (the pending-delta lists are free here) */
      num = cur - base;
    }
    si16(&GphLds::s_dpops, 0, pop, base);
    T = g_model.popAge[pop];
    if (pop < g_lay.Kc) T = g_model.sampleAge[pop];
    for (; num > 1; num--, nextId++) {
      t = -(g_model.theta[pop] / (num * (num - 1.))) * gph_log_u(l_rndu(rng));
      T += t;
      if (pop != g_lay.rootPop && T > g_model.popAge[g_model.popFather[pop]]) break;
      choice = (int)(num * l_rndu(rng));
      a = gi16(&GphLds::s_targets, base + choice);
      si16(&GphLds::s_targets, base + choice, gi16(&GphLds::s_targets, base + num - 1));
      choice = (int)((num - 1) * l_rndu(rng));
      b = gi16(&GphLds::s_targets, base + choice);
      si16(&GphLds::s_targets, base + choice, nextId);
      setRGHT(nextId, a);
      setLEFT(nextId, b);
      setFATH(nextId, -1);
      setAGE(nextId, T);
      setFATH(a, nextId);
      setFATH(b, nextId);
      setNPOP(nextId, pop);
    }
    cur = base + num;
  }
  setISC(IS_ROOT, nextId - 1);
}

// initializeMCMC per-locus body, GPhoCS.c:1197-1214
// preDraws: rndu() draws the locus's stream has already spent before the genealogy is sampled (the VAR-rate
// start-up draws one per locus, GPhoCS.c:1163)
GPH_DEV void kb_init(const GphDev &D, int g, uint32_t seedz, double mutRate, int preDraws)
{
  int i;
  /* blank page */
  for (i = GPH_LANE; i < (int)(sizeof(GphLds) / 4); i += GPH_NLANES) ((GPH_LDS int32_t *)&gph_lds)[i] = 0;
  if (D.P[2 * g] <= g_lay.huge_P) copy16_g2l(0, D.seq + D.seq_off[g], (int)(D.seq_off[g + 1] - D.seq_off[g]));
  GPH_SYNC();
  scratch_init(D, g, D.P[2 * g], D.cond_off[g]);
  setISC(IS_RX, 11);
  setISC(IS_RY, 23);
  setISC(IS_RZ, (int)seedz);
  setISC(IS_SV_ROOT, -1);
  setFS(FS_MUTRATE, mutRate);
  for (i = 0; i < GPH_MAX_MIGS; i++) {
    setMG(i, MG_BRANCH, -1); setMG(i, MG_BAND, -1); setMG(i, MG_SPOP, -1);
    setMG(i, MG_TPOP, -1); setMG(i, MG_SEV, -1); setMG(i, MG_TEV, -1);
  }
  {
    GphRng rng;
    rng_load(rng);
    for (i = 0; i < preDraws; i++) (void)l_rndu(rng);
    random_gtree(rng);
    rng_store(rng);
  }
  construct_event_chain();
  compute_genetree_stats();
  setFS(FS_GENLNL, gtree_lnl());
  lik_compute(0);
  lik_reset_saved();
  OUT(g, 0, FS(FS_GENLNL));
  OUT(g, 1, FS(FS_DATALNL));
  out_common(D, g);
  stage_out(D, g, D.pages, 1);
}

// ---------------------------------------------------------------- genealogy sweeps
// UpdateGB_InternalNode per-locus body, GPhoCS.c:2299-2425
template <class RNG> GPH_DEV void sweep_internal(const GphDev &D, int g, double finetune, RNG &rng)
{
  int pop, inode, i, son, mig, acc = 0;
  double t, tnew, lnacc, lnLd, dgen, tb0, tb1;
  /* the accumulators and the step size live in LDS scratch, not in registers that stay allocated (and get
   * spilled) across the whole sweep: they are touched once per proposal */
  sf64(&GphLds::s_cntf, 2, 0.0);
  sf64(&GphLds::s_cntf, 3, 0.0);
  sf64(&GphLds::s_cntf, 7, finetune);
  for (inode = g_lay.n; inode < g_lay.N; inode++) {
    const GphNodeS me = ld_node(inode);
    t = me.age;
    pop = me.npop;
    tb0 = g_model.popAge[pop];
    if (pop != g_lay.rootPop) tb1 = g_model.popAge[g_model.popFather[pop]];
    else tb1 = GPH_OLDAGE;
    int migs[2];
    mig_bounds(inode, me.left, me.right, mig, migs[0], migs[1]);
    if (mig >= 0) tb1 = gmin2(tb1, MAGE(mig));
    else if (inode != ISC(IS_ROOT)) tb1 = gmin2(tb1, AGE(me.father));
    for (i = 0; i < 2; i++) {
      son = i == 0 ? me.left : me.right;
      mig = migs[i];
      if (mig >= 0) tb0 = gmax2(tb0, MAGE(mig));
      else tb0 = gmax2(tb0, AGE(son));
    }
    tnew = t + gf64(&GphLds::s_cntf, 7) * l_rnd2normal8(rng);
    tnew = l_reflect(tnew, tb0, tb1);
    if (UNI(fabs(tnew - t) < 1e-15)) { acc++; continue; }
    GPH_SLOG(1, inode, t, tnew, 0, 0, 0);
    lik_adjust_age(inode, tnew);
    lnLd = -FS(FS_DATALNL);
    { STAMP_BEGIN(1); lnLd += lik_compute(1); STAMP_END(1); }
    { STAMPA_BEGIN(2); dgen = consider_event_move(0, NEV(inode), pop, t, pop, tnew); STAMPA_END(2); }
    lnacc = dgen + lnLd;
    if (gph_failed()) break;
    const bool take_ = UNI(lnacc >= 0) || UNI(l_rndu(rng) < gph_exp_u(lnacc));
    GPH_SLOG(3, take_, lnacc, 0, 0, 0, 0);
    if (take_) {
      acc++;
      setFS(FS_GENLNL, FS(FS_GENLNL) + dgen);
      sf64(&GphLds::s_cntf, 2, gf64(&GphLds::s_cntf, 2) + lnLd);
      sf64(&GphLds::s_cntf, 3, gf64(&GphLds::s_cntf, 3) + (dgen + lnLd) / D.Ltot);
      accept_event_chain_changes(0);
      lik_reset_saved();
    } else {
      reject_event_chain_changes(0);
      lik_revert();
    }
  }
  OUT(g, 0, acc);
  OUT(g, 3, gf64(&GphLds::s_cntf, 2));
  OUT(g, 4, gf64(&GphLds::s_cntf, 3));
}

// UpdateGB_MigrationNode per-locus body, GPhoCS.c:2453-2587
template <class RNG> GPH_DEV void sweep_mignodes(const GphDev &D, int g, double finetune, RNG &rng)
{
  int mi, mignode, pop_s, pop_t, ev_s, ev_t, below, mig_below, mig_above, father, acc = 0, totmigs = 0;
  double t, tnew, tb0, tb1, dgen, lnacc, dLog = 0;
  for (mi = 0; mi < ISC(IS_NUM_MIGS); mi++) {
    mignode = LIVING(mi);
    t = MAGE(mignode);
    pop_s = MG(mignode, MG_SPOP);
    pop_t = MG(mignode, MG_TPOP);
    ev_s = MG(mignode, MG_SEV);
    ev_t = MG(mignode, MG_TEV);
    below = MG(mignode, MG_BRANCH);
    tb0 = g_model.bandStart[MG(mignode, MG_BAND)];
    tb1 = g_model.bandEnd[MG(mignode, MG_BAND)];
    mig_below = find_last_mig(below, t);
    mig_above = find_first_mig(below, t);
    if (mig_below >= 0) tb0 = gmax2(tb0, MAGE(mig_below));
    else tb0 = gmax2(tb0, AGE(below));
    if (mig_above >= 0) tb1 = gmin2(tb1, MAGE(mig_above));
    else {
      father = FATH(below);
      if (father < 0) tb1 = gmin2(tb1, GPH_OLDAGE);
      else tb1 = gmin2(tb1, AGE(father));
    }
    tnew = t + finetune * l_rnd2normal8(rng);
    tnew = l_reflect(tnew, tb0, tb1);
    if (UNI(fabs(tnew - t) < 1e-15)) { acc++; continue; }
    GPH_SLOG(4, mignode, t, tnew, 0, 0, 0);
    dgen = consider_event_move(0, ev_s, pop_s, t, pop_s, tnew);
    dgen += consider_event_move(1, ev_t, pop_t, t, pop_t, tnew);
    lnacc = dgen;
    if (gph_failed()) break;
    const bool take_ = UNI(lnacc >= 0) || UNI(l_rndu(rng) < gph_exp_u(lnacc));
    GPH_SLOG(3, take_, lnacc, 0, 0, 0, 0);
    if (take_) {
      acc++;
      setFS(FS_GENLNL, FS(FS_GENLNL) + dgen);
      dLog += dgen / D.Ltot;
      accept_event_chain_changes(0);
      accept_event_chain_changes(1);
      setMAGE(mignode, tnew);
    } else {
      reject_event_chain_changes(0);
      reject_event_chain_changes(1);
    }
  }
  for (mi = 0; mi < g_lay.B; mi++) totmigs += NMIGB(mi);
  OUT(g, 1, acc);
  OUT(g, 5, dLog);
  OUT(g, 12, totmigs);
}

// UpdateGB_MigSPR per-locus body, GPhoCS.c:2610-2944 (no admixture)
template <class RNG> GPH_DEV void sweep_spr(const GphDev &D, int g, RNG &rng)
{
  int node, res, father, father_pop_old, sibling, b, i, mig, ev, target, pop, acc = 0, fpn, fen;
  double lnLd, lnacc, t_new;
  sf64(&GphLds::s_cntf, 5, 0.0);
  sf64(&GphLds::s_cntf, 6, 0.0);
  for (node = 0; node < g_lay.N; node++) {
    if (node == ISC(IS_ROOT)) continue;
    father = FATH(node);
    { const GphNodeS F_ = ld_node(father);     /* one LDS round trip for the father's record */
      father_pop_old = F_.npop;
      sibling = F_.left + F_.right - node; }
    GPH_SLOG(5, node, father, father_pop_old, 0, 0, 0);
#if GPH_BIG_BANDS || defined(GPH_TWO_WALKS)
    { STAMPA_BEGIN(3); trace_lineage<0>(node, rng); STAMPA_END(3); }
    { STAMPA_BEGIN(4); res = trace_lineage<1>(node, rng); STAMPA_END(4); }
#else
    { STAMPA_BEGIN(3); res = trace_pair(node, rng); STAMPA_END(3); }      /* both walks, their common prefix once (gph_locus.h) */
#endif
#ifdef GPH_WALKSTAT
    { extern long long gph_ws[4]; int n0 = DI(0, DI_NEV), n1 = DI(1, DI_NEV), c = 0;
      while (c < n0 && c < n1 && gph_lds.s_dev[0][c] == gph_lds.s_dev[1][c]) c++;
      gph_ws[0] += n0; gph_ws[1] += n1; gph_ws[2] += c; gph_ws[3] += 1; }
#endif
    lnLd = -FS(FS_DATALNL);
    { STAMP_BEGIN(1); lnLd += lik_compute(1); STAMP_END(1); }
    lnacc = lnLd;
    if (gph_failed()) break;
    const bool take_ = res >= 0 && (UNI(lnacc >= 0) || UNI(l_rndu(rng) < gph_exp_u(lnacc)));
    GPH_SLOG(3, take_, lnacc, 0, 0, 0, 0);
    if (take_) {
      STAMPC_BEGIN(2);
      acc++;
      setFS(FS_GENLNL, FS(FS_GENLNL) + (SPRLN(1) - SPRLN(0)));
      sf64(&GphLds::s_cntf, 5, gf64(&GphLds::s_cntf, 5) + lnLd);
      sf64(&GphLds::s_cntf, 6, gf64(&GphLds::s_cntf, 6) + (lnLd - SPRLN(0) + SPRLN(1)) / D.Ltot);
      target = SPRI(SI_TARGET);
      t_new = AGE(father);
      for (i = 0; i < ISC(IS_NUM_MIGS); i++) {
        mig = LIVING(i);
        if (MG(mig, MG_BRANCH) == father) setMG(mig, MG_BRANCH, sibling);
        if (target == father) target = sibling;
        if (MG(mig, MG_BRANCH) == target && MAGE(mig) >= t_new) setMG(mig, MG_BRANCH, father);
      }
      remove_event(SPRI(SI_FEV_OLD), father_pop_old);
      fen = SPRI(SI_FEV_NEW);
      setETYPE(fen, GPH_COAL);
      setENODE(fen, father);
      setNEV(father, fen);
      fpn = SPRI(SI_FPOP_NEW);
      if (fpn != father_pop_old) {
        setNPOP(father, fpn);
        setNCOAL(father_pop_old, NCOAL(father_pop_old) - 1);
        setNCOAL(fpn, NCOAL(fpn) + 1);
      }
      replace_mig_nodes(node);
      /* one lane per list entry / band / population (entries are distinct) */
      (void)ev; (void)b; (void)pop;
      GPH_EACH(k, DI(1, DI_NEV)) { const int q = gph_lds.s_dev[1][k]; gph_lds.ev[q].nlin = (uint8_t)(gph_lds.ev[q].nlin + 1); }
      GPH_EACH1(k, g_lay.B) gph_lds.migst[k] = gph_lds.migst[k] + (gph_lds.s_dmig[1][k] - gph_lds.s_dmig[0][k]);
      GPH_EACH1(k, g_lay.K) gph_lds.coal[k] = gph_lds.coal[k] + (gph_lds.s_dcoal[1][k] - gph_lds.s_dcoal[0][k]);
      lik_reset_saved();
      STAMPC_END(2);
    } else {
      STAMPC_BEGIN(3);
      if (res >= 0) remove_event(SPRI(SI_FEV_NEW), SPRI(SI_FPOP_NEW));
      for (i = 0; i < SPRI(SI_NNEW); i++) {
        b = SPRA(SA_NEWBAND, i);
        remove_event(SPRA(SA_NEWIN, i), g_model.bandTgt[b]);
        remove_event(SPRA(SA_NEWOUT, i), g_model.bandSrc[b]);
      }
      GPH_EACH(k, DI(0, DI_NEV)) { const int q = gph_lds.s_dev[0][k]; gph_lds.ev[q].nlin = (uint8_t)(gph_lds.ev[q].nlin + 1); }
      lik_revert();
      STAMPC_END(3);
    }
  }
  OUT(g, 2, acc);
  OUT(g, 6, gf64(&GphLds::s_cntf, 5));
  OUT(g, 7, gf64(&GphLds::s_cntf, 6));
}

// fused genealogy sweep: UpdateGB_InternalNode, UpdateGB_MigrationNode, UpdateGB_MigSPR
// run back to back on the LDS-resident locus (GPhoCS.c:1495-1538 calls them in this
// order with nothing in between) -- one load and one store of the locus instead of three
// flag 16: the mixing proposal of the previous iteration was ACCEPTED and its commit (GPhoCS.c:4815-4848) has not run: the
// evaluated state is taken from the shadow page and committed here, in LDS -- the page the sweep writes at its end is
// the only one written, and the commit costs no launch and no page round trip of its own
GPH_DEV void kb_sweep(const GphDev &D, int g, int flags, double ftCoal, double ftMig, double mix_c, double mix_lnc)
{
  STAMP_BEGIN(0);
  if (flags & 16) {
    stage_in_evaluated(D, g, 1);
    mix_commit_body(mix_c, mix_lnc);
  } else {
    stage_in(D, g, D.pages, 1);
  }
  GPH_SLOG_OPEN(D, g);
  GphRngB rng;       /* uniforms in batches of 64: gph_locus.h */
  rng_load(rng);
  /* flag 8: synchronizeEvents of the previous iteration (patch.c:3548), deferred into this kernel */
  OUT(g, 15, (flags & 8) ? (double)synchronize_events() : 1.0);
  OUT(g, 0, 0.0); OUT(g, 1, 0.0); OUT(g, 2, 0.0); OUT(g, 3, 0.0); OUT(g, 4, 0.0);
  OUT(g, 5, 0.0); OUT(g, 6, 0.0); OUT(g, 7, 0.0); OUT(g, 12, 0.0);
  { STAMP_BEGIN(5); if ((flags & 1) && ftCoal > 0.0) sweep_internal(D, g, ftCoal, rng); STAMP_END(5); }
  { STAMPC_BEGIN(4); if ((flags & 2) && ftMig > 0.0 && !gph_failed()) sweep_mignodes(D, g, ftMig, rng); STAMPC_END(4); }
  { STAMP_BEGIN(6); if ((flags & 4) && !gph_failed()) sweep_spr(D, g, rng); STAMP_END(6); }
  rng_store(rng);
  out_common(D, g);
  STAMP_END(0);
#if defined(GPH_STAMPS) && !defined(GPH_HOSTEMU)
  /* diagnostic build: the result slots carry cycle sums instead (tools/stamp_breakdown.py) */
  for (int k = 0; k < 8; k++) OUT(g, k, gph_lds.s_stamp[k]);
#endif
  stage_out(D, g, D.pages, 1);
}

// ---------------------------------------------------------------- UpdateTau
// loop 1 body of UpdateTau, GPhoCS.c:3491-3833.  Reads the main page, writes the
// evaluated state to the SHADOW page (see gph_types.h); new conditionals go to the
// non-current halves.  out: 0 ntj0, 1 ntj1, 2 conflict, 3 genDelta, 4 dataDelta
GPH_DEV void kb_tau_eval(const GphDev &D, int g, gph_ctau &A, int fuse)
{
  const int ap = A.ap, s0 = A.son0, s1 = A.son1;
  double age_mt, new_age = 0.0, dGen = 0, dData = 0, gd;
  int srcP, tgtP, fatherNode, inode, inORout = -1, ev = -1, n1_0 = 0, n1_1 = 0, i, mig, mig1, band, pop;
  int conflict = 0, k;
  stage_in_after_finish(D, g, fuse);
  setISC(IS_CONFLICT_LOG, 1);
  setISC(IS_RB_NUM, 0);
  for (i = 0; i < ISC(IS_NUM_MIGS); i++) {
    if (conflict == 0) {
      pop = -1;
      mig = LIVING(i);
      band = MG(mig, MG_BAND);
      srcP = MG(mig, MG_SPOP);
      tgtP = MG(mig, MG_TPOP);
      age_mt = MAGE(mig);
      if (age_mt < A.taub0 || age_mt > A.taub1) continue;
      if (A.mode) {
        /* UpdateSampleAge, GPhoCS.c:4222-4243: below / above the old sample age */
        const int up = age_mt > A.tauold;
        const double tb = up ? A.taub1 : A.taub0, tf = up ? A.taufactor1 : A.taufactor0;
        if (srcP == ap) {
          inORout = 1;
          ev = MG(mig, MG_TEV);
          pop = tgtP;
          new_age = tb + tf * (age_mt - tb);
          if (up) n1_1++; else n1_0++;
        } else if (tgtP == ap) {
          inORout = 0;
          ev = MG(mig, MG_SEV);
          pop = srcP;
          new_age = tb + tf * (age_mt - tb);
          if (up) n1_1++; else n1_0++;
        }
      } else
      if ((srcP == s0 && tgtP == s1) || (srcP == s1 && tgtP == s0)) {
        n1_0++;
      } else if (srcP == ap) {
        inORout = 1;
        ev = MG(mig, MG_TEV);
        pop = tgtP;
        new_age = A.taub1 + A.taufactor1 * (age_mt - A.taub1);
        n1_1++;
      } else if (tgtP == ap) {
        inORout = 0;
        ev = MG(mig, MG_SEV);
        pop = srcP;
        new_age = A.taub1 + A.taufactor1 * (age_mt - A.taub1);
        n1_1++;
      } else if ((srcP == s0 || srcP == s1) && MAGE(mig) > A.taub0) {
        inORout = 1;
        ev = MG(mig, MG_TEV);
        pop = tgtP;
        new_age = A.taub0 + A.taufactor0 * (age_mt - A.taub0);
        n1_0++;
      } else if ((tgtP == s0 || tgtP == s1) && MAGE(mig) > A.taub0) {
        inORout = 0;
        ev = MG(mig, MG_SEV);
        pop = srcP;
        new_age = A.taub0 + A.taufactor0 * (age_mt - A.taub0);
        n1_0++;
      }
      if (ev >= 0) {
        inode = MG(mig, MG_BRANCH);
        if (new_age >= g_model.bandEnd[band]) conflict = 1;
        else if (new_age <= g_model.bandStart[band]) conflict = 1;
        else if (inORout == 0 && new_age > age_mt) {
          fatherNode = FATH(inode);
          mig1 = find_first_mig(inode, MAGE(mig));
          if (mig1 >= 0 && MG(mig1, MG_SPOP) != ap && MG(mig1, MG_SPOP) != s0 && MG(mig1, MG_SPOP) != s1 &&
              new_age >= MAGE(mig1))
            conflict = 1;
          else if (fatherNode >= 0 && new_age >= AGE(fatherNode))
            conflict = 1;
        } else if (inORout == 1 && new_age < age_mt) {
          mig1 = find_last_mig(inode, MAGE(mig));
          if (mig1 >= 0 && MG(mig1, MG_TPOP) != ap && MG(mig1, MG_TPOP) != s0 && MG(mig1, MG_TPOP) != s1 &&
              new_age <= MAGE(mig1))
            conflict = 1;
          else if (new_age <= AGE(inode))
            conflict = 1;
        }
        if (conflict != 1) {
          k = ISC(IS_RB_NUM);
          setRBI(0, k, ev);
          setRBI(2, k, pop);
          setRBAGE(k, new_age);
          setISC(IS_RB_NUM, k + 1);
          ev = -1;
        }
      }
    }
  }
  if (conflict) {
    setISC(IS_RB_NUM, 0);
  } else {
    for (i = 0; i < A.num_aff; i++) {
      int guard = 0;
      band = A.aff_bands[i];
      tgtP = g_model.bandTgt[band];
      for (ev = FIRSTEV(tgtP); ev >= 0;) {
        const GphEvS R = ld_ev(ev);
        if (R.node == band &&
            ((R.type == GPH_MIG_BAND_START && A.start_or_end[i]) || R.type == GPH_MIG_BAND_END))
          break;
        if (++guard > GPH_CAP_E) { ev = -1; break; }
        ev = R.next;
      }
      if (ev < 0) { gph_fail(74); break; }
      k = ISC(IS_RB_NUM);
      setRBI(0, k, ev);
      setRBI(2, k, tgtP);
      setRBAGE(k, A.new_band_ages[i]);
      setISC(IS_RB_NUM, k + 1);
    }
    if (!gph_failed()) {
      gd = rubber_band_ripple(1);
      if (A.mode) {
        gd += rubber_band(ap, g_model.popAge[ap], A.taub1, A.tauold, A.taufactor1, 0, &n1_1);
        gd += rubber_band(ap, g_model.popAge[ap], A.taub0, A.tauold, A.taufactor0, 0, &n1_0);
      } else {
        if (A.isRoot) gd += rubber_band(ap, g_model.popAge[ap], A.taub0, A.tauold, A.taufactor1, 0, &n1_1);
        else gd += rubber_band(ap, g_model.popAge[ap], A.taub1, A.tauold, A.taufactor1, 0, &n1_1);
        gd += rubber_band(s0, g_model.popAge[s0], A.taub0, A.tauold, A.taufactor0, 0, &n1_0);
        gd += rubber_band(s1, g_model.popAge[s1], A.taub0, A.tauold, A.taufactor0, 0, &n1_0);
      }
      setFS(FS_GENDELTA, gd);
      dGen += gd;
      if (A.mode || n1_0 + n1_1) {     /* UpdateSampleAge always re-evaluates (GPhoCS.c:4431) */
        dData -= FS(FS_DATALNL);
        dData += lik_compute(1, true);
      }
    }
  }
  OUT(g, 0, conflict ? 0 : n1_0);
  OUT(g, 1, conflict ? 0 : n1_1);
  OUT(g, 2, conflict);
  OUT(g, 14, conflict ? (double)(D.orig[g] + D.locus_begin) : 1e300);
  OUT(g, 3, dGen);
  OUT(g, 4, dData);
  out_common(D, g);
  stage_out_evaluated(D, g);
}

// loop 2 body (commit), GPhoCS.c:3885-3936 (+ adjustRootEvents patch.c:1808 for the root), on the evaluated state in
// the LDS image; F = the decided proposal as gg_tau_decide froze it
GPH_DEV void tau_commit_body(gph_cfin &F)
{
  int dummy = 0, i, mig, nw, ev;
  double age;
  setFS(FS_GENLNL, FS(FS_GENLNL) + FS(FS_GENDELTA));
  if (F.mode) {   /* UpdateSampleAge commit, GPhoCS.c:4500-4509 */
    rubber_band(F.ap, F.age_ap, F.taub1, F.tauold, F.taufactor1, 1, &dummy);
    rubber_band(F.ap, F.age_ap, F.taub0, F.tauold, F.taufactor0, 1, &dummy);
  } else {
    if (F.isRoot) rubber_band(F.ap, F.age_ap, F.taub0, F.tauold, F.taufactor1, 1, &dummy);
    else rubber_band(F.ap, F.age_ap, F.taub1, F.tauold, F.taufactor1, 1, &dummy);
    rubber_band(F.son0, F.age_s0, F.taub0, F.tauold, F.taufactor0, 1, &dummy);
    rubber_band(F.son1, F.age_s1, F.taub0, F.tauold, F.taufactor0, 1, &dummy);
  }
  lik_reset_saved();
  for (i = 0; i < ISC(IS_RB_NUM); i++) {
    nw = RBI(1, i);
    mig = ENODE(nw);
    if (ETYPE(nw) == GPH_IN_MIG) {
      setMG(mig, MG_TEV, nw);
      setMAGE(mig, RBAGE(i));
    } else if (ETYPE(nw) == GPH_OUT_MIG) {
      setMG(mig, MG_SEV, nw);
    }
    remove_event(RBI(0, i), RBI(2, i));
  }
  setISC(IS_RB_NUM, 0);
  if (F.isRoot) {
    int guard = 0;
    ev = FIRSTEV(g_lay.rootPop);
    age = F.taunew;
    for (GphEvS R = ld_ev(ev); R.next >= 0; R = ld_ev(ev)) { age += R.time; ev = R.next; if (++guard > GPH_CAP_E) { gph_fail(97); break; } }
    setEVT(ev, GPH_OLDAGE - age);
  }
}
// loops 3/4 (reject), GPhoCS.c:3965-3989: only loci whose ripple moved events differ from their main page; everything
// else is bit-identical already (and loci past the first conflicting one were never touched by the serial reference)
GPH_DEV bool tau_revert_needed(const GphDev &D, int g, gph_cfin &F)
{
  if (D.orig[g] + D.locus_begin >= F.limit) return false;
  const int32_t *is = (const int32_t *)(D.shadow + (size_t)g * g_lay.page_bytes + g_lay.o_iscal);
  return RFL(is[IS_RB_NUM]) != 0;
}

// the finish of the decided proposal as a kernel of its own (the stepwise entry points; the iteration's last one when
// no evaluate kernel follows): commit or revert by the flag the decision stage froze (gph_global.h: gg_tau_decide)
GPH_DEV void kb_tau_finish(const GphDev &D, int g)
{
  gph_cfin &F = GPH_G->fin;
  if (RFL(F.flag)) {
    stage_in_evaluated(D, g, 0);
    tau_commit_body(F);
  } else {
    if (!tau_revert_needed(D, g, F)) return;
    stage_in_evaluated(D, g, 0);
    lik_revert();
    rubber_band_ripple(0);
  }
  out_common(D, g);
  stage_out(D, g, D.pages, 0);
}

// stage-in for a kernel that runs right behind a decision: with fuse != 0 the finish of the decided proposal is done
// first, in place -- accepted: the evaluated state comes from the shadow page, is committed in LDS and written to the
// main page (one page read less than finish kernel + stage-in, and one launch less); rejected: only loci whose ripple
// moved events take that route.  Either way the LDS image is the locus's current state when this returns.
GPH_DEV void stage_in_after_finish(const GphDev &D, int g, int fuse)
{
  if (!fuse) { stage_in(D, g, D.pages, 1); return; }
  gph_cfin &F = GPH_G->fin;
  if (RFL(F.flag)) {
    stage_in_evaluated(D, g, 1);
    tau_commit_body(F);
    stage_out(D, g, D.pages, 0);
  } else if (tau_revert_needed(D, g, F)) {
    stage_in_evaluated(D, g, 1);
    lik_revert();
    rubber_band_ripple(0);
    stage_out(D, g, D.pages, 0);
  } else {
    stage_in(D, g, D.pages, 1);
  }
}

// ---------------------------------------------------------------- mixing
// evaluate loop of mixing(), GPhoCS.c:4790-4801.  out: 0 dataDelta
GPH_DEV void kb_mix_eval(const GphDev &D, int g, double c, int fuse)
{
  double d;
  stage_in_after_finish(D, g, fuse);
  d = lik_scale_ages(c);
  OUT(g, 0, d);
  out_common(D, g);
  stage_out_evaluated(D, g);
}
// commit loop of mixing(), GPhoCS.c:4815-4848 + adjustRootEvents (patch.c:1808), on the evaluated state in the LDS image
GPH_DEV void mix_commit_body(double c, double lnc)
{
  int i, pop, b, ev;
  double age;
  lik_reset_saved();
  for (i = 0; i < ISC(IS_NUM_MIGS); i++) setMAGE(LIVING(i), MAGE(LIVING(i)) * c);
  setFS(FS_GENLNL, FS(FS_GENLNL) - lnc * (g_lay.n - 1 + ISC(IS_NUM_MIGS)));
  for (pop = 0; pop < g_lay.K; pop++) setCOALS(pop, COALS(pop) * c);
  for (b = 0; b < g_lay.B; b++) setMIGST(b, MIGST(b) * c);
  for (i = GPH_LANE; i < g_lay.E; i += GPH_NLANES) {
    double t = gph_lds.ev[i].time;
    if (t > 0) gph_lds.ev[i].time = t * c;
  }
  GPH_SYNC();
  {
    int guard = 0;
    ev = FIRSTEV(g_lay.rootPop);
    age = g_model.popAge[g_lay.rootPop];
    for (GphEvS R = ld_ev(ev); R.next >= 0; R = ld_ev(ev)) { age += R.time; ev = R.next; if (++guard > GPH_CAP_E) { gph_fail(97); break; } }
    setEVT(ev, GPH_OLDAGE - age);
  }
}
GPH_DEV void kb_mix_commit(const GphDev &D, int g, double c, double lnc)
{
  stage_in_evaluated(D, g, 0);
  mix_commit_body(c, lnc);
  out_common(D, g);
  stage_out(D, g, D.pages, 0);
}

// mixing reject (GPhoCS.c:4881-4887): revertToSaved restores every locus exactly, and the evaluated state only
// ever lived in the shadow pages -- nothing to do
GPH_DEV void kb_mix_finish(const GphDev &D, int g)
{
  if (RFL(GPH_G->mix_flag)) kb_mix_commit(D, g, GPH_G->mix_c, GPH_G->mix_lnc);
}

// ---------------------------------------------------------------- locus rates
// UpdateLocusRate, GPhoCS.c:4598-4680.  The proposal is serial over loci by construction: every locus g trades
// rate with the reference locus (genRateRef = 0), so the decision at g needs the reference locus's rate and
// likelihood after every earlier decision.  What does NOT depend on the earlier decisions is done in parallel:
//   kb_lrate_prep   one wavefront per locus, all loci at once: the step is drawn from the locus's own stream and
//                   the locus is recomputed at the rate the proposal almost always lands on (the step reflected
//                   at zero only; the upper bound rold + rref is the serial part).  The evaluated state goes to
//                   the shadow page (the main page and its current conditionals stay as they are); the scan gets
//                   a 64-byte record per locus, in input order.
//   kb_lrate_scan   ONE wavefront walks the records in input order: reflect against the actual bounds, take the
//                   prepared likelihood when the rate is the prepared one (bit for bit; otherwise evaluate the
//                   locus with the stateless evaluator lik_private), evaluate the reference locus at its new rate
//                   (lik_private: its node records and sequence block stay in LDS for the whole scan), decide,
//                   carry the reference locus's rate / likelihood and the three accumulators
//                   (dataLogLikelihood, logLikelihood, rateVar: same additions in the same order as the reference
//                   loop).  Writes one decision record per locus; touches no locus state.
//   kb_lrate_apply  one wavefront per locus, all loci at once: RNG state written back; accepted with the prepared
//                   rate = the shadow page becomes the page (resetSaved); accepted with another rate = recomputed
//                   in place (computeLocusDataLikelihood(0) + resetSaved) and checked against the scanned value bit
//                   for bit.  The reference locus ends at its last accepted rate; its buffer parity is the number
//                   of accepted proposals mod 2 (every accepted proposal flipped all its nodes once).
struct GphLrRec { double rate, lnl; uint32_t rx, ry, rz; int32_t flag; };   // flag: 1 accepted, 2 prepared rate used; reference locus: accept count << 2
struct alignas(16) GphLrPre {   // prepared proposal of one locus (input order)
  double cand, rspec, lspec, rold, likold;
  uint32_t rx, ry, rz; int32_t slot;
  double pad;
};
struct GphLrArgs {
  double finetune, alpha, dataLnL, logL, rateVar;
  int32_t o_gnd, o_rnd, o_rseq, o_scr, Pscr, o_prog;   // dynamic-LDS byte offsets; loci with P <= Pscr use LDS scratch
  int32_t o_pe, first;                                 // o_prog / o_pe: compiled program and edge probabilities of the reference locus; first = first local locus that proposes
  int32_t o_lf, pad0;                                  // o_lf: the reference locus's leaves as conditional arrays [n][P][4] (0 = no room: generic evaluator)
  // the reference locus as this rank sees it: its node records / scalars (page layout), its sequence block, and the
  // rate / likelihood it has after the loci scanned before this rank's block
  const char *ref_page, *ref_seq;
  int32_t ref_P, ref_seq_bytes;
  double rref0, likref0;
  double *result;            // [0] accepted [1] dataLogLikelihood [2] logLikelihood [3] rateVar [4] error code [5] prepared rates used
  GphLrRec *rec;             // one per slot
  GphLrPre *pre;             // one per locus, input order
  const int32_t *slot_of;    // input-order index -> slot
  double *gscr;              // [n-1][Pmax][4] scratch for loci with P > Pscr
};
#define GPH_LR_SLACK 0.000000001   /* reflect()'s slack, utils.c:335 */

GPH_DEV void kb_lrate_prep(const GphDev &D, int g, double finetune, GphLrPre *pre)
{
  const int go = D.orig[g];
  if (go + D.locus_begin == 0) {          /* the reference locus proposes nothing */
    if (GPH_LANE == 0) { double *o = D.out + (size_t)g * GPH_OUT_SLOTS; o[8] = 0; o[9] = 0; o[10] = 0; o[11] = 0; o[13] = 0; }
    return;
  }
  stage_in(D, g, D.pages, 1);
  GphRng rng;
  rng_load(rng);
  const double rold = FS(FS_MUTRATE), likold = FS(FS_DATALNL);
  const double cand = rold + finetune * l_rnd2normal8(rng);
  /* reflect(cand, 0, b) for every b the proposal does not reach: inside -> cand, at or below the slack -> mirrored */
  const double a = 0.0 + GPH_LR_SLACK;
  const double rspec = UNI(cand > a) ? cand : 2. * a - cand;
  rng_store(rng);
  setFS(FS_MUTRATE, rspec);
  const double lspec = lik_compute(0);
  if (GPH_LANE == 0) {
    GphLrPre r;
    r.cand = cand; r.rspec = rspec; r.lspec = lspec; r.rold = rold; r.likold = likold;
    r.rx = rng.x; r.ry = rng.y; r.rz = rng.z; r.slot = g; r.pad = 0.0;
    pre[go] = r;
  }
  out_common(D, g);
  stage_out(D, g, D.shadow, 2);
}

GPH_DEV void lr_load(const char *pg, const char *seqp, int seqbytes, int o_nd, int o_seq, int &root, double &rate, double &lnl, GphRng &rng)
{
  static_assert(sizeof(GphNode) == 16, "one 16-byte word per node");
#if GPH_BIG_TREE
  gph_copy16_in(GPH_SMB + o_nd, pg + g_lay.o_nd, g_lay.N);
#else
  gph_copy16_in1(GPH_SMB + o_nd, pg + g_lay.o_nd, g_lay.N);
#endif
  copy16_g2l(o_seq, seqp, seqbytes);
  const double *fs = (const double *)(pg + g_lay.o_fscal);
  const int32_t *is = (const int32_t *)(pg + g_lay.o_iscal);
  rate = RFLD(fs[FS_MUTRATE]);
  lnl = RFLD(fs[FS_DATALNL]);
  root = RFL(is[IS_ROOT]);
  rng.x = (uint32_t)RFL(is[IS_RX]); rng.y = (uint32_t)RFL(is[IS_RY]); rng.z = (uint32_t)RFL(is[IS_RZ]);
  GPH_SYNC();
}

#if GPH_LANE_NODES
// ---- the reference locus of the scan: its tree does not change while the scan runs, only its rate does, and it is
// evaluated once per locus by a lone wavefront (at most one instruction every 4 cycles, whatever its type).  The
// pruning order is therefore compiled ONCE per scan into a per-lane table: the internal nodes are sorted by height
// (children strictly lower), nodes of equal height share a step, floor(64 / P) nodes per step, lanes = (node slot,
// pattern).  A step is straight vector code -- table entry, the two edge probabilities, the two children's entries
// (an LDS array or a leaf code), four products, one store -- with no scalar control and no lane reads, and there are
// about half as many steps as nodes.  Same arithmetic per entry as child_factor4 / prune_node_q.
struct GphRefProg {
  int T;          // steps (0 = not compiled: the generic evaluator is used)
  double dt;      // per lane (= node): age(father) - age, the edge above the node
  int father;     // per lane
};
typedef int32_t gph_i4 __attribute__((ext_vector_type(4)));

GPH_DEVHOT void lr_ref_compile(const GphLrArgs &A, int P, GphRefProg &R)
{
  const int n = g_lay.n, N = g_lay.N, lane = GPH_LANE;
  R.T = 0; R.dt = 0.0; R.father = -1;
  if (P < 1 || P > GPH_WAVE || A.o_lf == 0) return;
  const int S = GPH_WAVE / P;
  typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));
  union { gu32x4 v; GphNode nd; } nu;
  GphNode me = {0.0, -1, -1, -1, -1};
  const bool isnode = lane < N, isint = isnode && lane >= n;
  if (isnode) { nu.v = ((GPH_LDS gu32x4 *)(GPH_SMB + A.o_rnd))[lane]; me = nu.nd; }
  const int le = me.left, ri = me.right;
  R.father = me.father;
  if (isnode && me.father >= 0) R.dt = ((lf64 *)(GPH_SMB + A.o_rnd))[2 * me.father] - me.age;
  /* heights */
  int h = isint ? -1 : 0;
  for (int it = 0; it <= N; it++) {
    const int hl = __builtin_amdgcn_ds_bpermute((isint ? le : lane) << 2, h), hr = __builtin_amdgcn_ds_bpermute((isint ? ri : lane) << 2, h);
    if (isint && h < 0 && hl >= 0 && hr >= 0) h = 1 + (hl > hr ? hl : hr);
    if (__ballot(isint && h < 0) == 0) break;
  }
  if (__ballot(isint && h < 0) != 0) { gph_fail(100); return; }
  /* rank of every internal node in (height, index) order, then steps: a new step when the height changes or the
   * slots are used up */
  int rank = 0;
  for (int k = n; k < N; k++) {
    const int hk = __builtin_amdgcn_readlane(h, k);
    rank += (hk < h || (hk == h && k < lane)) ? 1 : 0;
  }
  GPH_LDS int32_t *tmp = (GPH_LDS int32_t *)(GPH_SMB + A.o_pe);      /* N ints fit the N doubles of o_pe */
  if (isint) tmp[rank] = lane;
  GPH_SYNC();
  int mystep = -1, myslot = 0, t = 0, cnt = 0, curh = 0;
  for (int r = 0; r < n - 1; r++) {
    const int node = RFL(tmp[r]);
    const int hh = __builtin_amdgcn_readlane(h, node);
    if (r == 0) curh = hh;
    else if (hh != curh || cnt == S) { t++; cnt = 0; curh = hh; }
    if (lane == node) { mystep = t; myslot = cnt; }
    cnt++;
  }
  const int T = t + 1;
  GPH_SYNC();
  /* the leaves as conditional arrays, the way the reference keeps them (computeLeafConditionals,
   * LocusDataLikelihood.c:1321): one-hot, or all ones for N.  With S = 1 exactly the internal-child formula
   * S*pe + s*qe gives pe + qe / pe, bit for bit what the leaf shortcut of child_factor4 gives; N has S = 4 -> 1.0 */
  const int q_leaf = A.o_rseq + GPH_Q_LEAF;
  for (int i = lane; i < n * P; i += GPH_NLANES) {
    const int leaf = i / P, pat = i - leaf * P;
    const int code = GPH_LEAFCODE(q_leaf, pat, leaf);
    ld2 *o2 = (ld2 *)(GPH_SMB + A.o_lf + i * 32);
    gph_d2 a = {code == 4 || code == 0 ? 1.0 : 0.0, code == 4 || code == 1 ? 1.0 : 0.0};
    gph_d2 b = {code == 4 || code == 2 ? 1.0 : 0.0, code == 4 || code == 3 ? 1.0 : 0.0};
    o2[0] = a;
    o2[1] = b;
  }
  /* the table: T x 64 entries {left, right, own, child ids}; left / right = dynamic-LDS byte offset of the child's four
   * conditionals of this pattern (scratch for an internal child, leaf array for a leaf); own < 0 = idle lane */
  GPH_LDS gph_i4 *prog = (GPH_LDS gph_i4 *)(GPH_SMB + A.o_prog);
  for (int i = lane; i < T * GPH_WAVE; i += GPH_NLANES) { gph_i4 e = {A.o_lf, A.o_lf, -1, 0}; prog[i] = e; }
  GPH_SYNC();
  for (int pat = 0; pat < P; pat++) {
    if (isint) {
      gph_i4 e;
      e.x = le < n ? A.o_lf + (le * P + pat) * 32 : A.o_scr + ((le - n) * P + pat) * 32;
      e.y = ri < n ? A.o_lf + (ri * P + pat) * 32 : A.o_scr + ((ri - n) * P + pat) * 32;
      e.z = A.o_scr + ((lane - n) * P + pat) * 32;
      e.w = le | (ri << 8);
      prog[mystep * GPH_WAVE + myslot * P + pat] = e;
    }
  }
  GPH_SYNC();
  R.T = T;
}

GPH_DEVHOT void lr_ref_factors(int a, double pe, double qe, double &f0, double &f1, double &f2, double &f3)
{
  const ld2 *c2 = (const ld2 *)(GPH_SMB + a);
  const gph_d2 u = c2[0], v = c2[1];
  const double s0 = u.x, s1 = u.y, s2 = v.x, s3 = v.y;
  double S = s0;
  S += s1;
  S += s2;
  S += s3;
  const double Sp = S * pe;
  const bool miss = S >= 4;
  f0 = miss ? 1.0 : (Sp + s0 * qe);
  f1 = miss ? 1.0 : (Sp + s1 * qe);
  f2 = miss ? 1.0 : (Sp + s2 * qe);
  f3 = miss ? 1.0 : (Sp + s3 * qe);
}

// value of the reference locus at `rate` (== lik_private(o_rnd, o_rseq, P, root, rate, ...) bit for bit)
GPH_DEVHOT double lr_ref_eval(const GphLrArgs &A, const GphRefProg &R, int P, double rate)
{
  const int n = g_lay.n, N = g_lay.N, lane = GPH_LANE;
  const int q_phases = A.o_rseq + GPH_Q_PHASES(P, n), q_count = A.o_rseq + GPH_Q_COUNT(P, n);
  rate = RFLD(rate);
#ifdef GPH_LRSTAMP
  const uint64_t st0 = __builtin_readcyclecounter();
#endif
  double pe = 0.0;
  if (lane < N && R.father >= 0) pe = edge_prob_v(rate * R.dt);
  lf64 *pes = (lf64 *)(GPH_SMB + A.o_pe);
  if (lane < N) pes[lane] = pe;
  GPH_WAVE_FENCE();
  const GPH_LDS gph_i4 *prog = (const GPH_LDS gph_i4 *)(GPH_SMB + A.o_prog);
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  gph_i4 e = prog[lane];
#ifdef GPH_LRSTAMP
  e.x = RFL(e.x) * 0 + e.x;
  const uint64_t st1 = __builtin_readcyclecounter();
  gph_lds.s_cntf[1] += (double)(st1 - st0);
#endif
  for (int t = 0; t < R.T; t++) {
    const gph_i4 enext = prog[(t + 1 < R.T ? t + 1 : t) * GPH_WAVE + lane];   /* next step's entry rides under this step */
    const double pl = pes[e.w & 255], pr = pes[(e.w >> 8) & 255];
    const double ql = 1 - 4.0 * pl;
    const double qr = 1 - 4.0 * pr;
    double f0, f1, f2, f3, g0, g1, g2, g3;
    lr_ref_factors(e.x, pl, ql, f0, f1, f2, f3);
    lr_ref_factors(e.y, pr, qr, g0, g1, g2, g3);
    q0 = f0 * g0;
    q1 = f1 * g1;
    q2 = f2 * g2;
    q3 = f3 * g3;
    if (e.z >= 0) {
      ld2 *o2 = (ld2 *)(GPH_SMB + e.z);
      gph_d2 a = {q0, q1}, b = {q2, q3};
      o2[0] = a;
      o2[1] = b;
    }
    GPH_WAVE_FENCE();
    e = enext;
  }
#ifdef GPH_LRSTAMP
  q0 = RFLD(q0) * 0.0 + q0;
  const uint64_t st2 = __builtin_readcyclecounter();
  gph_lds.s_cntf[2] += (double)(st2 - st1);
  gph_lds.s_cntf[5] += (double)R.T;
#endif
  /* the root is alone in the last step, slot 0: lane p < P holds its four conditionals of pattern p.
   * Root reduction as in lik_compute / lik_private_t (LocusDataLikelihood.c:466-479) */
  double lnl = 0.0, term = 0.0;
  const int ph = lane < P ? gu16v(q_phases, lane) : 0;
  double prob = q0;
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
#ifdef GPH_LRSTAMP
  lnl = RFLD(lnl);
  gph_lds.s_cntf[3] += (double)(__builtin_readcyclecounter() - st2);
  gph_lds.s_cntf[4] += 1.0;
#endif
  return lnl;
}
#endif

GPH_DEV void kb_lrate_scan(const GphDev &D, const GphLrArgs &A)
{
  int k, Pr, rootr, P, root, accepted = 0, hits = 0;
  double rref, likref, dummy_r, dummy_l;
  GphRng rng, rngr;
  for (k = 0; k < CN_COUNT; k++) setCNT(k, 0);
#ifdef GPH_LRSTAMP
  for (k = 0; k < 8; k++) gph_lds.s_cntf[k] = 0.0;
#endif
#ifndef GPH_HOSTEMU
  const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  Pr = A.ref_P;
  lr_load(A.ref_page, A.ref_seq, A.ref_seq_bytes, A.o_rnd, A.o_rseq, rootr, dummy_r, dummy_l, rngr);
  rref = A.rref0;
  likref = A.likref0;
  double dataLnL = A.dataLnL, logL = A.logL, rateVar = A.rateVar;
  const double Ld = (double)D.Ltot;
  gdbl *gs = (gdbl *)A.gscr;
#if GPH_LANE_NODES
  GphRefProg RP;
  lr_ref_compile(A, Pr, RP);
#endif
  for (int base = A.first; base < D.L; base += GPH_NLANES) {
    /* one batch of prepared proposals: lane i holds locus base + i */
    const int mine = base + GPH_LANE < D.L ? base + GPH_LANE : D.L - 1;
    const GphLrPre pm = A.pre[mine];
    const int nb = D.L - base < GPH_NLANES ? D.L - base : GPH_NLANES;
    for (int i = 0; i < nb; i++) {
      const double cand = GPH_LANEVAL64(pm.cand, i), rspec = GPH_LANEVAL64(pm.rspec, i), lspec = GPH_LANEVAL64(pm.lspec, i),
                   rold = GPH_LANEVAL64(pm.rold, i), likold = GPH_LANEVAL64(pm.likold, i);
      const int j = GPH_LANEVAL32(pm.slot, i);
      rng.x = (uint32_t)GPH_LANEVAL32(pm.rx, i);
      rng.y = (uint32_t)GPH_LANEVAL32(pm.ry, i);
      rng.z = (uint32_t)GPH_LANEVAL32(pm.rz, i);
      const double rnew = l_reflect(cand, 0, rold + rref);
      const double rrefnew = rref + rold - rnew;
      /* Dirichlet(alpha) prior ratio; with alpha = 1 and both rates positive and finite the term is a signed zero that
       * cannot change the decision or any accumulator: the logarithm is skipped */
      const double prat = (rnew * rrefnew) / (rold * rref);
      double lnacc = 0.0;
      if (!(A.alpha == 1.0 && UNI(prat > 0 && prat < 1e300))) lnacc = (A.alpha - 1) * gph_log(prat);
      double lnLd = -(likold + likref);
      const bool hit = UNI(rnew == rspec);
      double lg;
      if (hit) {
        lg = lspec;
        hits++;
      } else {
        GphRng unused_rng;
        GPH_SYNC();
        P = RFL(D.P[2 * j]);
        lr_load(D.pages + (size_t)j * g_lay.page_bytes, D.seq + D.seq_off[j], (int)(D.seq_off[j + 1] - D.seq_off[j]), A.o_gnd, 0, root,
                dummy_r, dummy_l, unused_rng);
        lg = lik_private(A.o_gnd, 0, P, root, rnew, A.o_scr, P > A.Pscr ? gs : (gdbl *)0);
      }
      lnLd += lg;
#if GPH_LANE_NODES
      const double lr = RP.T > 0 && Pr <= A.Pscr ? lr_ref_eval(A, RP, Pr, rrefnew)
                                                  : lik_private(A.o_rnd, A.o_rseq, Pr, rootr, rrefnew, A.o_scr, Pr > A.Pscr ? gs : (gdbl *)0);
#else
      const double lr = lik_private(A.o_rnd, A.o_rseq, Pr, rootr, rrefnew, A.o_scr, Pr > A.Pscr ? gs : (gdbl *)0);
#endif
      lnLd += lr;
      lnacc += lnLd;
      bool acc = UNI(lnacc >= 0);
      if (!acc) acc = UNI(l_rndu(rng) < gph_exp(lnacc));
      if (acc) {
        accepted++;
        dataLnL += lnLd;
        logL += lnLd / Ld;
        rateVar += (rnew * rnew + rrefnew * rrefnew - rold * rold - rref * rref) / Ld;
        rref = rrefnew;
        likref = lr;
      }
      if (GPH_LANE == 0) {
        GphLrRec r;
        r.rate = acc ? rnew : rold; r.lnl = acc ? lg : likold;
        r.rx = rng.x; r.ry = rng.y; r.rz = rng.z; r.flag = (acc ? 1 : 0) | (hit ? 2 : 0);
        A.rec[j] = r;
      }
      if (gph_failed()) break;
    }
    if (gph_failed()) break;
  }
  if (GPH_LANE == 0) {
    /* the reference locus's own record is written by the host once every rank's block has been scanned */
    A.result[13] = rref; A.result[14] = likref;
    A.result[0] = accepted; A.result[1] = dataLnL; A.result[2] = logL; A.result[3] = rateVar; A.result[4] = gph_errcode();
    A.result[5] = hits;
#ifndef GPH_HOSTEMU
    A.result[6] = (double)(__builtin_readcyclecounter() - clk0);            /* shader-clock cycles of the scan */
    A.result[7] = (double)(__builtin_amdgcn_s_memrealtime() - rt0);         /* 100 MHz reference ticks */
#ifdef GPH_LRSTAMP
    A.result[8] = gph_lds.s_cntf[1]; A.result[9] = gph_lds.s_cntf[2]; A.result[10] = gph_lds.s_cntf[3]; A.result[11] = gph_lds.s_cntf[4]; A.result[12] = gph_lds.s_cntf[5];
#endif
#endif
  }
}

GPH_DEV void kb_lrate_apply(const GphDev &D, int g, const GphLrRec *rec)
{
  const GphLrRec r = rec[g];
  const bool isref = (D.orig[g] + D.locus_begin) == 0;
  const int flag = RFL(r.flag);
  const int accepts = isref ? flag >> 2 : flag & 1;
  if (accepts == 0) {
    /* rejected (or an untouched reference locus): only the stream moved on */
    if (GPH_LANE == 0) {
      int32_t *is = (int32_t *)(D.pages + (size_t)g * g_lay.page_bytes + g_lay.o_iscal);
      double *o = D.out + (size_t)g * GPH_OUT_SLOTS;
      is[IS_RX] = (int32_t)r.rx; is[IS_RY] = (int32_t)r.ry; is[IS_RZ] = (int32_t)r.rz;
      o[8] = 0; o[9] = 0; o[10] = 0; o[11] = 0; o[13] = 0;
    }
    return;
  }
  if (!isref && (flag & 2)) {
    /* accepted at the prepared rate: the evaluated state is in the shadow page */
    stage_in(D, g, D.shadow, 0);
    lik_reset_saved();
  } else {
    stage_in(D, g, D.pages, 1);
    setFS(FS_MUTRATE, RFLD(r.rate));
    const int reps = isref ? 2 - (accepts & 1) : 1;
    for (int k = 0; k < reps; k++) {
      lik_compute(0);
      lik_reset_saved();
    }
  }
  setISC(IS_RX, (int)r.rx); setISC(IS_RY, (int)r.ry); setISC(IS_RZ, (int)r.rz);
  if (!UNI(FS(FS_DATALNL) == RFLD(r.lnl)) || !UNI(FS(FS_MUTRATE) == RFLD(r.rate))) gph_fail(120);   /* scan and in-place evaluation agree bit for bit */
  out_common(D, g);
  stage_out(D, g, D.pages, 1);
}

// ---------------------------------------------------------------- end-of-iteration
// synchronizeEvents(gen), GPhoCS.c:1705-1714; refresh = start-mig recomputation of
// genLogLikelihood (GPhoCS.c:1749-1756).  out: 0 ok, 1 old genLnL, 2 new genLnL
GPH_DEV void kb_sync(const GphDev &D, int g, int refresh)
{
  int ok;
  stage_in(D, g, D.pages, 0);
  ok = synchronize_events();
  OUT(g, 0, ok);
  OUT(g, 1, FS(FS_GENLNL));
  if (refresh) setFS(FS_GENLNL, gtree_lnl());
  OUT(g, 2, FS(FS_GENLNL));
  out_common(D, g);
  stage_out(D, g, D.pages, 0);
}

// checkAll per-locus body, patch.c:2766-2790: structure check + statistics resync,
// full likelihood recomputation (checkLocusDataLikelihood, LocusDataLikelihood.c:717),
// genLogLikelihood resync.  out: 0 ok, 1 dataLnL, 2 genLnL
GPH_DEV void kb_check(const GphDev &D, int g)
{
  int ok;
  double lnLd_gen, PREC = 0.0000001, a, b;
  stage_in(D, g, D.pages, 1);
  ok = check_gtree_structure();
  lik_compute(0);
  a = FS(FS_DATALNL);
  b = FS(FS_SV_DATALNL);
  if (!(a == b || fabs(1 - a / b) < 0.000000001)) ok = 0;
  lik_reset_saved();
  lnLd_gen = gtree_lnl();
  if (fabs(FS(FS_GENLNL) - lnLd_gen) > PREC && fabs(1 - FS(FS_GENLNL) / lnLd_gen) > PREC) ok = 0;
  setFS(FS_GENLNL, lnLd_gen);
  OUT(g, 0, ok);
  OUT(g, 1, FS(FS_DATALNL));
  OUT(g, 2, lnLd_gen);
  out_common(D, g);
  stage_out(D, g, D.pages, 1);
}
// ---------------------------------------------------------------- kernel-level fixtures
// Single calls of the per-locus functions with deterministic arguments, every one undone afterwards (the locus is not
// written back): what oracle/ref_harness.c `unit` does with the real reference's functions on the same chain state
// (tests/golden/*.unit).  A parity break shows up as ONE differing line.
//   op 0  per internal node: adjustGenNodeAge + computeLocusDataLikelihood(useOld=1) + considerEventMove
//         (body of UpdateGB_InternalNode, GPhoCS.c:2316-2381) -> uo[3 (inode - n) + {0,1,2}] = tnew, lnLd, dprior
//   op 1  computeLocusDataLikelihood(useOld=0) -> uo[0]
//   op 2  rubberBand(pre) x3 of the ancestral population in the chain state's pending proposal + evaluation
//         (UpdateTau loop 1 without the ripple, GPhoCS.c:3705-3831) -> uo[0..3] = delta, n0, n1, lik
GPH_DEV void kb_unit(const GphDev &D, int g, int op, int arg, double *out, int stride)
{
  double *uo = out + (size_t)D.orig[g] * stride;
  stage_in(D, g, D.pages, 1);
  if (op == 0) {
    for (int inode = g_lay.n; inode < g_lay.N; inode++) {
      const GphNodeS me = ld_node(inode);
      const double t = me.age;
      const int pop = me.npop;
      double tb0 = g_model.popAge[pop], tb1;
      if (pop != g_lay.rootPop) tb1 = g_model.popAge[g_model.popFather[pop]];
      else tb1 = GPH_OLDAGE;
      int mig = find_first_mig(inode, -1);
      if (mig >= 0) tb1 = gmin2(tb1, MAGE(mig));
      else if (inode != ISC(IS_ROOT)) tb1 = gmin2(tb1, AGE(me.father));
      for (int i = 0; i < 2; i++) {
        const int son = i == 0 ? me.left : me.right;
        mig = find_last_mig(son, -1);
        if (mig >= 0) tb0 = gmax2(tb0, MAGE(mig));
        else tb0 = gmax2(tb0, AGE(son));
      }
      const double hi = gmin2(tb1, t * 1.5 + 1e-7);
      const double tnew = tb0 + 0.61803 * (hi - tb0);
      lik_adjust_age(inode, tnew);
      double lnLd = -FS(FS_DATALNL);
      lnLd += lik_compute(1);
      const double dprior = consider_event_move(0, NEV(inode), pop, t, pop, tnew);
      reject_event_chain_changes(0);
      lik_revert();
      if (GPH_LANE == 0) { uo[3 * (inode - g_lay.n)] = tnew; uo[3 * (inode - g_lay.n) + 1] = lnLd; uo[3 * (inode - g_lay.n) + 2] = dprior; }
    }
  } else if (op == 1) {
    const double v = lik_compute(0);
    lik_reset_saved();
    if (GPH_LANE == 0) uo[0] = v;
  } else if (op == 2) {
    gph_ctau &A = GPH_G->tau;
    int n0 = 0, n1 = 0;
    double d, lik = 0.0;
    if (A.isRoot) d = rubber_band(A.ap, g_model.popAge[A.ap], A.taub0, A.tauold, A.taufactor1, 0, &n1);
    else d = rubber_band(A.ap, g_model.popAge[A.ap], A.taub1, A.tauold, A.taufactor1, 0, &n1);
    d += rubber_band(A.son0, g_model.popAge[A.son0], A.taub0, A.tauold, A.taufactor0, 0, &n0);
    d += rubber_band(A.son1, g_model.popAge[A.son1], A.taub0, A.tauold, A.taufactor0, 0, &n0);
    if (n0 + n1) { lik = -FS(FS_DATALNL); lik += lik_compute(1); }
    lik_revert();
    if (GPH_LANE == 0) { uo[0] = d; uo[1] = n0; uo[2] = n1; uo[3] = lik; }
  } else if (op == 3) {
    /* executeGenSPR of node `arg` onto its father's, its sibling's, the root's and every fifth other legal branch, at a
     * fixed point of the legal age window, + computeLocusDataLikelihood(1), + revertToSaved (oracle/ref_harness.c: unit2 D):
     * uo[0] = calls, then target, age, return code, value, root after the call */
    const int node = arg, root = ISC(IS_ROOT);
    int cnt = 0;
    if (node != root) {
      const int father = FATH(node), sibling = LEFT(father) + RGHT(father) - node, grandpa = FATH(father);
      for (int target = 0; target < g_lay.N; target++) {
        if (target == node) continue;
        { int x = target, guard = 0, under = 0;
          while (x >= 0 && guard++ < g_lay.N) { if (x == node) { under = 1; break; } x = FATH(x); }
          if (under) continue; }
        if (!(target == father || target == sibling || target == root || (target + node) % 5 == 0)) continue;
        double lo, hi;
        if (target == father || target == sibling) {
          lo = gmax2(AGE(node), AGE(sibling));
          hi = grandpa >= 0 ? AGE(grandpa) : lo * 1.3 + 1e-6;
        } else {
          const int tf = FATH(target);
          lo = gmax2(AGE(node), AGE(target));
          hi = tf >= 0 ? AGE(tf) : lo * 1.3 + 1e-6;
        }
        if (!UNI(hi > lo)) continue;
        const double age = lo + 0.37 * (hi - lo);
        const int ret = lik_spr(node, target, age);
        const double lnl = lik_compute(1);
        const int nroot = ISC(IS_ROOT);
        lik_revert();
        if (GPH_LANE == 0 && 1 + 5 * (cnt + 1) <= stride) {
          double *r = uo + 1 + 5 * cnt;
          r[0] = target; r[1] = age; r[2] = ret; r[3] = lnl; r[4] = nroot;
        }
        cnt++;
      }
    }
    if (GPH_LANE == 0) uo[0] = cnt;
  } else if (op == 4) {
    /* scaleAllNodeAges(1 + arg / 1000), revertToSaved, full recompute (unit2 E) */
    const double factor = 1.0 + arg * 0.001;
    const double d = lik_scale_ages(factor);
    lik_revert();
    const double v2 = lik_compute(0);
    lik_reset_saved();
    if (GPH_LANE == 0) { uo[0] = d; uo[1] = v2; }
  } else if (op == 5) {
    /* rubberBandRipple(do) + rubberBandRipple(undo) over the source-side events of every migration event, 0.01 % older
     * (unit2 F) */
    int nm = 0;
    for (int i = 0; i < ISC(IS_NUM_MIGS); i++) {
      const int mig = LIVING(i), pop = MG(mig, MG_SPOP);
      const double na = MAGE(mig) * 1.0001;
      const double top = pop == g_lay.rootPop ? GPH_OLDAGE : g_model.popAge[g_model.popFather[pop]];
      if (!UNI(na < top)) continue;
      setRBI(0, nm, MG(mig, MG_SEV));
      setRBI(2, nm, pop);
      sf64(&GphLds::rb_age, nm, na);
      nm++;
    }
    setISC(IS_RB_NUM, nm);
    const double d1 = rubber_band_ripple(1);
    const double d0 = rubber_band_ripple(0);
    if (GPH_LANE == 0) { uo[0] = nm; uo[1] = d1; uo[2] = d0; }
  } else if (op == 8) {
    /* self-test of the checked build (-DGPH_BOUNDS): an index one past the node records' capacity must be caught (and redirected
     * to element 0); in a product build this op does nothing */
#ifdef GPH_BOUNDS
    const double x = AGE(arg);
    if (GPH_LANE == 0) uo[0] = x;
#endif
  } else if (op == 7) {
    /* rubberBandRipple(do / undo) over migration-band events (unit2 H): per band the MIG_BAND_START event of the target
     * population's chain 30 % into the gap to its successor, the MIG_BAND_END event 30 % into the gap to its predecessor */
    int nm = 0;
    for (int b = 0; b < g_lay.B; b++) {
      const int tp = g_model.bandTgt[b];
      double age = g_model.popAge[tp];
      int guard = 0;
      for (int ev = FIRSTEV(tp); ev >= 0 && guard++ <= GPH_CAP_E;) {
        const GphEvS R = ld_ev(ev);
        age += R.time;
        if (R.node == b && nm < GPH_CAP_RB) {
          if (R.type == GPH_MIG_BAND_START && R.next >= 0 && UNI(ld_ev(R.next).time > 0.0)) {
            setRBI(0, nm, ev); setRBI(2, nm, tp);
            sf64(&GphLds::rb_age, nm, age + 0.3 * ld_ev(R.next).time);
            nm++;
          } else if (R.type == GPH_MIG_BAND_END && UNI(R.time > 0.0)) {
            setRBI(0, nm, ev); setRBI(2, nm, tp);
            sf64(&GphLds::rb_age, nm, age - 0.3 * R.time);
            nm++;
          }
        }
        ev = R.next;
      }
    }
    setISC(IS_RB_NUM, nm);
    const double d1 = rubber_band_ripple(1);
    const double d0 = rubber_band_ripple(0);
    if (GPH_LANE == 0) { uo[0] = nm; uo[1] = d1; uo[2] = d0; }
  } else {
    /* op 6: traceLineage(arg, 0) + traceLineage(arg, 1) as UpdateGB_MigSPR calls them + the evaluation (unit2 G); nothing is
     * undone: this kernel does not write the page back */
    const int node = arg;
    if (node != ISC(IS_ROOT)) {
      const int father = FATH(node);
      GphRng rng;
      rng_load(rng);
#if GPH_BIG_BANDS || defined(GPH_TWO_WALKS)
      trace_lineage<0>(node, rng);
      const int res = trace_lineage<1>(node, rng);
#else
      const int res = trace_pair(node, rng);
#endif
      double lnl = -FS(FS_DATALNL);
      lnl += lik_compute(1);
      rng_store(rng);
      if (GPH_LANE == 0) {
        uo[0] = 1; uo[1] = res; uo[2] = res >= 0 ? SPRI(SI_TARGET) : -1; uo[3] = res >= 0 ? SPRI(SI_FPOP_NEW) : -1;
        uo[4] = SPRI(SI_NOLD); uo[5] = SPRI(SI_NNEW); uo[6] = SPRLN(0); uo[7] = SPRLN(1); uo[8] = AGE(father); uo[9] = lnl;
        uo[10] = (double)(uint32_t)ISC(IS_RX); uo[11] = (double)(uint32_t)ISC(IS_RY); uo[12] = (double)(uint32_t)ISC(IS_RZ);
      }
    } else if (GPH_LANE == 0) uo[0] = 0;
  }
  out_common(D, g);
}
};   // struct GphCtxT
using GphCtx = GphCtxT<false>;    // model in the kernel-argument segment (genealogy sweep, locus-rate kernels)
using GphCtxG = GphCtxT<true>;    // model in the device-resident chain state
using GphLrRec = GphCtx::GphLrRec;
using GphLrPre = GphCtx::GphLrPre;
using GphLrArgs = GphCtx::GphLrArgs;
