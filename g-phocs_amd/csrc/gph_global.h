// gph_global.h -- the part of performMCMC's iteration body that runs ABOVE the loci (upstream src/GPhoCS.c:
// UpdateTheta :3037-3107, UpdateMigRates :3115-3213, the host parts of UpdateTau :3224-3461 / :3835-3994,
// UpdateSampleAge :4006-4128 / :4447-4584 and mixing :4688-4789 / :4803-4912, the accumulators of
// performMCMC :1495-1757, checkAll's accumulator resynchronisation patch.c:2788-2875), written ONCE for both
// places it runs:
//   * on the device, as the body of k_global (one wavefront, every lane executes it uniformly): the decision of a
//     global proposal is taken straight from the reduced (and, over several GPUs, all-gathered) vectors in HBM and
//     left as a flag for the commit / revert kernel -- no host round trip inside an iteration;
//   * on the host, when a caller-supplied all-reduce hook forces a synchronisation at every reduction point anyway
//     (gloo tests, UpdateLocusRate's chained scan), and in the host-emulation build of the tests.
// Same expressions, same operation order as the reference (file compiled with -ffp-contract=off); exp / log are
// gph_math.h's (bit-identical to the glibc the reference links, on both sides).
#pragma once
#include <math.h>
#include <stdint.h>
#include "gph_types.h"
#include "gph_math.h"

#ifdef __HIPCC__
#define GPH_HD __host__ __device__ inline
#define GPH_HDM __host__ __device__ inline
#else
#define GPH_HD static inline
#define GPH_HDM inline
#endif

#define GG_OLDAGE 999.0
#define gg_max2(a, b) ((a) > (b) ? (a) : (b))
#define gg_min2(a, b) ((a) < (b) ? (a) : (b))

// the reduced vectors of the last launch as every rank sees them: `world` rows of GPH_RED_ROW doubles, combined in
// rank order (the same additions on every rank => identical decisions everywhere)
struct GphRed {
  const double *rows;
  int world;
  GPH_HDM double sum(int sec, int col) const
  {
    double s = rows[sec * GPH_RED_STRIDE + col];
    for (int r = 1; r < world; r++) s += rows[(size_t)r * GPH_RED_ROW + sec * GPH_RED_STRIDE + col];
    return s;
  }
  GPH_HDM double mn(int sec, int col) const
  {
    double s = rows[sec * GPH_RED_STRIDE + GPH_RED_COLS + col];
    for (int r = 1; r < world; r++) { const double v = rows[(size_t)r * GPH_RED_ROW + sec * GPH_RED_STRIDE + GPH_RED_COLS + col]; s = v < s ? v : s; }
    return s;
  }
  GPH_HDM double mx(int sec, int col) const
  {
    double s = rows[sec * GPH_RED_STRIDE + 2 * GPH_RED_COLS + col];
    for (int r = 1; r < world; r++) { const double v = rows[(size_t)r * GPH_RED_ROW + sec * GPH_RED_STRIDE + 2 * GPH_RED_COLS + col]; s = v > s ? v : s; }
    return s;
  }
  GPH_HDM double errword() const
  {
    double s = rows[3 * GPH_RED_COLS];
    for (int r = 1; r < world; r++) { const double v = rows[(size_t)r * GPH_RED_ROW + 3 * GPH_RED_COLS]; s = v > s ? v : s; }
    return s;
  }
};

// rndu / rndnormal / rnd2normal8 for the general slot: utils.c:459-513
GPH_HD double gg_rndu(GphGlobal &G)
{
  double r;
  G.gx = 171u * (G.gx % 177u) - 2u * (G.gx / 177u);
  G.gy = 172u * (G.gy % 176u) - 35u * (G.gy / 176u);
  G.gz = 170u * (G.gz % 178u) - 63u * (G.gz / 178u);
#if defined(__HIP_DEVICE_COMPILE__)
  /* the stage runs on ONE lane of a lone wavefront: instruction count is its time.  IEEE-exact quotients without the
   * divide expansion, as in the locus streams (gph_locus.h: l_rndu; exhaustively verified for all 32-bit x) */
  {
    const double rx = 1.0 / 30269.0, ry = 1.0 / 30307.0, rz = 1.0 / 30323.0;
    const double xd = (double)G.gx, yd = (double)G.gy, zd = (double)G.gz;
    double q;
    q = xd * rx; const double qx = __builtin_fma(__builtin_fma(-q, 30269.0, xd), rx, q);
    q = yd * ry; const double qy = __builtin_fma(__builtin_fma(-q, 30307.0, yd), ry, q);
    q = zd * rz; const double qz = __builtin_fma(__builtin_fma(-q, 30323.0, zd), rz, q);
    r = qx + qy + qz;
  }
#else
  r = G.gx / 30269.0 + G.gy / 30307.0 + G.gz / 30323.0;
#endif
  r = (r - (int)r);
  return r;
}
GPH_HD double gg_rndnormal(GphGlobal &G)
{
  double u, v, s;
  int guard = 0;
  for (;;) {
    u = 2 * gg_rndu(G) - 1;
    v = 2 * gg_rndu(G) - 1;
    s = u * u + v * v;
    if (s > 0 && s < 1) break;
    if (++guard > 100000) { if (!G.error) { G.error = 89; G.error_locus = -1; } return 0.0; }
  }
  s = sqrt(-2. * gph_log(s) / s);
  return u * s;
}
GPH_HD double gg_rnd2normal8(GphGlobal &G)
{
  const double m2s2 = 8.;
  double m2N = sqrt(m2s2 / (m2s2 + 1.));
  double s2N = sqrt(1. / (m2s2 + 1.));
  double z = m2N + gg_rndnormal(G) * s2N;
  z = gg_rndu(G) < 0.5 ? z : -z;
  return z;
}
// reflect, utils.c:333-398
GPH_HD double gg_reflect(double x, double a, double b)
{
  const double slack = 0.000000001;
  double xnew, di;
  int guard = 0;
  a += slack;
  b -= slack;
  if (b <= a) return (a + b) / 2.;
  if (x < b && x > a) return x;
  xnew = x;
  if (xnew <= a) xnew = 2. * a - xnew;
  di = 2. * (b - a);
  xnew = xnew - di * floor((xnew - a) / di);
  if (xnew >= b) xnew = 2. * b - xnew;
  while (xnew <= a || xnew >= b) {
    if (xnew >= b) xnew = 2. * b - xnew;
    else xnew = 2 * a - xnew;
    if (++guard > 64) return (a + b) / 2.;   /* the reference would spin forever here */
  }
  return xnew;
}

// updateMigrationBandTimes / computeMigrationBandTimes, PopulationTree.c:439-491
GPH_HD int gg_update_band_times(GphGlobal &G, int b)
{
  GphModel &M = G.model;
  int res = 0, src = M.bandSrc[b], tgt = M.bandTgt[b];
  double t = gg_max2(M.popAge[src], M.popAge[tgt]);
  if (t != M.bandStart[b]) { M.bandStart[b] = t; res = 1; }
  t = gg_min2(M.popAge[M.popFather[src]], M.popAge[M.popFather[tgt]]);
  if (t != M.bandEnd[b]) { M.bandEnd[b] = t; res = 1; }
  return res;
}
GPH_HD void gg_compute_band_times(GphGlobal &G)
{
  GphModel &M = G.model;
  for (int b = 0; b < G.B; b++) {
    gg_update_band_times(G, b);
    if (M.bandStart[b] >= M.bandEnd[b]) M.bandStart[b] = M.bandEnd[b] = M.popAge[M.bandTgt[b]];
  }
}
GPH_HD void gg_set_theta(GphGlobal &G, int pop, double v)
{
  G.model.theta[pop] = v; G.model.thetaInv[pop] = 1.0 / v; G.model.logTwoTheta[pop] = gph_log(2 / v);
}
GPH_HD void gg_set_mig(GphGlobal &G, int b, double v) { G.model.migRate[b] = v; G.model.logMigRate[b] = gph_log(v); }

GPH_HD void gg_rec(GphGlobal &G, int code, int idx, long long acc)
{
  if (G.nrec >= GPH_REC_MAX) return;
  GphRec &r = G.rec[G.nrec++];
  r.code = code; r.idx = idx; r.acc = acc; r.dataLnL = G.dataLogLikelihood; r.logL = G.logLikelihood;
}

// samplePopParameters, PopulationTree.c:339-403
GPH_HD void gg_sample_pop_parameters(GphGlobal &G)
{
  GphModel &M = G.model;
  int queue[GPH_MAXK];
  int head = 0, tail = 0, pop;
  double mean;
  queue[tail++] = G.rootPop;
  while (head < tail) {
    pop = queue[head++];
    mean = G.thetaStart[pop];
    gg_set_theta(G, pop, mean * (0.9 + 0.2 * gg_rndu(G)));
    if (M.popSon0[pop] >= 0) {
      mean = G.ageStart[pop];
      M.popAge[pop] = mean * (0.9 + 0.2 * gg_rndu(G));
      if (M.popFather[pop] >= 0 && M.popAge[M.popFather[pop]] < M.popAge[pop]) {
        M.popAge[pop] = gg_max2(M.sampleAge[M.popSon0[pop]], M.sampleAge[M.popSon1[pop]]);
        M.popAge[pop] += (M.popAge[M.popFather[pop]] - M.popAge[pop]) * (0.93 + 0.004 * gg_rndu(G));
      }
      queue[tail++] = M.popSon0[pop];
      queue[tail++] = M.popSon1[pop];
    }
  }
  for (int b = 0; b < G.B; b++) gg_set_mig(G, b, 0.0);
  gg_compute_band_times(G);
}

// counters of the launch whose reduction is in R, and its error words
GPH_HD void gg_count(GphGlobal &G, const GphRed &R, int which)
{
  G.cls_evals[which] += R.sum(0, 8);
  G.cls_nodes[which] += R.sum(0, 9);
  G.cls_bytes[which] += R.sum(0, 10);
  G.cnt_notenough += R.sum(0, 13);
  if (!G.error) {
    const double e1 = R.mx(0, 11), e2 = R.errword();
    if (e1 != 0.0) {       /* out_common: (2^30 - first failing locus) 2^14 + its code */
      const long long w = (long long)e1;
      G.error = (int32_t)(w & 16383);
      G.error_locus = ((long long)1 << 30) - (w >> 14);
    } else if (e2 != 0.0) { G.error = (int32_t)e2; G.error_locus = -1; }
  }
}
GPH_HD void gg_totals(GphGlobal &G, const GphRed &R)
{
  const int K = G.K, B = G.B;
  for (int p = 0; p < K; p++) { G.tot_coal[p] = R.sum(1, p); G.tot_ncoal[p] = R.sum(1, K + p); }
  for (int b = 0; b < B; b++) { G.tot_mig[b] = R.sum(1, 2 * K + b); G.tot_nmig[b] = R.sum(1, 2 * K + B + b); }
}

// accumulators after the fused genealogy sweep, GPhoCS.c:1495-1545
GPH_HD void gg_sweep_done(GphGlobal &G, const GphRed &R, int with_sync)
{
  G.nrec = 0;          /* first stage of an iteration */
  G.shownValid = 0;
  gg_count(G, R, 0);
  if (with_sync && R.mn(0, 15) < 1.0 && !G.error) { G.error = 75; G.error_locus = -1; }   /* synchronizeEvents, Fatal Error 0075/0076 */
  G.dataLogLikelihood += R.sum(0, 3);
  G.logLikelihood += R.sum(0, 4);
  G.acc[0] += (int64_t)R.sum(0, 0);
  gg_rec(G, REC_INT, 0, (long long)R.sum(0, 0));
  G.logLikelihood += R.sum(0, 5);
  G.acc[1] += (int64_t)R.sum(0, 1);
  G.acc[7] += (int64_t)R.sum(0, 12);
  gg_rec(G, REC_MIGN, 0, (long long)R.sum(0, 1));
  G.dataLogLikelihood += R.sum(0, 6);
  G.logLikelihood += R.sum(0, 7);
  G.acc[2] += (int64_t)R.sum(0, 2);
  gg_rec(G, REC_SPR, 0, (long long)R.sum(0, 2));
  gg_totals(G, R);
}

// UpdateTheta, GPhoCS.c:3037-3107 and UpdateMigRates, :3115-3213 (the latter only once iteration > start-mig,
// :1596): decisions from the statistics totals; the per-locus genLogLikelihood touch-ups (:3084-3093, :3192-3200)
// are queued in G.apply and applied to every locus in this order by k_apply_list
GPH_HD void gg_update_theta(GphGlobal &G)
{
  GphModel &M = G.model;
  int accepted = 0;
  if (G.ftTheta > 0.0) {
    for (int pop = 0; pop < G.K; pop++) {
      double thetaold = M.theta[pop];
      double lnc = G.ftTheta * gg_rnd2normal8(G);
      double c = gph_exp(lnc);
      double thetanew = thetaold * c;
      double lnacc = lnc + lnc * (G.thetaAlpha[pop] - 1) - (thetanew - thetaold) * G.thetaBeta[pop];
      double dLL = -(lnc * G.tot_ncoal[pop] + (1 / thetanew - 1 / thetaold) * G.tot_coal[pop]);
      lnacc += dLL;
      if (lnacc >= 0 || gg_rndu(G) < gph_exp(lnacc)) {
        accepted++;
        GphApply &a = G.apply[G.napply++];
        a.kind = 0; a.idx = pop; a.lnc = lnc; a.diff = (1 / thetanew - 1 / thetaold);
        G.logLikelihood += dLL / G.Ltot;
        gg_set_theta(G, pop, thetanew);
      }
    }
  }
  G.acc[3] += accepted;
  gg_rec(G, REC_THETA, 0, accepted);
}
GPH_HD void gg_update_mig_rates(GphGlobal &G)
{
  GphModel &M = G.model;
  int accepted = 0;
  if (G.ftMigRate > 0.0) {
    for (int b = 0; b < G.B; b++) {
      double old_rate = M.migRate[b];
      double lnc = G.ftMigRate * gg_rnd2normal8(G);
      double c = gph_exp(lnc);
      double new_rate = old_rate * c;
      if (new_rate < 0.00001) continue;
      double lnacc = lnc + lnc * (G.mrAlpha[b] - 1) - (new_rate - old_rate) * G.mrBeta[b];
      double dLL = (lnc * G.tot_nmig[b] - (new_rate - old_rate) * G.tot_mig[b]);
      lnacc += dLL;
      if (lnacc >= 0 || gg_rndu(G) < gph_exp(lnacc)) {
        accepted++;
        GphApply &a = G.apply[G.napply++];
        a.kind = 1; a.idx = b; a.lnc = lnc; a.diff = (new_rate - old_rate);
        gg_set_mig(G, b, new_rate);
        G.logLikelihood += dLL / G.Ltot;
      }
    }
  }
  G.acc[4] += accepted;
  gg_rec(G, REC_MIGR, 0, accepted);
}
GPH_HD void gg_theta_and_mig_rates(GphGlobal &G)
{
  G.napply = 0;
  gg_update_theta(G);
  if (G.iteration > G.startMig) gg_update_mig_rates(G);
}

// the model change of an accepted UpdateTau / UpdateSampleAge proposal: upstream assigns it after its commit loop
// (GPhoCS.c:3946, :4541), and the commit kernel reads the OLD age -- so it is applied by the stage that follows the
// finish kernel
GPH_HD void gg_apply_pending(GphGlobal &G)
{
  if (G.pend_kind == 1) G.model.popAge[G.pend_pop] = G.pend_taunew;
  else if (G.pend_kind == 2) G.model.sampleAge[G.pend_pop] = G.pend_taunew;
  G.pend_kind = 0;
}

// UpdateTau, GPhoCS.c:3224-3461: bounds, proposal, affected bands, prior ratio of one ancestral population
GPH_HD void gg_tau_propose(GphGlobal &G, int ap)
{
  GphModel &M = G.model;
  GphTauArgs &A = G.tau;
  int sons[2], isRoot, k, b, src, tgt, res, num_aff = 0;
  double tauold, taunew, taub[2], taufactor[2];
  gg_apply_pending(G);
  A.ap = 0; A.son0 = 0; A.son1 = 0; A.isRoot = 0; A.num_aff = 0; A.mode = 0;
  G.accArr[ap] = 0;
  isRoot = (ap == G.rootPop);
  tauold = M.popAge[ap];
  sons[0] = M.popSon0[ap];
  sons[1] = M.popSon1[ap];
  taub[0] = gg_max2(M.popAge[sons[0]], M.popAge[sons[1]]);
  taub[0] = gg_max2(taub[0], M.sampleAge[sons[0]]);
  taub[0] = gg_max2(taub[0], M.sampleAge[sons[1]]);
  if (isRoot) taub[1] = GG_OLDAGE;
  else taub[1] = M.popAge[M.popFather[ap]];
  for (b = 0; b < G.B; b++) {
    src = M.bandSrc[b];
    tgt = M.bandTgt[b];
    if (src == ap || tgt == ap) taub[1] = gg_min2(taub[1], M.bandEnd[b]);
    else if (src == sons[0] || src == sons[1] || tgt == sons[0] || tgt == sons[1]) taub[0] = gg_max2(taub[0], M.bandStart[b]);
  }
  taunew = tauold + G.ftTaus[ap] * gg_rnd2normal8(G);
  taunew = gg_reflect(taunew, taub[0], taub[1]);
  M.popAge[ap] = taunew;   /* temporarily: band times under the proposal (GPhoCS.c:3302) */
  for (k = 0; k < 2; k++) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
  if (isRoot) taufactor[1] = taufactor[0];
  for (b = 0; b < G.B; b++) {
    src = M.bandSrc[b];
    tgt = M.bandTgt[b];
    res = gg_update_band_times(G, b);
    if ((src == sons[0] && tgt == sons[1]) || (src == sons[1] && tgt == sons[0])) {
    } else if (tgt == ap) {
      if (M.bandEnd[b] < taub[1]) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
        A.new_band_ages[num_aff] = taub[1] + (M.bandEnd[b] - taub[1]) / taufactor[1];
        num_aff++;
      }
      if (M.bandStart[b] < taub[1] && M.popAge[src] > gg_min2(tauold, taunew)) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
        A.new_band_ages[num_aff] = taub[1] + (M.bandStart[b] - taub[1]) / taufactor[1];
        if (A.new_band_ages[num_aff] < tauold) A.new_band_ages[num_aff] = tauold;
        num_aff++;
      }
    } else if (tgt == sons[0] || tgt == sons[1]) {
      if (M.bandStart[b] > taub[0]) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
        A.new_band_ages[num_aff] = taub[0] + (M.bandStart[b] - taub[0]) / taufactor[0];
        num_aff++;
      }
      if (M.bandEnd[b] > taub[0] && M.popAge[M.popFather[src]] < gg_max2(tauold, taunew)) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
        A.new_band_ages[num_aff] = taub[0] + (M.bandEnd[b] - taub[0]) / taufactor[0];
        num_aff++;
      }
    } else if (res && src == ap) {
      A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
      A.new_band_ages[num_aff] = M.bandStart[b];
      num_aff++;
    } else if (res && (src == sons[0] || src == sons[1])) {
      A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
      A.new_band_ages[num_aff] = M.bandEnd[b];
      num_aff++;
    }
  }
  M.popAge[ap] = tauold;   /* restored (GPhoCS.c:3444): kernels see the OLD age, NEW band times */
  G.pend_lnacc = gph_log(taunew / tauold) * (G.ageAlpha[ap] - 1) - (taunew - tauold) * G.ageBeta[ap];
  A.ap = ap; A.son0 = sons[0]; A.son1 = sons[1]; A.isRoot = isRoot; A.num_aff = num_aff;
  A.tauold = tauold; A.taunew = taunew; A.taub0 = taub[0]; A.taub1 = taub[1];
  A.taufactor0 = taufactor[0]; A.taufactor1 = taufactor[1];
  G.pend_pop = ap; G.pend_tauold = tauold; G.pend_taunew = taunew;
  G.pend_taufactor0 = taufactor[0]; G.pend_taufactor1 = taufactor[1];
}

// UpdateSampleAge, GPhoCS.c:4006-4128: the same for a current population with an estimated sample age (mode 1)
GPH_HD void gg_sage_propose(GphGlobal &G, int pop)
{
  GphModel &M = G.model;
  GphTauArgs &A = G.tau;
  int k, b, num_aff = 0;
  double tauold, taunew, taub[2], taufactor[2], age;
  gg_apply_pending(G);
  G.accArr[pop] = 0;
  tauold = M.sampleAge[pop];
  taub[0] = 0.0;
  taub[1] = M.popAge[M.popFather[pop]];
  taunew = tauold + G.ftTaus[pop] * gg_rnd2normal8(G);
  taunew = gg_reflect(taunew, taub[0], taub[1]);
  for (k = 0; k < 2; ++k) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
  for (b = 0; b < G.B; ++b) {
    if (M.bandTgt[b] != pop) continue;
    if (M.bandEnd[b] < taub[1] && M.bandEnd[b] > taub[0]) {
      age = M.bandEnd[b];
      A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
      A.new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
      ++num_aff;
    }
    if (M.bandStart[b] < taub[1] && M.bandStart[b] > taub[0]) {
      age = M.bandStart[b];
      A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
      A.new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
      if (A.new_band_ages[num_aff] < tauold) A.new_band_ages[num_aff] = tauold;
      ++num_aff;
    }
  }
  /* kernels see the OLD sample age (GPhoCS.c:4116) */
  G.pend_lnacc = gph_log(taunew / tauold) * (G.ageAlpha[pop] - 1) - (taunew - tauold) * G.ageBeta[pop];
  A.ap = pop; A.son0 = -1; A.son1 = -1; A.isRoot = 0; A.num_aff = num_aff; A.mode = 1;
  A.tauold = tauold; A.taunew = taunew; A.taub0 = taub[0]; A.taub1 = taub[1];
  A.taufactor0 = taufactor[0]; A.taufactor1 = taufactor[1];
  G.pend_pop = pop; G.pend_tauold = tauold; G.pend_taunew = taunew;
  G.pend_taufactor0 = taufactor[0]; G.pend_taufactor1 = taufactor[1];
}

// freeze what the finish of the decided proposal needs (gph_types.h: GphTauFin)
GPH_HD void gg_fill_fin(GphGlobal &G, int flag, long long limit)
{
  const GphTauArgs &A = G.tau;
  const GphModel &M = G.model;
  GphTauFin &F = G.fin;
  F.ap = A.ap; F.son0 = A.son0; F.son1 = A.son1; F.isRoot = A.isRoot; F.mode = A.mode; F.flag = flag; F.limit = limit;
  F.tauold = A.tauold; F.taunew = A.taunew; F.taub0 = A.taub0; F.taub1 = A.taub1;
  F.taufactor0 = A.taufactor0; F.taufactor1 = A.taufactor1;
  F.age_ap = M.popAge[A.ap];
  F.age_s0 = A.son0 >= 0 ? M.popAge[A.son0] : 0.0;
  F.age_s1 = A.son1 >= 0 ? M.popAge[A.son1] : 0.0;
}

// the decision of UpdateTau (GPhoCS.c:3835-3858, 3946, 3960) / UpdateSampleAge (:4447-4470) from the reduced
// vector of the evaluate kernel: section 0 sums 0 ntj0, 1 ntj1, 3 genDelta, 4 dataDelta, min 14 first conflicting
// locus (1e300 = none).  kind 1 = tau, 2 = sample age
GPH_HD void gg_tau_decide(GphGlobal &G, const GphRed &R, int kind)
{
  gg_count(G, R, 1);
  const int pop = G.pend_pop;
  const double ntj0 = (double)(int64_t)R.sum(0, 0), ntj1 = (double)(int64_t)R.sum(0, 1);
  const double genDelta = R.sum(0, 3), dataDelta = R.sum(0, 4);
  const double fc = R.mn(0, 14);
  const int mig_conflict = fc < 1e299;
  double lnacc = G.pend_lnacc;
  lnacc += dataDelta + genDelta + ntj0 * gph_log(G.pend_taufactor0) + ntj1 * gph_log(G.pend_taufactor1);
  if (!mig_conflict && (lnacc >= 0 || gg_rndu(G) < gph_exp(lnacc))) {
    G.accArr[pop]++;
    G.dataLogLikelihood += dataDelta;
    G.logLikelihood += (dataDelta + genDelta) / G.Ltot;
    G.tau_flag = 1;
    G.tau_limit = (long long)1 << 62;
    G.pend_kind = kind;          /* popAge / sampleAge = taunew once the commit kernel has run */
  } else {
    if (kind == 1) gg_compute_band_times(G);
    if (mig_conflict) G.rubberband_conflicts++;
    G.tau_flag = 0;
    G.tau_limit = mig_conflict ? (long long)fc : (long long)1 << 62;
    G.pend_kind = 0;
  }
  gg_fill_fin(G, G.tau_flag, G.tau_limit);
}
// end of UpdateTau / UpdateSampleAge: the record lines of performMCMC's caller (one per population, then the
// conflict count) and the accept totals (GPhoCS.c:1621-1650)
GPH_HD void gg_tau_end(GphGlobal &G, int kind)
{
  gg_apply_pending(G);
  if (kind == 1) {
    for (int ap = G.Kc; ap < G.K; ++ap) { G.acc[5] += G.accArr[ap]; G.accTau[ap] += G.accArr[ap]; }
    for (int ap = G.Kc; ap < G.K; ++ap) gg_rec(G, REC_TAU, ap, G.accArr[ap]);
    gg_rec(G, REC_CONFLICTS, 0, G.rubberband_conflicts);
  } else {
    for (int pop = 0; pop < G.Kc; ++pop) { if (!G.updateSampleAge[pop]) G.accArr[pop] = 0; G.acc[5] += G.accArr[pop]; G.accTau[pop] += G.accArr[pop]; }
    for (int pop = 0; pop < G.Kc; ++pop) {
      if (!G.updateSampleAge[pop]) continue;
      gg_rec(G, REC_SAGE, pop, G.accArr[pop]);
      gg_rec(G, REC_CONFLICTS, 0, G.rubberband_conflicts);
    }
  }
}

// mixing, GPhoCS.c:4688-4789: the proposal scales every parameter of the model
GPH_HD void gg_mix_propose(GphGlobal &G)
{
  GphModel &M = G.model;
  int pop, b;
  double xold, xnew, c, lnc, lnacc, dGen;
  long long num_events = 0;
  lnc = G.ftMixing * gg_rnd2normal8(G);
  c = gph_exp(lnc);
  for (pop = 0; pop < G.K; pop++) num_events += (long long)G.tot_ncoal[pop];
  for (b = 0; b < G.B; b++) num_events += (long long)G.tot_nmig[b];
  lnacc = lnc * (2 * G.K - G.Kc - G.B + num_events);
  dGen = 0.0;
  for (pop = 0; pop < G.K; pop++) {
    xold = M.theta[pop];
    xnew = xold * c;
    gg_set_theta(G, pop, xnew);
    lnacc += lnc * (G.thetaAlpha[pop] - 1) - (xnew - xold) * G.thetaBeta[pop];
    dGen -= lnc * G.tot_ncoal[pop];
    if (pop < G.Kc && M.sampleAge[pop] > 0.0) M.sampleAge[pop] *= c;
  }
  for (pop = G.Kc; pop < G.K; pop++) {
    xold = M.popAge[pop];
    M.popAge[pop] = xnew = xold * c;
    lnacc += lnc * (G.ageAlpha[pop] - 1) - (xnew - xold) * G.ageBeta[pop];
  }
  for (b = 0; b < G.B; b++) {
    xold = M.migRate[b];
    xnew = xold / c;
    gg_set_mig(G, b, xnew);
    lnacc += -lnc * (G.mrAlpha[b] - 1) - (xnew - xold) * G.mrBeta[b];
    M.bandStart[b] *= c;
    M.bandEnd[b] *= c;
    dGen -= lnc * G.tot_nmig[b];
  }
  G.mix_c = c; G.mix_lnc = lnc; G.pend_lnacc = lnacc; G.pend_dGen = dGen;
}
// mixing, GPhoCS.c:4803-4912: decision from the summed data log-likelihood change (section 0, sum 0)
GPH_HD void gg_mix_decide(GphGlobal &G, const GphRed &R)
{
  GphModel &M = G.model;
  gg_count(G, R, 2);
  const double dData = R.sum(0, 0), dGen = G.pend_dGen, c = G.mix_c;
  int pop, b;
  double lnacc = G.pend_lnacc;
  lnacc += (dData + dGen);
  if (lnacc >= 0 || gg_rndu(G) < gph_exp(lnacc)) {
    for (pop = 0; pop < G.K; pop++) G.tot_coal[pop] *= c;
    for (b = 0; b < G.B; b++) G.tot_mig[b] *= c;
    G.dataLogLikelihood += dData;
    G.logLikelihood += (dData + dGen) / G.Ltot;
    G.mix_flag = 1;
    G.acc[6] += 1;
    gg_rec(G, REC_MIX, 0, 1);
    return;
  }
  G.mix_flag = 0;
  for (pop = 0; pop < G.K; pop++) gg_set_theta(G, pop, M.theta[pop] / c);
  for (pop = 0; pop < G.K; pop++) {
    M.popAge[pop] /= c;
    if (pop < G.Kc && M.sampleAge[pop] > 0.0) M.sampleAge[pop] /= c;
  }
  for (b = 0; b < G.B; b++) {
    gg_set_mig(G, b, M.migRate[b] * c);
    M.bandStart[b] /= c;
    M.bandEnd[b] /= c;
  }
  gg_rec(G, REC_MIX, 0, 0);
}

// one stage above the loci; `arg` = population / flag of the stage
GPH_HD void gg_stage(GphGlobal &G, const GphRed &R, int stage, int arg)
{
  switch (stage) {
  case GS_INIT_DONE:      /* initializeMCMC, GPhoCS.c:1216-1224: out 0 = genLnL, 1 = dataLnL */
    gg_count(G, R, 3);
    G.dataLogLikelihood = R.sum(0, 1);
    G.logLikelihood = (R.sum(0, 0) + R.sum(0, 1)) / G.Ltot;
    gg_totals(G, R);
    gg_rec(G, REC_INIT, 0, (long long)(G.Ltot * (G.n - 1)));
    break;
  case GS_SWEEP_DONE: gg_sweep_done(G, R, arg); break;
  case GS_TOTALS: gg_totals(G, R); break;
  case GS_THETA: gg_theta_and_mig_rates(G); break;
  case GS_TAU_PROPOSE: gg_tau_propose(G, arg); break;
  case GS_TAU_DECIDE: gg_tau_decide(G, R, 1); break;
  case GS_TAU_END: gg_tau_end(G, 1); break;
  case GS_SAGE_PROPOSE: gg_sage_propose(G, arg); break;
  case GS_SAGE_DECIDE: gg_tau_decide(G, R, 2); break;
  case GS_SAGE_END: gg_tau_end(G, 2); break;
  case GS_MIX_PROPOSE:    /* finetune <= 0: mixing() returns 0 before drawing anything (GPhoCS.c:4700) */
    if (G.ftMixing <= 0.0) gg_rec(G, REC_MIX, 0, 0);
    else gg_mix_propose(G);
    break;
  case GS_MIX_DECIDE: gg_mix_decide(G, R); break;
  case GS_STARTMIG:       /* sampleMigRates, PopulationTree.c:414-429 (iteration == start-mig, GPhoCS.c:1738-1745) */
    for (int b = 0; b < G.B; b++) G.migRateShown[b] = G.model.migRate[b];
    G.shownValid = 1;
    for (int b = 0; b < G.B; b++) {
      double mean = G.mrAlpha[b] / G.mrBeta[b];
      gg_set_mig(G, b, mean * (0.9 + 0.2 * gg_rndu(G)));
    }
    break;
  case GS_REFRESH_DONE:   /* genLogLikelihood refresh, GPhoCS.c:1749-1757: out 0 ok, 1 old, 2 new genLnL */
    gg_count(G, R, 8);
    if (R.mn(0, 0) < 1.0 && !G.error) { G.error = 75; G.error_locus = -1; }
    if (arg) {
      G.logLikelihood -= R.sum(0, 1) / G.Ltot;
      G.logLikelihood += R.sum(0, 2) / G.Ltot;
    }
    break;
  case GS_CHECK_DONE:     /* checkAll, patch.c:2745-2884: out 0 ok, 1 dataLnL, 2 genLnL; statistics in section 1 */
    gg_count(G, R, 4);
    if (R.mn(0, 0) < 1.0 && !G.error) { G.error = 9999; G.error_locus = -1; }
    G.dataLogLikelihood = R.sum(0, 1);
    G.logLikelihood = (R.sum(0, 2) + R.sum(0, 1)) / G.Ltot;
    gg_totals(G, R);
    gg_rec(G, REC_CHECK, 0, 1);
    break;
  case GS_COUNT_ONLY: gg_count(G, R, arg); break;
  case GS_THETA_ONLY: G.napply = 0; gg_update_theta(G); break;
  case GS_MIGR_ONLY: G.napply = 0; gg_update_mig_rates(G); break;
  default: break;
  }
}
