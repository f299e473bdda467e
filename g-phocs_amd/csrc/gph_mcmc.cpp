// gph_mcmc.cpp -- host MCMC driver above the engine's C ABI (include/gphocs_hip.h).
//
// One iteration of the reference's performMCMC (upstream src/GPhoCS.c:1476-1821) with the
// per-locus loops replaced by engine calls.  The host keeps what the reference keeps on its
// main thread: the general-purpose RNG slot (utils.h:34), the priors, the accept decisions of
// the global proposals (UpdateTheta GPhoCS.c:3037, UpdateMigRates :3115, UpdateTau :3224,
// mixing :4688) and the accumulators dataState.{logLikelihood,dataLogLikelihood}.
// With one process per GPU every rank runs this driver on identical inputs (same general
// RNG stream, all-reduced sums), so every rank takes the same decisions.
//
// Compiled with -ffp-contract=off: the expressions keep the reference's operation order.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/gphocs_hip.h"

#define OLDAGE 999.0
#define max2(a, b) ((a) > (b) ? (a) : (b))
#define min2(a, b) ((a) < (b) ? (a) : (b))

struct gph_mcmc {
  gph_engine *e;
  int n, Kc, K, B, rootPop;
  int64_t Ltot;
  std::vector<int> popFather, popSon0, popSon1, bandSrc, bandTgt, updateSampleAge;
  std::vector<double> theta, popAge, sampleAge, migRate, bandStart, bandEnd;
  std::vector<double> thetaAlpha, thetaBeta, thetaStart, ageAlpha, ageBeta, ageStart, mrAlpha, mrBeta, ftTaus;
  std::vector<double> printFactors, paramVals;
  double ftCoalTime, ftMigTime, ftTheta, ftMigRate, ftMixing;
  int mutRateMode;              // 0 CONST, 1 VAR (UpdateLocusRate is live), 2 FIXED
  double varRatesAlpha, ftLocusRate, rateVar;
  int64_t accLocusRate;
  int seed, startMig, doMixing, samplesPerLog, numParameters;
  uint32_t gx, gy, gz;          // general RNG slot
  double logLikelihood, dataLogLikelihood;
  std::vector<double> tot_coal, tot_ncoal, tot_mig, tot_nmig;
  int64_t rubberband_conflicts;
  int64_t acc[9];               // coalTime, migTime, SPR, theta, migRate, taus(sum), mixing, totalMigNodes, -
  std::vector<int64_t> accTau;  // per population: accepted UpdateTau / UpdateSampleAge proposals
  FILE *rec;
};

// rndu / rndnormal / rnd2normal8 for the general slot: utils.c:459-513
static double g_rndu(gph_mcmc *m)
{
  double r;
  m->gx = 171u * (m->gx % 177u) - 2u * (m->gx / 177u);
  m->gy = 172u * (m->gy % 176u) - 35u * (m->gy / 176u);
  m->gz = 170u * (m->gz % 178u) - 63u * (m->gz / 178u);
  r = m->gx / 30269.0 + m->gy / 30307.0 + m->gz / 30323.0;
  r = (r - (int)r);
  return r;
}
static double g_rndnormal(gph_mcmc *m)
{
  double u, v, s;
  for (;;) {
    u = 2 * g_rndu(m) - 1;
    v = 2 * g_rndu(m) - 1;
    s = u * u + v * v;
    if (s > 0 && s < 1) break;
  }
  s = sqrt(-2. * log(s) / s);
  return u * s;
}
static double g_rnd2normal8(gph_mcmc *m)
{
  const double m2s2 = 8.;
  double m2N = sqrt(m2s2 / (m2s2 + 1.));
  double s2N = sqrt(1. / (m2s2 + 1.));
  double z = m2N + g_rndnormal(m) * s2N;
  z = g_rndu(m) < 0.5 ? z : -z;
  return z;
}
// reflect, utils.c:333-398
static double h_reflect(double x, double a, double b)
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
static int update_band_times(gph_mcmc *m, int b)
{
  int res = 0, src = m->bandSrc[b], tgt = m->bandTgt[b];
  double t = max2(m->popAge[src], m->popAge[tgt]);
  if (t != m->bandStart[b]) { m->bandStart[b] = t; res = 1; }
  t = min2(m->popAge[m->popFather[src]], m->popAge[m->popFather[tgt]]);
  if (t != m->bandEnd[b]) { m->bandEnd[b] = t; res = 1; }
  return res;
}
static void compute_band_times(gph_mcmc *m)
{
  for (int b = 0; b < m->B; b++) {
    update_band_times(m, b);
    if (m->bandStart[b] >= m->bandEnd[b]) m->bandStart[b] = m->bandEnd[b] = m->popAge[m->bandTgt[b]];
  }
}
static int push_model(gph_mcmc *m)
{
  return gph_engine_set_model(m->e, m->theta.data(), m->popAge.data(), m->sampleAge.data(), m->migRate.data(),
                              m->bandStart.data(), m->bandEnd.data());
}
static int refresh_totals(gph_mcmc *m)
{
  return gph_engine_get_totals(m->e, m->tot_coal.data(), m->tot_ncoal.data(), m->tot_mig.data(), m->tot_nmig.data());
}
static void rec_line(gph_mcmc *m, int it, const char *what, long acc)
{
  if (m->rec) fprintf(m->rec, "IT %d %s %ld %a %a\n", it, what, acc, m->dataLogLikelihood, m->logLikelihood);
}

// samplePopParameters, PopulationTree.c:339-403
static void sample_pop_parameters(gph_mcmc *m)
{
  std::vector<int> queue(m->K);
  int head = 0, tail = 0, pop;
  double mean;
  queue[tail++] = m->rootPop;
  while (head < tail) {
    pop = queue[head++];
    mean = m->thetaStart[pop];
    m->theta[pop] = mean * (0.9 + 0.2 * g_rndu(m));
    if (m->popSon0[pop] >= 0) {
      mean = m->ageStart[pop];
      m->popAge[pop] = mean * (0.9 + 0.2 * g_rndu(m));
      if (m->popFather[pop] >= 0 && m->popAge[m->popFather[pop]] < m->popAge[pop]) {
        m->popAge[pop] = max2(m->sampleAge[m->popSon0[pop]], m->sampleAge[m->popSon1[pop]]);
        m->popAge[pop] += (m->popAge[m->popFather[pop]] - m->popAge[pop]) * (0.93 + 0.004 * g_rndu(m));
      }
      queue[tail++] = m->popSon0[pop];
      queue[tail++] = m->popSon1[pop];
    }
  }
  for (int b = 0; b < m->B; b++) m->migRate[b] = 0.0;
  compute_band_times(m);
}

// UpdateTheta, GPhoCS.c:3037-3107
static int update_theta(gph_mcmc *m, double finetune, int *accepted)
{
  int rc;
  *accepted = 0;
  if (finetune <= 0.0) return 0;
  for (int pop = 0; pop < m->K; pop++) {
    double thetaold = m->theta[pop];
    double lnc = finetune * g_rnd2normal8(m);
    double c = exp(lnc);
    double thetanew = thetaold * c;
    double lnacc = lnc + lnc * (m->thetaAlpha[pop] - 1) - (thetanew - thetaold) * m->thetaBeta[pop];
    double dLL = -(lnc * m->tot_ncoal[pop] + (1 / thetanew - 1 / thetaold) * m->tot_coal[pop]);
    lnacc += dLL;
    if (lnacc >= 0 || g_rndu(m) < exp(lnacc)) {
      (*accepted)++;
      if ((rc = gph_engine_apply_theta(m->e, pop, lnc, thetaold, thetanew))) return rc;
      m->logLikelihood += dLL / m->Ltot;
      m->theta[pop] = thetanew;
    }
  }
  return 0;
}

// UpdateMigRates, GPhoCS.c:3115-3213
static int update_mig_rates(gph_mcmc *m, double finetune, int *accepted)
{
  int rc;
  *accepted = 0;
  if (finetune <= 0.0) return 0;
  for (int b = 0; b < m->B; b++) {
    double old_rate = m->migRate[b];
    double lnc = finetune * g_rnd2normal8(m);
    double c = exp(lnc);
    double new_rate = old_rate * c;
    if (new_rate < 0.00001) continue;
    double lnacc = lnc + lnc * (m->mrAlpha[b] - 1) - (new_rate - old_rate) * m->mrBeta[b];
    double dLL = (lnc * m->tot_nmig[b] - (new_rate - old_rate) * m->tot_mig[b]);
    lnacc += dLL;
    if (lnacc >= 0 || g_rndu(m) < exp(lnacc)) {
      (*accepted)++;
      if ((rc = gph_engine_apply_migrate(m->e, b, lnc, old_rate, new_rate))) return rc;
      m->migRate[b] = new_rate;
      m->logLikelihood += dLL / m->Ltot;
    }
  }
  return 0;
}

// UpdateTau, GPhoCS.c:3224-3994: host part (bounds, proposal, affected bands, decision)
static int update_tau(gph_mcmc *m, int iteration, int *accepted)
{
  int rc;
  for (int ap = m->Kc; ap < m->K; ++ap) {
    gph_tau_args A;
    gph_tau_result R;
    int sons[2], isRoot, k, b, src, tgt, res, num_aff = 0;
    double tauold, taunew, taub[2], taufactor[2], lnacc;
    memset(&A, 0, sizeof A);
    accepted[ap] = 0;
    isRoot = (ap == m->rootPop);
    tauold = m->popAge[ap];
    sons[0] = m->popSon0[ap];
    sons[1] = m->popSon1[ap];
    taub[0] = max2(m->popAge[sons[0]], m->popAge[sons[1]]);
    taub[0] = max2(taub[0], m->sampleAge[sons[0]]);
    taub[0] = max2(taub[0], m->sampleAge[sons[1]]);
    if (isRoot) taub[1] = OLDAGE;
    else taub[1] = m->popAge[m->popFather[ap]];
    for (b = 0; b < m->B; b++) {
      src = m->bandSrc[b];
      tgt = m->bandTgt[b];
      if (src == ap || tgt == ap) taub[1] = min2(taub[1], m->bandEnd[b]);
      else if (src == sons[0] || src == sons[1] || tgt == sons[0] || tgt == sons[1]) taub[0] = max2(taub[0], m->bandStart[b]);
    }
    taunew = tauold + m->ftTaus[ap] * g_rnd2normal8(m);
    taunew = h_reflect(taunew, taub[0], taub[1]);
    m->popAge[ap] = taunew;   /* temporarily: band times under the proposal (GPhoCS.c:3302) */
    for (k = 0; k < 2; k++) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
    if (isRoot) taufactor[1] = taufactor[0];
    for (b = 0; b < m->B; b++) {
      src = m->bandSrc[b];
      tgt = m->bandTgt[b];
      res = update_band_times(m, b);
      if ((src == sons[0] && tgt == sons[1]) || (src == sons[1] && tgt == sons[0])) {
      } else if (tgt == ap) {
        if (m->bandEnd[b] < taub[1]) {
          A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
          A.new_band_ages[num_aff] = taub[1] + (m->bandEnd[b] - taub[1]) / taufactor[1];
          num_aff++;
        }
        if (m->bandStart[b] < taub[1] && m->popAge[src] > min2(tauold, taunew)) {
          A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
          A.new_band_ages[num_aff] = taub[1] + (m->bandStart[b] - taub[1]) / taufactor[1];
          if (A.new_band_ages[num_aff] < tauold) A.new_band_ages[num_aff] = tauold;
          num_aff++;
        }
      } else if (tgt == sons[0] || tgt == sons[1]) {
        if (m->bandStart[b] > taub[0]) {
          A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
          A.new_band_ages[num_aff] = taub[0] + (m->bandStart[b] - taub[0]) / taufactor[0];
          num_aff++;
        }
        if (m->bandEnd[b] > taub[0] && m->popAge[m->popFather[src]] < max2(tauold, taunew)) {
          A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
          A.new_band_ages[num_aff] = taub[0] + (m->bandEnd[b] - taub[0]) / taufactor[0];
          num_aff++;
        }
      } else if (res && src == ap) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
        A.new_band_ages[num_aff] = m->bandStart[b];
        num_aff++;
      } else if (res && (src == sons[0] || src == sons[1])) {
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
        A.new_band_ages[num_aff] = m->bandEnd[b];
        num_aff++;
      }
    }
    m->popAge[ap] = tauold;   /* restored (GPhoCS.c:3444): kernels see the OLD age, NEW band times */
    lnacc = log(taunew / tauold) * (m->ageAlpha[ap] - 1) - (taunew - tauold) * m->ageBeta[ap];
    A.ap = ap; A.son0 = sons[0]; A.son1 = sons[1]; A.isRoot = isRoot; A.num_aff = num_aff;
    A.tauold = tauold; A.taunew = taunew; A.taub0 = taub[0]; A.taub1 = taub[1];
    A.taufactor0 = taufactor[0]; A.taufactor1 = taufactor[1];
    if ((rc = push_model(m))) return rc;
    if ((rc = gph_engine_tau_evaluate(m->e, &A, &R))) return rc;
    int mig_conflict = R.first_conflict_locus >= 0;
    lnacc += R.dataDelta + R.genDelta + R.ntj0 * log(taufactor[0]) + R.ntj1 * log(taufactor[1]);
    if (!mig_conflict && (lnacc >= 0 || g_rndu(m) < exp(lnacc))) {
      accepted[ap]++;
      m->dataLogLikelihood += R.dataDelta;
      m->logLikelihood += (R.dataDelta + R.genDelta) / m->Ltot;
      if ((rc = gph_engine_tau_commit(m->e))) return rc;
      m->popAge[ap] = taunew;
    } else {
      compute_band_times(m);
      if (mig_conflict) m->rubberband_conflicts++;
      if ((rc = push_model(m))) return rc;
      if ((rc = gph_engine_tau_revert(m->e, R.first_conflict_locus))) return rc;
    }
  }
  for (int ap = m->Kc; ap < m->K; ++ap) {
    char nm[32];
    snprintf(nm, sizeof nm, "TAU%d", ap);
    rec_line(m, iteration, nm, accepted[ap]);
  }
  if (m->rec) fprintf(m->rec, "CONFLICTS %lld\n", (long long)m->rubberband_conflicts);
  return push_model(m);
}

// UpdateSampleAge, GPhoCS.c:4006-4584: host part (bounds, proposal, affected bands, decision);
// the per-locus loops run in the tau kernels with mode = 1
static int update_sample_age(gph_mcmc *m, int iteration, int *accepted)
{
  int rc;
  for (int pop = 0; pop < m->Kc; ++pop) {
    accepted[pop] = 0;
    if (!m->updateSampleAge[pop]) continue;
    gph_tau_args A;
    gph_tau_result R;
    int k, b, num_aff = 0;
    double tauold, taunew, taub[2], taufactor[2], lnacc, age;
    memset(&A, 0, sizeof A);
    tauold = m->sampleAge[pop];
    taub[0] = 0.0;
    taub[1] = m->popAge[m->popFather[pop]];
    taunew = tauold + m->ftTaus[pop] * g_rnd2normal8(m);
    taunew = h_reflect(taunew, taub[0], taub[1]);
    for (k = 0; k < 2; ++k) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
    for (b = 0; b < m->B; ++b) {
      if (m->bandTgt[b] != pop) continue;
      if (m->bandEnd[b] < taub[1] && m->bandEnd[b] > taub[0]) {
        age = m->bandEnd[b];
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 0;
        A.new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
        ++num_aff;
      }
      if (m->bandStart[b] < taub[1] && m->bandStart[b] > taub[0]) {
        age = m->bandStart[b];
        A.aff_bands[num_aff] = b; A.start_or_end[num_aff] = 1;
        A.new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
        if (A.new_band_ages[num_aff] < tauold) A.new_band_ages[num_aff] = tauold;
        ++num_aff;
      }
    }
    /* kernels see the OLD sample age (GPhoCS.c:4116) */
    lnacc = log(taunew / tauold) * (m->ageAlpha[pop] - 1) - (taunew - tauold) * m->ageBeta[pop];
    A.ap = pop; A.son0 = -1; A.son1 = -1; A.isRoot = 0; A.num_aff = num_aff; A.mode = 1;
    A.tauold = tauold; A.taunew = taunew; A.taub0 = taub[0]; A.taub1 = taub[1];
    A.taufactor0 = taufactor[0]; A.taufactor1 = taufactor[1];
    if ((rc = push_model(m))) return rc;
    if ((rc = gph_engine_tau_evaluate(m->e, &A, &R))) return rc;
    int mig_conflict = R.first_conflict_locus >= 0;
    lnacc += R.dataDelta + R.genDelta + R.ntj0 * log(taufactor[0]) + R.ntj1 * log(taufactor[1]);
    if (!mig_conflict && (lnacc >= 0 || g_rndu(m) < exp(lnacc))) {
      ++accepted[pop];
      m->dataLogLikelihood += R.dataDelta;
      m->logLikelihood += (R.dataDelta + R.genDelta) / m->Ltot;
      if ((rc = gph_engine_tau_commit(m->e))) return rc;
      m->sampleAge[pop] = taunew;
    } else {
      if (mig_conflict) m->rubberband_conflicts++;
      if ((rc = gph_engine_tau_revert(m->e, R.first_conflict_locus))) return rc;
    }
  }
  for (int pop = 0; pop < m->Kc; ++pop) {
    char nm[32];
    if (!m->updateSampleAge[pop]) continue;
    snprintf(nm, sizeof nm, "SAGE%d", pop);
    rec_line(m, iteration, nm, accepted[pop]);
    if (m->rec) fprintf(m->rec, "CONFLICTS %lld\n", (long long)m->rubberband_conflicts);
  }
  return push_model(m);
}

// mixing, GPhoCS.c:4688-4912: host part
static int mixing(gph_mcmc *m, double finetune, int *accepted)
{
  int rc, pop, b;
  double xold, xnew, c, lnc, lnacc, dData, dGen;
  long num_events = 0;
  *accepted = 0;
  if (finetune <= 0.0) return 0;
  lnc = finetune * g_rnd2normal8(m);
  c = exp(lnc);
  for (pop = 0; pop < m->K; pop++) num_events += (long)m->tot_ncoal[pop];
  for (b = 0; b < m->B; b++) num_events += (long)m->tot_nmig[b];
  lnacc = lnc * (2 * m->K - m->Kc - m->B + num_events);
  dData = 0.0;
  dGen = 0.0;
  for (pop = 0; pop < m->K; pop++) {
    xold = m->theta[pop];
    m->theta[pop] = xnew = xold * c;
    lnacc += lnc * (m->thetaAlpha[pop] - 1) - (xnew - xold) * m->thetaBeta[pop];
    dGen -= lnc * m->tot_ncoal[pop];
    if (pop < m->Kc && m->sampleAge[pop] > 0.0) m->sampleAge[pop] *= c;
  }
  for (pop = m->Kc; pop < m->K; pop++) {
    xold = m->popAge[pop];
    m->popAge[pop] = xnew = xold * c;
    lnacc += lnc * (m->ageAlpha[pop] - 1) - (xnew - xold) * m->ageBeta[pop];
  }
  for (b = 0; b < m->B; b++) {
    xold = m->migRate[b];
    m->migRate[b] = xnew = xold / c;
    lnacc += -lnc * (m->mrAlpha[b] - 1) - (xnew - xold) * m->mrBeta[b];
    m->bandStart[b] *= c;
    m->bandEnd[b] *= c;
    dGen -= lnc * m->tot_nmig[b];
  }
  if ((rc = push_model(m))) return rc;
  if ((rc = gph_engine_mixing_evaluate(m->e, c, &dData))) return rc;
  lnacc += (dData + dGen);
  if (lnacc >= 0 || g_rndu(m) < exp(lnacc)) {
    if ((rc = gph_engine_mixing_commit(m->e, c, lnc))) return rc;
    for (pop = 0; pop < m->K; pop++) m->tot_coal[pop] *= c;
    for (b = 0; b < m->B; b++) m->tot_mig[b] *= c;
    m->dataLogLikelihood += dData;
    m->logLikelihood += (dData + dGen) / m->Ltot;
    *accepted = 1;
    return 0;
  }
  if ((rc = gph_engine_mixing_revert(m->e))) return rc;
  for (pop = 0; pop < m->K; pop++) m->theta[pop] /= c;
  for (pop = 0; pop < m->K; pop++) {
    m->popAge[pop] /= c;
    if (pop < m->Kc && m->sampleAge[pop] > 0.0) m->sampleAge[pop] /= c;
  }
  for (b = 0; b < m->B; b++) {
    m->migRate[b] *= c;
    m->bandStart[b] /= c;
    m->bandEnd[b] /= c;
  }
  return push_model(m);
}

// recordParamVals, GPhoCS.c:802-849
static void record_param_vals(gph_mcmc *m)
{
  int ind = 0;
  m->paramVals.assign(m->numParameters + 4, 0.0);
  for (int pop = 0; pop < m->K; pop++) m->paramVals[ind++] = m->theta[pop];
  for (int pop = m->Kc; pop < m->K; pop++) m->paramVals[ind++] = m->popAge[pop];
  for (int b = 0; b < m->B; b++) m->paramVals[ind++] = m->migRate[b];
  for (int pop = 0; pop < m->Kc; pop++)
    if (m->updateSampleAge[pop] || m->sampleAge[pop] > 0.0) m->paramVals[ind++] = m->sampleAge[pop];
  if (m->mutRateMode == 1) m->paramVals[ind++] = sqrt(m->rateVar);   /* GPhoCS.c:842-847 */
}

extern "C" {

int gph_mcmc_create(gph_engine *e, const gph_config *cfg, const gph_mcmc_config *mc, gph_mcmc **out)
{
  if (!e || !cfg || !mc || !out) return GPH_EARG;
  gph_mcmc *m = new gph_mcmc();
  m->e = e;
  m->n = cfg->n; m->Kc = cfg->Kc; m->K = cfg->K; m->B = cfg->B; m->rootPop = cfg->rootPop; m->Ltot = cfg->L_total;
  m->popFather.assign(cfg->popFather, cfg->popFather + m->K);
  m->popSon0.assign(cfg->popSon0, cfg->popSon0 + m->K);
  m->popSon1.assign(cfg->popSon1, cfg->popSon1 + m->K);
  if (m->B) { m->bandSrc.assign(cfg->bandSrc, cfg->bandSrc + m->B); m->bandTgt.assign(cfg->bandTgt, cfg->bandTgt + m->B); }
  m->theta.assign(m->K, 0.0); m->popAge.assign(m->K, 0.0);
  m->sampleAge.assign(mc->sampleAge, mc->sampleAge + m->K);
  m->updateSampleAge.assign(m->K, 0);
  if (mc->updateSampleAge) for (int p = 0; p < m->Kc; p++) m->updateSampleAge[p] = mc->updateSampleAge[p] != 0;
  m->migRate.assign(m->B + 1, 0.0); m->bandStart.assign(m->B + 1, 0.0); m->bandEnd.assign(m->B + 1, 0.0);
  m->thetaAlpha.assign(mc->thetaAlpha, mc->thetaAlpha + m->K);
  m->thetaBeta.assign(mc->thetaBeta, mc->thetaBeta + m->K);
  m->thetaStart.assign(mc->thetaStart, mc->thetaStart + m->K);
  m->ageAlpha.assign(mc->ageAlpha, mc->ageAlpha + m->K);
  m->ageBeta.assign(mc->ageBeta, mc->ageBeta + m->K);
  m->ageStart.assign(mc->ageStart, mc->ageStart + m->K);
  m->mrAlpha.assign(m->B + 1, 0.0); m->mrBeta.assign(m->B + 1, 1.0);
  for (int b = 0; b < m->B; b++) { m->mrAlpha[b] = mc->mrAlpha[b]; m->mrBeta[b] = mc->mrBeta[b]; }
  m->ftTaus.assign(mc->ftTaus, mc->ftTaus + m->K);
  m->ftCoalTime = mc->ftCoalTime; m->ftMigTime = mc->ftMigTime; m->ftTheta = mc->ftTheta;
  m->ftMigRate = mc->ftMigRate; m->ftMixing = mc->ftMixing;
  m->mutRateMode = mc->mutRateMode; m->varRatesAlpha = mc->varRatesAlpha; m->ftLocusRate = mc->ftLocusRate;
  m->rateVar = 0.0; m->accLocusRate = 0;
  m->seed = mc->seed; m->startMig = mc->startMig; m->doMixing = mc->doMixing; m->samplesPerLog = mc->samplesPerLog;
  m->numParameters = mc->numParameters;
  m->printFactors.assign(mc->printFactors, mc->printFactors + mc->numParameters);
  m->tot_coal.assign(m->K, 0.0); m->tot_ncoal.assign(m->K, 0.0); m->tot_mig.assign(m->B + 1, 0.0); m->tot_nmig.assign(m->B + 1, 0.0);
  m->gx = 11; m->gy = 23; m->gz = 170u * ((uint32_t)m->seed % 178u) + 137u;   // utils.c:421-426
  m->logLikelihood = m->dataLogLikelihood = 0.0;
  m->rubberband_conflicts = 0;
  memset(m->acc, 0, sizeof m->acc);
  m->accTau.assign(m->K, 0);
  m->rec = nullptr;
  *out = m;
  return 0;
}

void gph_mcmc_destroy(gph_mcmc *m)
{
  if (!m) return;
  if (m->rec) fclose(m->rec);
  delete m;
}

int gph_mcmc_set_record_file(gph_mcmc *m, const char *path)
{
  if (!m) return GPH_EARG;
  if (m->rec) fclose(m->rec);
  m->rec = path ? fopen(path, "w") : nullptr;
  return (path && !m->rec) ? GPH_EARG : 0;
}

// initializeMCMC, GPhoCS.c:1122-1225
int gph_mcmc_initialize(gph_mcmc *m, int64_t *totalCoals)
{
  int rc;
  double sumGen = 0, sumData = 0;
  if (!m) return GPH_EARG;
  sample_pop_parameters(m);
  if (m->mutRateMode == 1) {
    /* locus rates 0.8 + 0.4 u from each locus's own stream, normalised to mean 1 (GPhoCS.c:1157-1178).  Every
     * locus stream starts in the same state (utils.c:421-426), so every locus draws the same u: the host
     * replays the sums of the reference loop and hands the engine the rates and the one spent draw. */
    uint32_t x = 11, y = 23, z = 170u * ((uint32_t)m->seed % 178u) + 137u;
    x = 171u * (x % 177u) - 2u * (x / 177u);
    y = 172u * (y % 176u) - 35u * (y / 176u);
    z = 170u * (z % 178u) - 63u * (z / 178u);
    double u = x / 30269.0 + y / 30307.0 + z / 30323.0;
    u = (u - (int)u);
    const double r0 = 0.8 + 0.4 * u;
    double total = 0.0;
    for (int64_t g = 0; g < m->Ltot; g++) total += r0;
    total /= m->Ltot;
    const double r = r0 / total;
    m->rateVar = 0.0;
    for (int64_t g = 0; g < m->Ltot; g++) m->rateVar += (r - 1) * (r - 1);
    m->rateVar /= m->Ltot;
    std::vector<double> rates((size_t)gph_engine_num_loci(m->e), r);
    if ((rc = gph_engine_set_locus_rates(m->e, rates.data(), 1, 1))) return rc;
  }
  if ((rc = push_model(m))) return rc;
  if ((rc = gph_engine_seed(m->e, (uint32_t)m->seed))) return rc;
  if ((rc = gph_engine_init_genealogies(m->e, &sumGen, &sumData))) return rc;
  m->dataLogLikelihood = sumData;
  m->logLikelihood = (sumGen + sumData) / m->Ltot;
  if ((rc = refresh_totals(m))) return rc;
  if (totalCoals) *totalCoals = m->Ltot * (m->n - 1);
  rec_line(m, -1, "INIT", (long)(m->Ltot * (m->n - 1)));
  return 0;
}

int gph_mcmc_iteration(gph_mcmc *m, int32_t iteration)
{
  int rc, acc;
  gph_sweep_result S;
  std::vector<int> accArr(m->K, 0);
  if (!m) return GPH_EARG;
  // the three genealogy proposals run fused in one launch (GPhoCS.c:1495-1538)
  if ((rc = gph_engine_genealogy_sweep(m->e, 7, m->ftCoalTime, m->ftMigTime, &S))) return rc;
  m->dataLogLikelihood += S.dData_internal;
  m->logLikelihood += S.dLog_internal;
  m->acc[0] += S.accepted_internal;
  rec_line(m, iteration, "INT", (long)S.accepted_internal);
  m->logLikelihood += S.dLog_mignode;
  m->acc[1] += S.accepted_mignode;
  m->acc[7] += S.total_mig_nodes;
  rec_line(m, iteration, "MIGN", (long)S.accepted_mignode);
  m->dataLogLikelihood += S.dData_spr;
  m->logLikelihood += S.dLog_spr;
  m->acc[2] += S.accepted_spr;
  rec_line(m, iteration, "SPR", (long)S.accepted_spr);
  if (m->mutRateMode == 1) {
    // UpdateLocusRate, GPhoCS.c:1554-1563, 4598-4680
    gph_locus_rate_result R;
    R.accepted = 0; R.dataLogLikelihood = m->dataLogLikelihood; R.logLikelihood = m->logLikelihood; R.rateVar = m->rateVar;
    if ((rc = gph_engine_locus_rate_update(m->e, m->ftLocusRate, m->varRatesAlpha, &R))) return rc;
    m->dataLogLikelihood = R.dataLogLikelihood; m->logLikelihood = R.logLikelihood; m->rateVar = R.rateVar;
    m->accLocusRate += R.accepted;
    rec_line(m, iteration, "LRATE", (long)R.accepted);
  }
  if ((rc = refresh_totals(m))) return rc;
  if ((rc = update_theta(m, m->ftTheta, &acc))) return rc;
  m->acc[3] += acc;
  rec_line(m, iteration, "THETA", acc);
  if (iteration > m->startMig) {
    if ((rc = update_mig_rates(m, m->ftMigRate, &acc))) return rc;
    m->acc[4] += acc;
    rec_line(m, iteration, "MIGR", acc);
  }
  if ((rc = update_tau(m, iteration, accArr.data()))) return rc;
  for (int pop = m->Kc; pop < m->K; pop++) { m->acc[5] += accArr[pop]; m->accTau[pop] += accArr[pop]; }
  if ((rc = update_sample_age(m, iteration, accArr.data()))) return rc;
  for (int pop = 0; pop < m->Kc; pop++) { m->acc[5] += accArr[pop]; m->accTau[pop] += accArr[pop]; }
  if (m->doMixing) {
    /* mixing reads the event COUNTS only (GPhoCS.c:4722-4760); UpdateTau / UpdateSampleAge move events inside
     * their populations and change no count, so the totals taken after the genealogy sweep still hold them */
    if ((rc = mixing(m, m->ftMixing, &acc))) return rc;
    m->acc[6] += acc;
    rec_line(m, iteration, "MIX", acc);
  }
  record_param_vals(m);
  {
    int refresh = (iteration == m->startMig);
    double oldGen = 0, newGen = 0;
    if (refresh) {
      // sampleMigRates, PopulationTree.c:414-429, then genLogLikelihood refresh (GPhoCS.c:1738-1757)
      for (int b = 0; b < m->B; b++) {
        double mean = m->mrAlpha[b] / m->mrBeta[b];
        m->migRate[b] = mean * (0.9 + 0.2 * g_rndu(m));
      }
    }
    if ((rc = push_model(m))) return rc;
    if ((rc = gph_engine_synchronize(m->e, refresh, &oldGen, &newGen))) return rc;
    if (refresh) {
      m->logLikelihood -= oldGen / m->Ltot;
      m->logLikelihood += newGen / m->Ltot;
    }
  }
  if ((iteration + 1) % m->samplesPerLog == 0) {
    // checkAll, patch.c:2745-2884: consistency checks + accumulator resynchronisation
    int32_t ok = 0;
    double sumData = 0, sumGen = 0;
    if ((rc = gph_engine_check_all(m->e, &ok, &sumData, &sumGen))) return rc;
    if (!ok) { fprintf(stderr, "gphocs_hip: checkAll failed at iteration %d\n", iteration); return GPH_EKERNEL; }
    m->dataLogLikelihood = sumData;
    m->logLikelihood = (sumGen + sumData) / m->Ltot;
    if ((rc = refresh_totals(m))) return rc;
    rec_line(m, iteration, "CHECK", 1);
  }
  if (m->rec) {
    // trace row, GPhoCS.c:746-754, 1763-1769
    fprintf(m->rec, "TRACE %d\t", iteration);
    for (int i = 0; i < m->numParameters; i++) fprintf(m->rec, "%8.5f\t", m->paramVals[i] * m->printFactors[i]);
    fprintf(m->rec, "\t%.6f\t%.6f\n", m->logLikelihood, m->dataLogLikelihood);
    fflush(m->rec);
  }
  return 0;
}

int gph_mcmc_get_state(gph_mcmc *m, double *logL, double *dataLogL, double *theta, double *popAge, double *migRate)
{
  if (!m) return GPH_EARG;
  if (logL) *logL = m->logLikelihood;
  if (dataLogL) *dataLogL = m->dataLogLikelihood;
  if (theta) for (int p = 0; p < m->K; p++) theta[p] = m->theta[p];
  if (popAge) for (int p = 0; p < m->K; p++) popAge[p] = m->popAge[p];
  if (migRate) for (int b = 0; b < m->B; b++) migRate[b] = m->migRate[b];
  return 0;
}

// canonical state dump, same text format as the oracle's (oracle/gphocs_oracle_io.c)
int gph_mcmc_dump_state(gph_mcmc *m, const char *path, int32_t withCond)
{
  if (!m || !path) return GPH_EARG;
  int rc;
  if ((rc = refresh_totals(m))) return rc;
  FILE *f = fopen(path, "w");
  if (!f) return GPH_EARG;
  fprintf(f, "STATE %lld\n", (long long)m->Ltot);
  fprintf(f, "MODEL");
  for (int p = 0; p < m->K; p++) fprintf(f, " %a %a %a", m->theta[p], m->popAge[p], m->sampleAge[p]);
  for (int b = 0; b < m->B; b++) fprintf(f, " %a %a %a", m->migRate[b], m->bandStart[b], m->bandEnd[b]);
  fprintf(f, "\n");
  fprintf(f, "GLOBAL %a %a %u %u %u\n", m->logLikelihood, m->dataLogLikelihood, m->gx, m->gy, m->gz);
  if (m->mutRateMode == 1) fprintf(f, "RATEVAR %a\n", m->rateVar);
  fprintf(f, "TOTALS");
  for (int p = 0; p < m->K; p++) fprintf(f, " %a %d", m->tot_coal[p], (int)m->tot_ncoal[p]);
  for (int b = 0; b < m->B; b++) fprintf(f, " %a %d", m->tot_mig[b], (int)m->tot_nmig[b]);
  fprintf(f, "\n");
  fclose(f);
  if ((rc = gph_engine_dump_loci(m->e, path, withCond, 1))) return rc;
  f = fopen(path, "a");
  fprintf(f, "ENDSTATE\n");
  fclose(f);
  return 0;
}

int gph_mcmc_param_vals(gph_mcmc *m, double *vals, int32_t n)
{
  if (!m || !vals) return GPH_EARG;
  for (int i = 0; i < n && i < (int)m->paramVals.size(); i++) vals[i] = m->paramVals[i];
  return 0;
}

int gph_mcmc_tau_accept_counts(gph_mcmc *m, int64_t *perPop)
{
  if (!m || !perPop) return GPH_EARG;
  for (int p = 0; p < m->K; p++) perPop[p] = m->accTau[p];
  return 0;
}

int gph_mcmc_set_finetunes(gph_mcmc *m, double coalTime, double migTime, double theta, double migRate, double mixing,
                           const double *taus)
{
  if (!m) return GPH_EARG;
  m->ftCoalTime = coalTime; m->ftMigTime = migTime; m->ftTheta = theta; m->ftMigRate = migRate; m->ftMixing = mixing;
  if (taus) for (int p = 0; p < m->K; p++) m->ftTaus[p] = taus[p];
  return 0;
}

int gph_mcmc_set_locus_rate_finetune(gph_mcmc *m, double locusRate)
{
  if (!m) return GPH_EARG;
  m->ftLocusRate = locusRate;
  return 0;
}

int gph_mcmc_locus_rate_state(gph_mcmc *m, int64_t *accepted, double *rateVar)
{
  if (!m) return GPH_EARG;
  if (accepted) *accepted = m->accLocusRate;
  if (rateVar) *rateVar = m->rateVar;
  return 0;
}

int gph_mcmc_set_log_period(gph_mcmc *m, int32_t iterations)
{
  if (!m || iterations <= 0) return GPH_EARG;
  m->samplesPerLog = iterations;
  return 0;
}

int gph_mcmc_accept_counts(gph_mcmc *m, int64_t *counts9)
{
  if (!m || !counts9) return GPH_EARG;
  for (int i = 0; i < 9; i++) counts9[i] = m->acc[i];
  counts9[8] = m->rubberband_conflicts;
  return 0;
}

} // extern "C"
