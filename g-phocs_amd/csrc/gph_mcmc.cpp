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
#include "gph_global.h"

// the engine owns the chain state (host mirror + the copy in HBM the kernels and the k_global stages work on)
extern "C" GphGlobal *gph_engine_global_(gph_engine *e);             /* mutable: marks the mirror for upload */
extern "C" const GphGlobal *gph_engine_global_ro_(gph_engine *e);
extern "C" int gph_engine_iteration_(gph_engine *e, int32_t iteration, const double *lr_alpha_finetune, int64_t *lr_accepted, double *lr_rateVar);
extern "C" int gph_engine_part_(gph_engine *e, int32_t part, int32_t iteration, const double *lr_alpha_finetune, int64_t *lr_accepted, double *lr_rateVar);

struct gph_mcmc {
  gph_engine *e;
  int n, Kc, K, B;
  int64_t Ltot;
  std::vector<double> printFactors, paramVals;
  int mutRateMode;              // 0 CONST, 1 VAR (UpdateLocusRate is live), 2 FIXED
  double varRatesAlpha, ftLocusRate, rateVar;
  int64_t accLocusRate;
  int seed, numParameters;
  FILE *rec;
};

static const GphGlobal &GG(const gph_mcmc *m) { return *gph_engine_global_ro_(m->e); }

// the record lines of the stages of one iteration ("IT <iter> <proposal> <accepted> <dataLnL %a> <logL %a>")
static void print_records(gph_mcmc *m, int it)
{
  if (!m->rec) return;
  const GphGlobal &G = GG(m);
  for (int i = 0; i < G.nrec; i++) {
    const GphRec &r = G.rec[i];
    char nm[32];
    switch (r.code) {
    case REC_INIT: snprintf(nm, sizeof nm, "INIT"); break;
    case REC_INT: snprintf(nm, sizeof nm, "INT"); break;
    case REC_MIGN: snprintf(nm, sizeof nm, "MIGN"); break;
    case REC_SPR: snprintf(nm, sizeof nm, "SPR"); break;
    case REC_LRATE: snprintf(nm, sizeof nm, "LRATE"); break;
    case REC_THETA: snprintf(nm, sizeof nm, "THETA"); break;
    case REC_MIGR: snprintf(nm, sizeof nm, "MIGR"); break;
    case REC_TAU: snprintf(nm, sizeof nm, "TAU%d", r.idx); break;
    case REC_SAGE: snprintf(nm, sizeof nm, "SAGE%d", r.idx); break;
    case REC_MIX: snprintf(nm, sizeof nm, "MIX"); break;
    case REC_CHECK: snprintf(nm, sizeof nm, "CHECK"); break;
    case REC_CONFLICTS: fprintf(m->rec, "CONFLICTS %lld\n", (long long)r.acc); continue;
    default: continue;
    }
    fprintf(m->rec, "IT %d %s %ld %a %a\n", it, nm, (long)r.acc, r.dataLnL, r.logL);
  }
}

// recordParamVals, GPhoCS.c:802-849
static void record_param_vals(gph_mcmc *m)
{
  const GphGlobal &G = GG(m);
  int ind = 0;
  m->paramVals.assign(m->numParameters + 4, 0.0);
  for (int pop = 0; pop < m->K; pop++) m->paramVals[ind++] = G.model.theta[pop];
  for (int pop = m->Kc; pop < m->K; pop++) m->paramVals[ind++] = G.model.popAge[pop];
  for (int b = 0; b < m->B; b++) m->paramVals[ind++] = G.shownValid ? G.migRateShown[b] : G.model.migRate[b];
  for (int pop = 0; pop < m->Kc; pop++)
    if (G.updateSampleAge[pop] || G.model.sampleAge[pop] > 0.0) m->paramVals[ind++] = G.model.sampleAge[pop];
  if (m->mutRateMode == 1) m->paramVals[ind++] = sqrt(m->rateVar);   /* GPhoCS.c:842-847 */
}

extern "C" {

int gph_mcmc_create(gph_engine *e, const gph_config *cfg, const gph_mcmc_config *mc, gph_mcmc **out)
{
  if (!e || !cfg || !mc || !out) return GPH_EARG;
  gph_mcmc *m = new gph_mcmc();
  m->e = e;
  m->n = cfg->n; m->Kc = cfg->Kc; m->K = cfg->K; m->B = cfg->B; m->Ltot = cfg->L_total;
  GphGlobal &G = *gph_engine_global_(e);
  for (int p = 0; p < m->K; p++) {
    G.model.sampleAge[p] = mc->sampleAge[p];
    G.updateSampleAge[p] = (p < m->Kc && mc->updateSampleAge) ? (mc->updateSampleAge[p] != 0) : 0;
    G.thetaAlpha[p] = mc->thetaAlpha[p]; G.thetaBeta[p] = mc->thetaBeta[p]; G.thetaStart[p] = mc->thetaStart[p];
    G.ageAlpha[p] = mc->ageAlpha[p]; G.ageBeta[p] = mc->ageBeta[p]; G.ageStart[p] = mc->ageStart[p];
    G.ftTaus[p] = mc->ftTaus[p];
  }
  for (int b = 0; b < GPH_MAXB; b++) { G.mrAlpha[b] = 0.0; G.mrBeta[b] = 1.0; }
  for (int b = 0; b < m->B; b++) { G.mrAlpha[b] = mc->mrAlpha[b]; G.mrBeta[b] = mc->mrBeta[b]; }
  G.ftCoalTime = mc->ftCoalTime; G.ftMigTime = mc->ftMigTime; G.ftTheta = mc->ftTheta;
  G.ftMigRate = mc->ftMigRate; G.ftMixing = mc->ftMixing;
  G.startMig = mc->startMig; G.doMixing = mc->doMixing; G.samplesPerLog = mc->samplesPerLog;
  G.gx = 11; G.gy = 23; G.gz = 170u * ((uint32_t)mc->seed % 178u) + 137u;   // utils.c:421-426
  G.logLikelihood = G.dataLogLikelihood = 0.0;
  G.rubberband_conflicts = 0;
  memset(G.acc, 0, sizeof G.acc);
  memset(G.accTau, 0, sizeof G.accTau);
  m->mutRateMode = mc->mutRateMode; m->varRatesAlpha = mc->varRatesAlpha; m->ftLocusRate = mc->ftLocusRate;
  m->rateVar = 0.0; m->accLocusRate = 0;
  m->seed = mc->seed;
  m->numParameters = mc->numParameters;
  m->printFactors.assign(mc->printFactors, mc->printFactors + mc->numParameters);
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
  if (!m) return GPH_EARG;
  gg_sample_pop_parameters(*gph_engine_global_(m->e));
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
  { /* the engine's own bookkeeping of "a model has been set" */
    const GphGlobal &G = GG(m);
    if ((rc = gph_engine_set_model(m->e, G.model.theta, G.model.popAge, G.model.sampleAge, G.model.migRate, G.model.bandStart, G.model.bandEnd))) return rc;
  }
  if ((rc = gph_engine_seed(m->e, (uint32_t)m->seed))) return rc;
  if ((rc = gph_engine_init_genealogies(m->e, nullptr, nullptr))) return rc;
  if (totalCoals) *totalCoals = m->Ltot * (m->n - 1);
  print_records(m, -1);
  return 0;
}

int gph_mcmc_iteration(gph_mcmc *m, int32_t iteration)
{
  if (!m) return GPH_EARG;
  const double lr[2] = {m->varRatesAlpha, m->ftLocusRate};
  int rc = gph_engine_iteration_(m->e, iteration, m->mutRateMode == 1 ? lr : nullptr, &m->accLocusRate, &m->rateVar);
  if (rc) return rc;
  record_param_vals(m);
  print_records(m, iteration);
  if (m->rec) {
    // trace row, GPhoCS.c:746-754, 1763-1769
    const GphGlobal &G = GG(m);
    fprintf(m->rec, "TRACE %d\t", iteration);
    for (int i = 0; i < m->numParameters; i++) fprintf(m->rec, "%8.5f\t", m->paramVals[i] * m->printFactors[i]);
    fprintf(m->rec, "\t%.6f\t%.6f\n", G.logLikelihood, G.dataLogLikelihood);
    fflush(m->rec);
  }
  return 0;
}

// ---------------------------------------------------------------- the reference's functions, one call each
// (include/gphocs_hip.h: the boundary at the granularity performMCMC calls them, GPhoCS.h:84-100)
int gph_mcmc_get_chain(gph_mcmc *m, gph_chain_state *o)
{
  if (!m || !o) return GPH_EARG;
  const GphGlobal &G = GG(m);
  memset(o, 0, sizeof *o);
  for (int p = 0; p < m->K; p++) {
    o->theta[p] = G.model.theta[p]; o->popAge[p] = G.model.popAge[p]; o->sampleAge[p] = G.model.sampleAge[p];
    o->coal_stats[p] = G.tot_coal[p]; o->num_coals[p] = G.tot_ncoal[p];
  }
  for (int b = 0; b < m->B; b++) {
    o->migRate[b] = G.model.migRate[b]; o->bandStart[b] = G.model.bandStart[b]; o->bandEnd[b] = G.model.bandEnd[b];
    o->mig_stats[b] = G.tot_mig[b]; o->num_migs[b] = G.tot_nmig[b];
  }
  o->rng[0] = G.gx; o->rng[1] = G.gy; o->rng[2] = G.gz;
  o->logLikelihood = G.logLikelihood; o->dataLogLikelihood = G.dataLogLikelihood; o->rateVar = m->rateVar;
  o->rubberband_mig_conflicts = G.rubberband_conflicts;
  return 0;
}
int gph_mcmc_set_chain(gph_mcmc *m, const gph_chain_state *in)
{
  if (!m || !in) return GPH_EARG;
  GphGlobal &G = *gph_engine_global_(m->e);
  for (int p = 0; p < m->K; p++) {
    if (G.model.theta[p] != in->theta[p]) gg_set_theta(G, p, in->theta[p]);
    G.model.popAge[p] = in->popAge[p]; G.model.sampleAge[p] = in->sampleAge[p];
    G.tot_coal[p] = in->coal_stats[p]; G.tot_ncoal[p] = in->num_coals[p];
  }
  for (int b = 0; b < m->B; b++) {
    if (G.model.migRate[b] != in->migRate[b]) gg_set_mig(G, b, in->migRate[b]);
    G.model.bandStart[b] = in->bandStart[b]; G.model.bandEnd[b] = in->bandEnd[b];
    G.tot_mig[b] = in->mig_stats[b]; G.tot_nmig[b] = in->num_migs[b];
  }
  G.gx = in->rng[0]; G.gy = in->rng[1]; G.gz = in->rng[2];
  G.logLikelihood = in->logLikelihood; G.dataLogLikelihood = in->dataLogLikelihood; m->rateVar = in->rateVar;
  G.rubberband_conflicts = in->rubberband_mig_conflicts;
  return 0;
}
static int run_part(gph_mcmc *m, int part, int32_t iteration)
{
  const double lr[2] = {m->varRatesAlpha, m->ftLocusRate};
  gph_engine_global_(m->e)->nrec = 0;
  int rc = gph_engine_part_(m->e, part, iteration, m->mutRateMode == 1 ? lr : nullptr, &m->accLocusRate, &m->rateVar);
  if (!rc) print_records(m, iteration);
  return rc;
}
int gph_mcmc_initialize_genealogies(gph_mcmc *m)
{
  if (!m) return GPH_EARG;
  int rc;
  const GphGlobal &G = GG(m);
  if ((rc = gph_engine_set_model(m->e, G.model.theta, G.model.popAge, G.model.sampleAge, G.model.migRate, G.model.bandStart, G.model.bandEnd))) return rc;
  if ((rc = gph_engine_seed(m->e, (uint32_t)m->seed))) return rc;
  if ((rc = gph_engine_init_genealogies(m->e, nullptr, nullptr))) return rc;
  print_records(m, -1);
  return 0;
}
int gph_mcmc_update_gb(gph_mcmc *m, int32_t iteration, double ftCoalTime, double ftMigTime, int64_t accepted[3], int64_t *total_mig_nodes)
{
  if (!m || !accepted) return GPH_EARG;
  GphGlobal &G = *gph_engine_global_(m->e);
  G.ftCoalTime = ftCoalTime; G.ftMigTime = ftMigTime;
  const int64_t a0 = G.acc[0], a1 = G.acc[1], a2 = G.acc[2], a7 = G.acc[7];
  int rc = run_part(m, GPH_PART_SWEEP, iteration);
  if (rc) return rc;
  const GphGlobal &H = GG(m);
  accepted[0] = H.acc[0] - a0; accepted[1] = H.acc[1] - a1; accepted[2] = H.acc[2] - a2;
  if (total_mig_nodes) *total_mig_nodes = H.acc[7] - a7;
  return 0;
}
int gph_mcmc_update_locus_rate(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted)
{
  if (!m || !accepted) return GPH_EARG;
  *accepted = 0;
  if (m->mutRateMode != 1) return 0;
  m->ftLocusRate = finetune;
  const int64_t a = m->accLocusRate;
  int rc = run_part(m, GPH_PART_LRATE, iteration);
  if (!rc) *accepted = m->accLocusRate - a;
  return rc;
}
static int one_count(gph_mcmc *m, int part, int slot, int32_t iteration, int64_t *accepted)
{
  const int64_t a = GG(m).acc[slot];
  int rc = run_part(m, part, iteration);
  if (!rc) *accepted = GG(m).acc[slot] - a;
  return rc;
}
int gph_mcmc_update_theta(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted)
{
  if (!m || !accepted) return GPH_EARG;
  gph_engine_global_(m->e)->ftTheta = finetune;
  return one_count(m, GPH_PART_THETA, 3, iteration, accepted);
}
int gph_mcmc_update_mig_rates(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted)
{
  if (!m || !accepted) return GPH_EARG;
  gph_engine_global_(m->e)->ftMigRate = finetune;
  return one_count(m, GPH_PART_MIGR, 4, iteration, accepted);
}
int gph_mcmc_update_tau(gph_mcmc *m, int32_t iteration, const double *finetunes, int32_t *accepted)
{
  if (!m || !finetunes || !accepted) return GPH_EARG;
  GphGlobal &G = *gph_engine_global_(m->e);
  for (int p = m->Kc; p < m->K; p++) G.ftTaus[p] = finetunes[p];
  int rc = run_part(m, GPH_PART_TAU, iteration);
  if (rc) return rc;
  for (int p = m->Kc; p < m->K; p++) accepted[p] = GG(m).accArr[p];   /* upstream zeroes and fills the ancestral entries only (GPhoCS.c:3255) */
  return 0;
}
int gph_mcmc_update_sample_age(gph_mcmc *m, int32_t iteration, const double *finetunes, int32_t *accepted)
{
  if (!m || !finetunes || !accepted) return GPH_EARG;
  GphGlobal &G = *gph_engine_global_(m->e);
  for (int p = 0; p < m->Kc; p++) G.ftTaus[p] = finetunes[p];
  int rc = run_part(m, GPH_PART_SAGE, iteration);
  if (rc) return rc;
  for (int p = 0; p < m->Kc; p++) accepted[p] = GG(m).accArr[p];      /* the current populations' entries (GPhoCS.c:4027) */
  return 0;
}
int gph_mcmc_mixing(gph_mcmc *m, int32_t iteration, double finetune, int64_t *accepted)
{
  if (!m || !accepted) return GPH_EARG;
  gph_engine_global_(m->e)->ftMixing = finetune;
  return one_count(m, GPH_PART_MIX, 6, iteration, accepted);
}
int gph_mcmc_synchronize_events(gph_mcmc *m, int32_t iteration, int32_t refresh)
{
  if (!m) return GPH_EARG;
  return run_part(m, refresh ? GPH_PART_REFRESH : GPH_PART_SYNC, iteration);
}
int gph_mcmc_check_all(gph_mcmc *m, int32_t iteration, int32_t *ok)
{
  if (!m) return GPH_EARG;
  int rc = run_part(m, GPH_PART_CHECK, iteration);
  if (ok) *ok = rc == 0;
  return rc;
}

int gph_mcmc_get_state(gph_mcmc *m, double *logL, double *dataLogL, double *theta, double *popAge, double *migRate)
{
  if (!m) return GPH_EARG;
  const GphGlobal &G = GG(m);
  if (logL) *logL = G.logLikelihood;
  if (dataLogL) *dataLogL = G.dataLogLikelihood;
  if (theta) for (int p = 0; p < m->K; p++) theta[p] = G.model.theta[p];
  if (popAge) for (int p = 0; p < m->K; p++) popAge[p] = G.model.popAge[p];
  if (migRate) for (int b = 0; b < m->B; b++) migRate[b] = G.model.migRate[b];
  return 0;
}

// canonical state dump, same text format as the oracle's (oracle/gphocs_oracle_io.c)
int gph_mcmc_dump_state(gph_mcmc *m, const char *path, int32_t withCond)
{
  if (!m || !path) return GPH_EARG;
  int rc;
  if ((rc = gph_engine_get_totals(m->e, nullptr, nullptr, nullptr, nullptr))) return rc;   /* refreshes the totals of the chain state */
  const GphGlobal &G = GG(m);
  FILE *f = fopen(path, "w");
  if (!f) return GPH_EARG;
  fprintf(f, "STATE %lld\n", (long long)m->Ltot);
  fprintf(f, "MODEL");
  for (int p = 0; p < m->K; p++) fprintf(f, " %a %a %a", G.model.theta[p], G.model.popAge[p], G.model.sampleAge[p]);
  for (int b = 0; b < m->B; b++) fprintf(f, " %a %a %a", G.model.migRate[b], G.model.bandStart[b], G.model.bandEnd[b]);
  fprintf(f, "\n");
  fprintf(f, "GLOBAL %a %a %u %u %u\n", G.logLikelihood, G.dataLogLikelihood, G.gx, G.gy, G.gz);
  if (m->mutRateMode == 1) fprintf(f, "RATEVAR %a\n", m->rateVar);
  fprintf(f, "TOTALS");
  for (int p = 0; p < m->K; p++) fprintf(f, " %a %d", G.tot_coal[p], (int)G.tot_ncoal[p]);
  for (int b = 0; b < m->B; b++) fprintf(f, " %a %d", G.tot_mig[b], (int)G.tot_nmig[b]);
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
  for (int p = 0; p < m->K; p++) perPop[p] = GG(m).accTau[p];
  return 0;
}

int gph_mcmc_set_finetunes(gph_mcmc *m, double coalTime, double migTime, double theta, double migRate, double mixing,
                           const double *taus)
{
  if (!m) return GPH_EARG;
  GphGlobal &G = *gph_engine_global_(m->e);
  G.ftCoalTime = coalTime; G.ftMigTime = migTime; G.ftTheta = theta; G.ftMigRate = migRate; G.ftMixing = mixing;
  if (taus) for (int p = 0; p < m->K; p++) G.ftTaus[p] = taus[p];
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
  gph_engine_global_(m->e)->samplesPerLog = iterations;
  return 0;
}

int gph_mcmc_accept_counts(gph_mcmc *m, int64_t *counts9)
{
  if (!m || !counts9) return GPH_EARG;
  for (int i = 0; i < 9; i++) counts9[i] = GG(m).acc[i];
  counts9[8] = GG(m).rubberband_conflicts;
  return 0;
}

} // extern "C"
