/* oracle/integration_binding.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference-side binding INTEGRATION.md describes, written out in full, COMPILED against the reference's own
 * headers and RUN: `make -C oracle boundary` (only where /root/reference exists) links this file with the reference's
 * own objects -- its main(), control-file and sequence readers, performMCMC with its trace writer, finetune search and
 * log lines, samplePopParameters / sampleMigRates, all unmodified -- into oracle/_ref/gphocs_boundary_{emu,hip}.  The
 * functions below REPLACE the bodies of the functions performMCMC calls (upstream src/GPhoCS.h:84-100, patch.h:258-260):
 * the reference objects are compiled from the sources where they lie, the replaced definitions are made weak in the
 * object files (objcopy --weaken-symbol), and these strong definitions win at link time.  Nothing of the reference's
 * per-locus path (LocusDataLikelihood.c pruning, patch.c event chains, the OpenMP loops of GPhoCS.c) runs: every
 * per-locus loop is an engine call.  tests/test_boundary_run.py runs the binary on the golden control files and
 * compares the trace file it writes with the one the unmodified reference binary wrote (tests/golden/*.trace).
 *
 * No reference source text is copied: this file #includes the reference's headers (and, like oracle/ref_harness.c,
 * LocusDataLikelihood.c for the struct the processed pattern table lives in) and calls its functions by name.  Every
 * replacement cites the upstream function it stands in for.
 *
 * What a replacement does: hand the process-wide globals the reference keeps on its main thread to the engine
 * (gph_chain_state: model parameters, the general RNG slot, dataState's accumulators, genetree_stats_total), make ONE
 * call of include/gphocs_hip.h's per-function boundary (gph_mcmc_update_*), copy the globals back, return what the
 * reference function returns.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "MultiCoreUtils.h"
#include "PopulationTree.h"
#include "GenericTree.h"
#include "MCMCcontrol.h"
#include "AlignmentProcessor.h"
#include "utils.h"
#include "LocusDataLikelihood.c"
#include "patch.h"
#include "GPhoCS.h"
#include "gphocs_hip.h"

extern RandGeneratorContext RndCtx;        /* as GPhoCS.c:32 declares it */
extern int gphocs_main(int argc, char *argv[]);   /* the reference's main(), compiled with -Dmain=gphocs_main */
static gph_engine *eng;
static gph_mcmc *mc;
static int thetaCalls;                     /* UpdateTheta runs once per iteration: performMCMC's `iteration` is -burnin + this */
static int64_t sweepAccepted[3], sweepMigNodes;

#define ITER() (-mcmcSetup.burnin + thetaCalls)
#define DIE(what, rc) do { fprintf(stderr, "boundary: %s failed with status %d\n", (what), (int)(rc)); exit(-1); } while (0)

/* ---- the globals of the reference's main thread <-> the engine's chain state */
static void push_chain(void)
{
  PopulationTree *pt = dataSetup.popTree;
  gph_chain_state c;
  int p, b;
  memset(&c, 0, sizeof c);
  for (p = 0; p < pt->numPops; p++) {
    c.theta[p] = pt->pops[p]->theta; c.popAge[p] = pt->pops[p]->age; c.sampleAge[p] = pt->pops[p]->sampleAge;
    c.coal_stats[p] = genetree_stats_total.coal_stats[p]; c.num_coals[p] = genetree_stats_total.num_coals[p];
  }
  for (b = 0; b < pt->numMigBands; b++) {
    c.migRate[b] = pt->migBands[b].migRate; c.bandStart[b] = pt->migBands[b].startTime; c.bandEnd[b] = pt->migBands[b].endTime;
    c.mig_stats[b] = genetree_stats_total.mig_stats[b]; c.num_migs[b] = genetree_stats_total.num_migs[b];
  }
  c.rng[0] = RndCtx.rndu_x[RAND_GENERAL_SLOT]; c.rng[1] = RndCtx.rndu_y[RAND_GENERAL_SLOT]; c.rng[2] = RndCtx.rndu_z[RAND_GENERAL_SLOT];
  c.logLikelihood = dataState.logLikelihood; c.dataLogLikelihood = dataState.dataLogLikelihood; c.rateVar = dataState.rateVar;
  c.rubberband_mig_conflicts = misc_stats.rubberband_mig_conflicts;
  if (gph_mcmc_set_chain(mc, &c)) DIE("gph_mcmc_set_chain", 1);
}
static void pull_chain(void)
{
  PopulationTree *pt = dataSetup.popTree;
  gph_chain_state c;
  int p, b;
  if (gph_mcmc_get_chain(mc, &c)) DIE("gph_mcmc_get_chain", 1);
  for (p = 0; p < pt->numPops; p++) {
    pt->pops[p]->theta = c.theta[p]; pt->pops[p]->age = c.popAge[p]; pt->pops[p]->sampleAge = c.sampleAge[p];
    genetree_stats_total.coal_stats[p] = c.coal_stats[p]; genetree_stats_total.num_coals[p] = (int)c.num_coals[p];
  }
  for (b = 0; b < pt->numMigBands; b++) {
    pt->migBands[b].migRate = c.migRate[b]; pt->migBands[b].startTime = c.bandStart[b]; pt->migBands[b].endTime = c.bandEnd[b];
    genetree_stats_total.mig_stats[b] = c.mig_stats[b]; genetree_stats_total.num_migs[b] = (int)c.num_migs[b];
  }
  RndCtx.rndu_x[RAND_GENERAL_SLOT] = c.rng[0]; RndCtx.rndu_y[RAND_GENERAL_SLOT] = c.rng[1]; RndCtx.rndu_z[RAND_GENERAL_SLOT] = c.rng[2];
  dataState.logLikelihood = c.logLikelihood; dataState.dataLogLikelihood = c.dataLogLikelihood; dataState.rateVar = c.rateVar;
  misc_stats.rubberband_mig_conflicts = (int)c.rubberband_mig_conflicts;
}

/* after processAlignments() + allocateAllMemory() (GPhoCS.c:205-235): hand the processed loci and the priors to the
 * engine once */
static int hip_startup(int device)
{
  PopulationTree *pt = dataSetup.popTree;
  static int32_t father[2 * NSPECIES - 1], son0[2 * NSPECIES - 1], son1[2 * NSPECIES - 1], bsrc[MAX_MIG_BANDS], btgt[MAX_MIG_BANDS], spp[NSPECIES];
  static double thA[2 * NSPECIES - 1], thB[2 * NSPECIES - 1], thS[2 * NSPECIES - 1], agA[2 * NSPECIES - 1], agB[2 * NSPECIES - 1], agS[2 * NSPECIES - 1];
  static double sage[2 * NSPECIES - 1], mrA[MAX_MIG_BANDS], mrB[MAX_MIG_BANDS], ftT[2 * NSPECIES - 1];
  static int32_t usa[2 * NSPECIES - 1];
  int64_t *offs, Ptot = 0;
  uint8_t *leaf;
  uint16_t *phases;
  int32_t *counts;
  double *rates;
  int g, p, b, l, rc;
  gph_config cfg;
  gph_mcmc_config mcc;
  for (p = 0; p < pt->numPops; p++) {
    father[p] = pt->pops[p]->father ? pt->pops[p]->father->id : -1;
    son0[p] = pt->pops[p]->sons[0] ? pt->pops[p]->sons[0]->id : -1;
    son1[p] = pt->pops[p]->sons[1] ? pt->pops[p]->sons[1]->id : -1;
    thA[p] = pt->pops[p]->thetaPrior.alpha; thB[p] = pt->pops[p]->thetaPrior.beta; thS[p] = pt->pops[p]->thetaPrior.sampleStart;
    agA[p] = pt->pops[p]->agePrior.alpha; agB[p] = pt->pops[p]->agePrior.beta; agS[p] = pt->pops[p]->agePrior.sampleStart;
    sage[p] = pt->pops[p]->sampleAge; usa[p] = pt->pops[p]->updateSampleAge;
    ftT[p] = mcmcSetup.finetunes.taus[p];
  }
  for (b = 0; b < pt->numMigBands; b++) {
    bsrc[b] = pt->migBands[b].sourcePop; btgt[b] = pt->migBands[b].targetPop;
    mrA[b] = pt->migBands[b].migRatePrior.alpha; mrB[b] = pt->migBands[b].migRatePrior.beta;
  }
  for (p = 0; p < pt->numCurPops; p++) spp[p] = dataSetup.numSamplesPerPop[p];
  memset(&cfg, 0, sizeof cfg);
  cfg.n = dataSetup.numSamples; cfg.Kc = pt->numCurPops; cfg.K = pt->numPops; cfg.B = pt->numMigBands; cfg.rootPop = pt->rootPop;
  cfg.samplesPerPop = spp; cfg.popFather = father; cfg.popSon0 = son0; cfg.popSon1 = son1; cfg.bandSrc = bsrc; cfg.bandTgt = btgt;
  cfg.device = device; cfg.L_total = dataSetup.numLoci; cfg.locus_begin = 0;
  if ((rc = gph_engine_create(&cfg, &eng))) return rc;
  /* the phased pattern table initializeLocusData() stored (LocusDataLikelihood.c:239-305): leaf conditionals are
   * one-hot (T, C, A, G) or all ones (N); numPhases non-zero on the first phase of each pattern */
  offs = (int64_t *)malloc(sizeof(int64_t) * (dataSetup.numLoci + 1));
  offs[0] = 0;
  for (g = 0; g < dataSetup.numLoci; g++) { Ptot += dataState.lociData[g]->seqData.numPatterns; offs[g + 1] = Ptot; }
  leaf = (uint8_t *)malloc((size_t)Ptot * cfg.n + 1);
  phases = (uint16_t *)malloc(sizeof(uint16_t) * (Ptot + 1));
  counts = (int32_t *)malloc(sizeof(int32_t) * (Ptot + 1));
  rates = (double *)malloc(sizeof(double) * dataSetup.numLoci);
  for (g = 0; g < dataSetup.numLoci; g++) {
    LocusData *ld = dataState.lociData[g];
    rates[g] = ld->mutationRate;
    for (p = 0; p < ld->seqData.numPatterns; p++) {
      for (l = 0; l < ld->numLeaves; l++) {
        const double *cp = ld->nodeArray[l]->conditionalProbs + 4 * p;
        leaf[(size_t)(offs[g] + p) * cfg.n + l] = (cp[0] + cp[1] + cp[2] + cp[3] >= 4) ? 4 : cp[0] == 1.0 ? 0 : cp[1] == 1.0 ? 1 : cp[2] == 1.0 ? 2 : 3;
      }
      phases[offs[g] + p] = (uint16_t)ld->seqData.numPhases[p];
      counts[offs[g] + p] = ld->seqData.patternCount[p];
    }
  }
  rc = gph_engine_load_loci(eng, dataSetup.numLoci, offs, leaf, phases, counts, mcmcSetup.mutRateMode == 2 ? rates : NULL);
  free(offs); free(leaf); free(phases); free(counts); free(rates);
  if (rc) return rc;
  memset(&mcc, 0, sizeof mcc);
  mcc.thetaAlpha = thA; mcc.thetaBeta = thB; mcc.thetaStart = thS; mcc.ageAlpha = agA; mcc.ageBeta = agB; mcc.ageStart = agS;
  mcc.sampleAge = sage; mcc.updateSampleAge = usa; mcc.mrAlpha = mrA; mcc.mrBeta = mrB;
  mcc.ftCoalTime = mcmcSetup.finetunes.coalTime; mcc.ftMigTime = mcmcSetup.finetunes.migTime; mcc.ftTheta = mcmcSetup.finetunes.theta;
  mcc.ftMigRate = mcmcSetup.finetunes.migRate; mcc.ftMixing = mcmcSetup.finetunes.mixing; mcc.ftTaus = ftT;
  mcc.seed = mcmcSetup.randomSeed; mcc.startMig = mcmcSetup.startMig; mcc.doMixing = mcmcSetup.doMixing;
  mcc.samplesPerLog = ioSetup.samplesPerLog; mcc.numParameters = mcmcSetup.numParameters; mcc.printFactors = mcmcSetup.printFactors;
  mcc.mutRateMode = mcmcSetup.mutRateMode; mcc.varRatesAlpha = mcmcSetup.varRatesAlpha; mcc.ftLocusRate = mcmcSetup.finetunes.locusRate;
  return gph_mcmc_create(eng, &cfg, &mcc, &mc);
}

/* initializeMCMC, GPhoCS.c:1122-1225.  Kept from upstream: samplePopParameters (the reference's own function, on the
 * reference's general RNG slot) and the constant / fixed locus rates; the per-locus loop :1197-1214 (GetRandomGtree,
 * constructEventChain, computeGenetreeStats, gtreeLnLikelihood, computeLocusDataLikelihood) and computeTotalStats are
 * the engine's */
int initializeMCMC(void)
{
  int gen, rc;
  const char *dev = getenv("GPH_DEVICE");
  samplePopParameters(dataSetup.popTree);
  if (mcmcSetup.mutRateMode == 0) {
    dataState.rateVar = 0.0;
  } else if (mcmcSetup.mutRateMode == 2) {
    if (0 != readRateFile(ioSetup.rateFileName)) { fprintf(stderr, "Error: Unable to reading rate file '%s'. Aborting !!\n", ioSetup.rateFileName); return -1; }
  }
  if ((rc = hip_startup(dev ? atoi(dev) : 0))) DIE("engine start-up", rc);
  if (mcmcSetup.mutRateMode == 1) {
    /* locus-mut-rate VAR (GPhoCS.c:1157-1178): 0.8 + 0.4 u from every locus's own stream, normalised to mean 1; the
     * engine gets the rates and is told that every locus stream has spent one draw */
    double *rates = (double *)malloc(sizeof(double) * dataSetup.numLoci), total = 0.0;
    for (gen = 0; gen < dataSetup.numLoci; gen++) { rates[gen] = 0.8 + 0.4 * rndu(gen); total += rates[gen]; }
    total /= dataSetup.numLoci;
    dataState.rateVar = 0.0;
    for (gen = 0; gen < dataSetup.numLoci; gen++) { rates[gen] = rates[gen] / total; dataState.rateVar += (rates[gen] - 1) * (rates[gen] - 1); }
    dataState.rateVar /= dataSetup.numLoci;
    mcmcSetup.genRateRef = 0;
    if ((rc = gph_engine_set_locus_rates(eng, rates, 1, 1))) DIE("gph_engine_set_locus_rates", rc);
    free(rates);
  }
  dataState.logLikelihood = 0.0;
  dataState.dataLogLikelihood = 0.0;
  for (gen = 0; gen < dataSetup.numLoci; gen++) locus_data[gen].genLogLikelihood = 0.0;   /* performMCMC reads them at start-mig (:1751) */
  push_chain();
  if ((rc = gph_mcmc_initialize_genealogies(mc))) DIE("gph_mcmc_initialize_genealogies", rc);
  pull_chain();
  thetaCalls = 0;
  return dataSetup.numLoci * (dataSetup.numSamples - 1);
}

/* UpdateGB_InternalNode (GPhoCS.c:2287), UpdateGB_MigrationNode (:2439), UpdateGB_MigSPR (:2598): performMCMC calls them
 * back to back (:1495-1538) and the engine runs the three sweeps of a locus in ONE launch, so the first does the work
 * and the other two report their share.  genetree_stats_total.num_migs as performMCMC sums it right after the
 * migration-node sweep (:1517-1520) is the count after THAT sweep, before the SPR changes it */
int UpdateGB_InternalNode(double finetune)
{
  int rc;
  push_chain();
  if ((rc = gph_mcmc_update_gb(mc, ITER(), finetune, mcmcSetup.finetunes.migTime, sweepAccepted, &sweepMigNodes))) DIE("gph_mcmc_update_gb", rc);
  pull_chain();
  return (int)sweepAccepted[0];
}
static int savedNumMigs[MAX_MIG_BANDS];
int UpdateGB_MigrationNode(double finetune)
{
  int b;
  (void)finetune;
  for (b = 0; b < dataSetup.popTree->numMigBands; b++) { savedNumMigs[b] = genetree_stats_total.num_migs[b]; genetree_stats_total.num_migs[b] = 0; }
  if (dataSetup.popTree->numMigBands > 0) genetree_stats_total.num_migs[0] = (int)sweepMigNodes;
  return (int)sweepAccepted[1];
}
int UpdateGB_MigSPR(void)
{
  int b;
  for (b = 0; b < dataSetup.popTree->numMigBands; b++) genetree_stats_total.num_migs[b] = savedNumMigs[b];
  return (int)sweepAccepted[2];
}

/* UpdateLocusRate, GPhoCS.c:4598 */
int UpdateLocusRate(double finetune)
{
  int64_t acc = 0;
  int rc;
  push_chain();
  if ((rc = gph_mcmc_update_locus_rate(mc, ITER(), finetune, &acc))) DIE("gph_mcmc_update_locus_rate", rc);
  pull_chain();
  return (int)acc;
}

/* UpdateTheta, GPhoCS.c:3037 */
int UpdateTheta(double finetune)
{
  int64_t acc = 0;
  int rc;
  push_chain();
  if ((rc = gph_mcmc_update_theta(mc, ITER(), finetune, &acc))) DIE("gph_mcmc_update_theta", rc);
  pull_chain();
  thetaCalls++;
  return (int)acc;
}

/* UpdateMigRates, GPhoCS.c:3115 (performMCMC calls it once iteration > start-mig, :1596; UpdateTheta has already
 * counted this iteration) */
int UpdateMigRates(double finetune)
{
  int64_t acc = 0;
  int rc;
  push_chain();
  if ((rc = gph_mcmc_update_mig_rates(mc, ITER() - 1, finetune, &acc))) DIE("gph_mcmc_update_mig_rates", rc);
  pull_chain();
  return (int)acc;
}

/* UpdateTau, GPhoCS.c:3224: accepted[] is zeroed and filled for the ancestral populations only (:3255) */
void UpdateTau(double *finetunes, int *accepted)
{
  int32_t acc[2 * NSPECIES - 1];
  int p, rc;
  push_chain();
  if ((rc = gph_mcmc_update_tau(mc, ITER() - 1, finetunes, acc))) DIE("gph_mcmc_update_tau", rc);
  pull_chain();
  for (p = dataSetup.popTree->numCurPops; p < dataSetup.popTree->numPops; p++) accepted[p] = acc[p];
}

/* UpdateSampleAge, GPhoCS.c:4006: accepted[] of the current populations (:4027) */
void UpdateSampleAge(double *finetunes, int *accepted)
{
  int32_t acc[2 * NSPECIES - 1];
  int p, rc;
  push_chain();
  if ((rc = gph_mcmc_update_sample_age(mc, ITER() - 1, finetunes, acc))) DIE("gph_mcmc_update_sample_age", rc);
  pull_chain();
  for (p = 0; p < dataSetup.popTree->numCurPops; p++) accepted[p] = acc[p];
}

/* mixing, GPhoCS.c:4688 */
int mixing(double finetune)
{
  int64_t acc = 0;
  int rc;
  push_chain();
  if ((rc = gph_mcmc_mixing(mc, ITER() - 1, finetune, &acc))) DIE("gph_mcmc_mixing", rc);
  pull_chain();
  return (int)acc;
}

/* synchronizeEvents, patch.c:3548: performMCMC calls it for every locus in turn (GPhoCS.c:1705-1714); the engine's pass
 * covers all loci (and is deferred into the head of the next sweep kernel), so the first call asks for it and every
 * call reports success -- an inconsistency surfaces as Fatal Error 0075/0076 from the engine */
int synchronizeEvents(int gen)
{
  int rc;
  if (gen == 0) {
    push_chain();
    if ((rc = gph_mcmc_synchronize_events(mc, ITER() - 1, 0))) DIE("gph_mcmc_synchronize_events", rc);
    pull_chain();
  }
  return 1;
}

/* gtreeLnLikelihood, patch.c:2702: with every proposal function replaced, the one caller left is performMCMC's start-mig
 * block (GPhoCS.c:1749-1757) -- after sampleMigRates it subtracts every locus's old genLogLikelihood from the average
 * and adds the recomputed one.  The engine does that for all loci in one pass (old and new sums come back from the
 * same kernel); the per-locus slots of the reference stay 0 and locus 0 hands over the whole change */
double gtreeLnLikelihood(int gen)
{
  double before, after;
  int rc;
  if (gen != 0) return 0.0;
  before = dataState.logLikelihood;
  push_chain();                              /* the freshly sampled migration rates and the RNG state behind them */
  if ((rc = gph_mcmc_synchronize_events(mc, ITER() - 1, 1))) DIE("gph_mcmc_synchronize_events(refresh)", rc);
  pull_chain();
  after = dataState.logLikelihood;
  dataState.logLikelihood = before;          /* performMCMC adds the return value / numLoci itself */
  locus_data[0].genLogLikelihood = 0.0;
  return (after - before) * dataSetup.numLoci;
}

/* checkAll, patch.c:2745: consistency checks + the accumulator resynchronisation that is part of the trajectory */
int checkAll(void)
{
  int32_t ok = 0;
  push_chain();
  (void)gph_mcmc_check_all(mc, ITER() - 1, &ok);
  pull_chain();
  locus_data[0].genLogLikelihood = 0.0;
  return ok;
}

int main(int argc, char *argv[]) { return gphocs_main(argc, argv); }
