/* oracle/integration_binding.c -- TEST INFRASTRUCTURE ONLY (compile check, never linked into the product).
 *
 * The reference-side binding INTEGRATION.md describes, written out in full and COMPILED against the reference's own
 * headers (oracle/Makefile target `binding`, only where /root/reference exists): a maintainer of G-PhoCS would add
 * this file, call hip_startup() after allocateAllMemory() (GPhoCS.c:224) and replace the bodies of the proposal
 * functions by the hip_* functions below.  No reference source text is copied: the file #includes the reference's
 * headers (and, like oracle/ref_harness.c, LocusDataLikelihood.c for the struct the pattern table lives in) and calls
 * its functions by name.  Every hip_* function cites the upstream function whose per-locus loop it replaces.
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
static gph_engine *eng;

/* the model tables the kernels read (theta, tau, sample ages, rates, band times): after every change of a parameter */
static void hip_push_model(void)
{
  PopulationTree *pt = dataSetup.popTree;
  double theta[2 * NSPECIES - 1], age[2 * NSPECIES - 1], sage[2 * NSPECIES - 1];
  double rate[MAX_MIG_BANDS], start[MAX_MIG_BANDS], end[MAX_MIG_BANDS];
  int p, b;
  for (p = 0; p < pt->numPops; p++) { theta[p] = pt->pops[p]->theta; age[p] = pt->pops[p]->age; sage[p] = pt->pops[p]->sampleAge; }
  for (b = 0; b < pt->numMigBands; b++) { rate[b] = pt->migBands[b].migRate; start[b] = pt->migBands[b].startTime; end[b] = pt->migBands[b].endTime; }
  gph_engine_set_model(eng, theta, age, sage, rate, start, end);
}

/* after processAlignments() + allocateAllMemory() (GPhoCS.c:205-235): hand the processed loci to the engine once */
int hip_startup(int device)
{
  PopulationTree *pt = dataSetup.popTree;
  int32_t father[2 * NSPECIES - 1], son0[2 * NSPECIES - 1], son1[2 * NSPECIES - 1], bsrc[MAX_MIG_BANDS], btgt[MAX_MIG_BANDS], spp[NSPECIES];
  int64_t *offs, Ptot = 0;
  uint8_t *leaf;
  uint16_t *phases;
  int32_t *counts;
  double *rates;
  int g, p, b, l, rc;
  gph_config cfg;
  for (p = 0; p < pt->numPops; p++) {
    father[p] = pt->pops[p]->father ? pt->pops[p]->father->id : -1;
    son0[p] = pt->pops[p]->sons[0] ? pt->pops[p]->sons[0]->id : -1;
    son1[p] = pt->pops[p]->sons[1] ? pt->pops[p]->sons[1]->id : -1;
  }
  for (b = 0; b < pt->numMigBands; b++) { bsrc[b] = pt->migBands[b].sourcePop; btgt[b] = pt->migBands[b].targetPop; }
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
  leaf = (uint8_t *)malloc((size_t)Ptot * cfg.n);
  phases = (uint16_t *)malloc(sizeof(uint16_t) * Ptot);
  counts = (int32_t *)malloc(sizeof(int32_t) * Ptot);
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
  return gph_engine_seed(eng, (uint32_t)mcmcSetup.randomSeed);                 /* initRandomGenerator, utils.c:411 */
}

/* initializeMCMC (GPhoCS.c:1122): after samplePopParameters(), instead of the per-locus loop :1197-1214 */
int hip_initializeMCMC(void)
{
  double sumGen, sumData;
  int rc;
  hip_push_model();
  if ((rc = gph_engine_init_genealogies(eng, &sumGen, &sumData))) return rc;
  dataState.dataLogLikelihood = sumData;
  dataState.logLikelihood = (sumGen + sumData) / dataSetup.numLoci;
  return 0;
}

/* UpdateGB_InternalNode + UpdateGB_MigrationNode + UpdateGB_MigSPR (GPhoCS.c:2287, 2439, 2598; called back to back
 * at :1495-1538): one fused launch; accepted[] = the three functions' return values */
int hip_UpdateGB(int accepted[3])
{
  gph_sweep_result r;
  double nc[2 * NSPECIES - 1], nm[MAX_MIG_BANDS];
  int rc, p, b;
  if ((rc = gph_engine_genealogy_sweep(eng, 7, mcmcSetup.finetunes.coalTime, mcmcSetup.finetunes.migTime, &r))) return rc;
  dataState.dataLogLikelihood += r.dData_internal + r.dData_spr;
  dataState.logLikelihood += r.dLog_internal + r.dLog_mignode + r.dLog_spr;
  accepted[0] = (int)r.accepted_internal; accepted[1] = (int)r.accepted_mignode; accepted[2] = (int)r.accepted_spr;
  /* computeTotalStats (patch.c:2134) for UpdateTheta / UpdateMigRates / mixing */
  if ((rc = gph_engine_get_totals(eng, genetree_stats_total.coal_stats, nc, genetree_stats_total.mig_stats, nm))) return rc;
  for (p = 0; p < dataSetup.popTree->numPops; p++) genetree_stats_total.num_coals[p] = (int)nc[p];
  for (b = 0; b < dataSetup.popTree->numMigBands; b++) genetree_stats_total.num_migs[b] = (int)nm[b];
  return 0;
}

/* UpdateTheta accepted branch (GPhoCS.c:3084-3093) / UpdateMigRates accepted branch (:3192-3200) */
int hip_apply_theta(int pop, double lnc, double thetaold, double thetanew) { return gph_engine_apply_theta(eng, pop, lnc, thetaold, thetanew); }
int hip_apply_migrate(int band, double lnc, double oldrate, double newrate) { return gph_engine_apply_migrate(eng, band, lnc, oldrate, newrate); }

/* UpdateTau (GPhoCS.c:3224): upstream keeps :3256-3461 (bounds, proposal, affected bands -> a) and the decision
 * :3835-3858; loops 1, 2 and 3/4 (:3491-3833, :3885-3936, :3965-3989) become these calls.  Returns 1 if accepted. */
int hip_UpdateTau_loops(const gph_tau_args *a, double lnacceptance_prior, double taufactor[2])
{
  gph_tau_result res;
  double lnacceptance = lnacceptance_prior;
  hip_push_model();                                  /* old tau, proposed band times, exactly as at :3444 */
  if (gph_engine_tau_evaluate(eng, a, &res)) exit(-1);
  lnacceptance += res.dataDelta + res.genDelta + res.ntj0 * log(taufactor[0]) + res.ntj1 * log(taufactor[1]);
  if (res.first_conflict_locus < 0 && (lnacceptance >= 0 || rndu(RAND_GENERAL_SLOT) < exp(lnacceptance))) {
    dataState.dataLogLikelihood += res.dataDelta;
    dataState.logLikelihood += (res.dataDelta + res.genDelta) / dataSetup.numLoci;
    if (gph_engine_tau_commit(eng)) exit(-1);        /* then pops[pop]->age = taunew (:3946) */
    return 1;
  }
  if (res.first_conflict_locus >= 0) misc_stats.rubberband_mig_conflicts++;
  computeMigrationBandTimes(dataSetup.popTree);
  hip_push_model();
  if (gph_engine_tau_revert(eng, res.first_conflict_locus)) exit(-1);
  return 0;
}

/* mixing (GPhoCS.c:4688): upstream keeps the parameter scaling :4709-4788 and the decision; the loops :4793, :4818,
 * :4884 become */
int hip_mixing_loops(double c, double lnc, double lnacceptance_rest)
{
  double dData;
  hip_push_model();
  if (gph_engine_mixing_evaluate(eng, c, &dData)) exit(-1);
  if (lnacceptance_rest + dData >= 0 || rndu(RAND_GENERAL_SLOT) < exp(lnacceptance_rest + dData)) {
    if (gph_engine_mixing_commit(eng, c, lnc)) exit(-1);
    dataState.dataLogLikelihood += dData;
    return 1;
  }
  return gph_engine_mixing_revert(eng);
}

/* end of an iteration (GPhoCS.c:1705-1757, 1811-1821): synchronizeEvents for every locus, checkAll every log period */
int hip_end_of_iteration(int iteration, int samplesPerLog)
{
  double oldGen, newGen, sumData, sumGen;
  int32_t ok = 1;
  hip_push_model();
  if (gph_engine_synchronize(eng, iteration == mcmcSetup.startMig, &oldGen, &newGen)) return -1;
  if (iteration == mcmcSetup.startMig) dataState.logLikelihood += (newGen - oldGen) / dataSetup.numLoci;
  if ((iteration + 1) % samplesPerLog == 0) {
    if (gph_engine_check_all(eng, &ok, &sumData, &sumGen) || !ok) return -1;
    dataState.dataLogLikelihood = sumData;
    dataState.logLikelihood = (sumGen + sumData) / dataSetup.numLoci;
  }
  return 0;
}
