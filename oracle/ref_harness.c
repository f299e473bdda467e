/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (not part of the product path).
 *
 * Drives the *unmodified* reference implementation (compiled from its sources
 * where they lie under /root/reference/src by oracle/Makefile, output only
 * into oracle/_ref/) so that
 *   (1) golden vectors can be generated from the real reference
 *       (tests/golden/, generator: tests/golden/make_goldens.sh), and
 *   (2) the reference's own per-locus CPU path can be timed as the
 *       cpu_baseline of bench.py ("kind": "reference").
 *
 * No reference source text is copied here: this file only #includes the
 * reference headers / one .c at build time and calls their public functions
 * in the order the reference's own main() and performMCMC() call them
 * (GPhoCS.c:147-235 start-up, GPhoCS.c:1476-1821 one MCMC iteration).
 *
 * Sub-commands (argv[1]):
 *   pack  <ctl> <out.gpk>            dump model + processed loci ("pack")
 *   run   <ctl> <iters> <out.trace> [statefile] [state_iter]
 *                                    replay iterations, one record per proposal
 *   time  <ctl> <iters>              timed iterations (cpu_baseline)
 *   timesweep <ctl> <iters> <warm> <t1,t2,...>
 *                                    ONE start-up, then `iters` timed iterations per OpenMP thread count of the
 *                                    list (omp_set_num_threads, as GPhoCS.c:145 does for `-n`): the cpu_baseline
 *                                    at the benchmark's own data-set size without paying the start-up per count
 *   rng   <seed> <count>             RNG golden stream
 *   reflect                          reflect() golden table
 *   main  <args...>                  the reference's own main()
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#include "PopulationTree.h"
#include "GenericTree.h"
#include "MCMCcontrol.h"
#include "AlignmentProcessor.h"
#include "utils.h"
/* Including the .c (instead of linking its object) gives this harness access
 * to the file-local struct LOCUS_LIKELIHOOD so that conditional-likelihood
 * arrays can be dumped.  The reference object for this file is NOT linked. */
#include "LocusDataLikelihood.c"
#include "patch.h"
#include "GPhoCS.h"

extern RandGeneratorContext RndCtx;
extern int gphocs_main(int argc, char *argv[]);
extern int initializeMCMC();
extern int sampleMigRates(PopulationTree *popTree);
extern int freeAlignmentData();
extern int readSeqFile(const char *seqFileName, int numSamples,
                       char **sampleNames, int numLociToRead);

/* ------------------------------------------------------------------ */
/* start-up: same call order as the reference's main(), GPhoCS.c:147-235 */
static void startup(char *ctl)
{
  int res;
  debug = 0;
  initGeneralInfo();
  res = readControlFile(ctl);
  if (res != 0) { fprintf(stderr, "harness: readControlFile failed\n"); exit(2); }
  /* the optional secondary control file of main() (GPhoCS.c:154-164), named by the environment for pack / run */
  if (getenv("GPH_REF_CTL2") && getenv("GPH_REF_CTL2")[0]) {
    res = readSecondaryControlFile(getenv("GPH_REF_CTL2"));
    if (res != 0) { fprintf(stderr, "harness: readSecondaryControlFile failed\n"); exit(2); }
  }
  res = checkSettings();
  finalizeNumParameters();
  if (res > 0) { fprintf(stderr, "harness: %d control errors\n", res); exit(2); }
  if (mcmcSetup.randomSeed < 0) mcmcSetup.randomSeed = 12345;
  res = processAlignments();
  if (res < 0) { fprintf(stderr, "harness: processAlignments failed\n"); exit(2); }
  allocateAllMemory();
  initRandomGenerator(dataSetup.numLoci, mcmcSetup.randomSeed);
}

/* ------------------------------------------------------------------ */
static void write_model(FILE *f)
{
  PopulationTree *pt = dataSetup.popTree;
  int pop, b;
  fprintf(f, "GPHOCS-PACK 1\n");
  fprintf(f, "numLoci %d\nnumSamples %d\nnumCurPops %d\nnumPops %d\nnumMigBands %d\nrootPop %d\n",
          dataSetup.numLoci, dataSetup.numSamples, pt->numCurPops, pt->numPops,
          pt->numMigBands, pt->rootPop);
  fprintf(f, "samplesPerPop");
  for (pop = 0; pop < pt->numCurPops; pop++) fprintf(f, " %d", dataSetup.numSamplesPerPop[pop]);
  fprintf(f, "\n");
  for (pop = 0; pop < pt->numPops; pop++) {
    Population *p = pt->pops[pop];
    fprintf(f, "pop %d %s %d %d %d %a %d %a %a %a %a %a %a\n", pop, p->name,
            p->father ? p->father->id : -1,
            p->sons[0] ? p->sons[0]->id : -1, p->sons[1] ? p->sons[1]->id : -1,
            p->sampleAge, (int)(p->updateSampleAge ? 1 : 0),
            p->thetaPrior.alpha, p->thetaPrior.beta, p->thetaPrior.sampleStart,
            p->agePrior.alpha, p->agePrior.beta, p->agePrior.sampleStart);
  }
  for (b = 0; b < pt->numMigBands; b++) {
    fprintf(f, "band %d %d %d %a %a\n", b, pt->migBands[b].sourcePop,
            pt->migBands[b].targetPop, pt->migBands[b].migRatePrior.alpha,
            pt->migBands[b].migRatePrior.beta);
  }
  fprintf(f, "mcmc %d %d %d %d %d %d %d %d\n", mcmcSetup.randomSeed, mcmcSetup.burnin,
          mcmcSetup.numSamples, mcmcSetup.sampleSkip, mcmcSetup.startMig,
          (int)mcmcSetup.doMixing, ioSetup.samplesPerLog, (int)mcmcSetup.mutRateMode);
  fprintf(f, "finetunes %a %a %a %a %a", mcmcSetup.finetunes.coalTime,
          mcmcSetup.finetunes.migTime, mcmcSetup.finetunes.theta,
          mcmcSetup.finetunes.migRate, mcmcSetup.finetunes.mixing);
  for (pop = 0; pop < pt->numPops; pop++) fprintf(f, " %a", mcmcSetup.finetunes.taus[pop]);
  fprintf(f, "\n");
  if (mcmcSetup.mutRateMode == 1)
    fprintf(f, "locusrate %a %a\n", mcmcSetup.varRatesAlpha, mcmcSetup.finetunes.locusRate);
  fprintf(f, "printFactors %d", mcmcSetup.numParameters);
  for (pop = 0; pop < mcmcSetup.numParameters; pop++) fprintf(f, " %a", mcmcSetup.printFactors[pop]);
  fprintf(f, "\n");
}

/* pack: per locus, the phased pattern table exactly as initializeLocusData
 * stored it (LocusDataLikelihood.c:239-305): one row per phased pattern with
 * the leaf characters, numPhases (non-zero on the first phase only) and count */
static int cmd_pack(char *ctl, char *out)
{
  int g, p, leaf, c;
  FILE *f;
  startup(ctl);
  /* locus-mut-rate FIXED: the rates reach the loci in initializeMCMC (GPhoCS.c:1147-1155); the pack carries them as
   * readRateFile leaves them (normalised to mean 1) */
  if (mcmcSetup.mutRateMode == 2 && 0 != readRateFile(ioSetup.rateFileName)) { fprintf(stderr, "harness: readRateFile failed\n"); return 2; }
  f = fopen(out, "w");
  if (!f) { perror(out); return 2; }
  write_model(f);
  for (g = 0; g < dataSetup.numLoci; g++) {
    LocusData *ld = dataState.lociData[g];
    int P = ld->seqData.numPatterns;
    fprintf(f, "locus %d %d %a\n", g, P, ld->mutationRate);
    for (p = 0; p < P; p++) {
      for (leaf = 0; leaf < ld->numLeaves; leaf++) {
        double *cp = ld->nodeArray[leaf]->conditionalProbs + 4 * p;
        double s = cp[0] + cp[1] + cp[2] + cp[3];
        if (s >= 4) c = 'N';
        else if (cp[0] == 1.0) c = 'T';
        else if (cp[1] == 1.0) c = 'C';
        else if (cp[2] == 1.0) c = 'A';
        else c = 'G';
        fputc(c, f);
      }
      fprintf(f, " %d %d\n", ld->seqData.numPhases[p], ld->seqData.patternCount[p]);
    }
  }
  fprintf(f, "end\n");
  fclose(f);
  return 0;
}

/* ------------------------------------------------------------------ */
/* canonical per-locus state dump (shared format with the oracle restatement
 * and the HIP engine's download; compared field by field in tests) */
static void dump_state(FILE *f, int withCond)
{
  PopulationTree *pt = dataSetup.popTree;
  int g, i, pop, b, ev, N = 2 * dataSetup.numSamples - 1;
  fprintf(f, "STATE %d\n", dataSetup.numLoci);
  fprintf(f, "MODEL");
  for (pop = 0; pop < pt->numPops; pop++)
    fprintf(f, " %a %a %a", pt->pops[pop]->theta, pt->pops[pop]->age, pt->pops[pop]->sampleAge);
  for (b = 0; b < pt->numMigBands; b++)
    fprintf(f, " %a %a %a", pt->migBands[b].migRate, pt->migBands[b].startTime, pt->migBands[b].endTime);
  fprintf(f, "\n");
  fprintf(f, "GLOBAL %a %a %u %u %u\n", dataState.logLikelihood, dataState.dataLogLikelihood,
          RndCtx.rndu_x[RndCtx.nOfSlots - 1], RndCtx.rndu_y[RndCtx.nOfSlots - 1],
          RndCtx.rndu_z[RndCtx.nOfSlots - 1]);
  if (mcmcSetup.mutRateMode == 1) fprintf(f, "RATEVAR %a\n", dataState.rateVar);
  fprintf(f, "TOTALS");
  for (pop = 0; pop < pt->numPops; pop++)
    fprintf(f, " %a %d", genetree_stats_total.coal_stats[pop], genetree_stats_total.num_coals[pop]);
  for (b = 0; b < pt->numMigBands; b++)
    fprintf(f, " %a %d", genetree_stats_total.mig_stats[b], genetree_stats_total.num_migs[b]);
  fprintf(f, "\n");
  for (g = 0; g < dataSetup.numLoci; g++) {
    LocusData *ld = dataState.lociData[g];
    fprintf(f, "LOCUS %d root %d dataLnL %a genLnL %a rng %u %u %u\n", g, ld->root,
            ld->dataLogLikelihood, locus_data[g].genLogLikelihood,
            RndCtx.rndu_x[g], RndCtx.rndu_y[g], RndCtx.rndu_z[g]);
    if (mcmcSetup.mutRateMode == 1) fprintf(f, "R %a\n", ld->mutationRate);
    for (i = 0; i < N; i++) {
      fprintf(f, "N %d %d %d %d %a %d %d\n", i, ld->nodeArray[i]->father,
              ld->nodeArray[i]->leftSon, ld->nodeArray[i]->rightSon,
              ld->nodeArray[i]->age, nodePops[g][i], i < dataSetup.numSamples ? -1 : nodeEvents[g][i]);
    }
    for (pop = 0; pop < pt->numPops; pop++) {
      fprintf(f, "C %d", pop);
      for (ev = event_chains[g].first_event[pop]; ev >= 0; ev = event_chains[g].events[ev].next) {
        Event *e = &event_chains[g].events[ev];
        fprintf(f, " %d:%d:%d:%d:%a", ev, (int)e->type, e->node_id, e->num_lineages, e->elapsed_time);
      }
      fprintf(f, "\n");
    }
    fprintf(f, "S");
    for (pop = 0; pop < pt->numPops; pop++)
      fprintf(f, " %a %d", genetree_stats[g].coal_stats[pop], genetree_stats[g].num_coals[pop]);
    for (b = 0; b < pt->numMigBands; b++)
      fprintf(f, " %a %d", genetree_stats[g].mig_stats[b], genetree_stats[g].num_migs[b]);
    fprintf(f, "\n");
    fprintf(f, "M %d", genetree_migs[g].num_migs);
    for (i = 0; i < genetree_migs[g].num_migs; i++) {
      int m = genetree_migs[g].living_mignodes[i];
      struct MIGNODE *mn = &genetree_migs[g].mignodes[m];
      fprintf(f, " %d:%d:%d:%d:%d:%d:%d:%a", m, mn->gtree_branch, mn->migration_band,
              mn->source_pop, mn->target_pop, mn->source_event, mn->target_event, mn->age);
    }
    fprintf(f, "\n");
    if (withCond) {
      int P = ld->seqData.numPatterns, p, a;
      for (i = dataSetup.numSamples; i < N; i++) {
        fprintf(f, "K %d", i);
        for (p = 0; p < P; p++)
          for (a = 0; a < 4; a++)
            fprintf(f, " %a", ld->nodeArray[i]->conditionalProbs[4 * p + a]);
        fprintf(f, "\n");
      }
    }
  }
  fprintf(f, "ENDSTATE\n");
}

/* one line per proposal call: what the reference's return value and the
 * dataState accumulators were right after it (GPhoCS.h:84-100 contract) */
static void rec(FILE *f, int it, const char *what, int acc)
{
  fprintf(f, "IT %d %s %d %a %a\n", it, what, acc, dataState.dataLogLikelihood,
          dataState.logLikelihood);
}

static double *g_paramVals = NULL;

/* the per-iteration call sequence of performMCMC, GPhoCS.c:1476-1821,
 * (genetreeSamples == 1; no find-finetunes; no admixture; CONST/FIXED rates) */
static int one_iteration(FILE *tf, int iteration, int *acceptCountArray, int verboseTrace)
{
  PopulationTree *pt = dataSetup.popTree;
  int pop, gen, acc;
  acc = UpdateGB_InternalNode(mcmcSetup.finetunes.coalTime);
  if (verboseTrace) rec(tf, iteration, "INT", acc);
  acc = UpdateGB_MigrationNode(mcmcSetup.finetunes.migTime);
  if (verboseTrace) rec(tf, iteration, "MIGN", acc);
  acc = UpdateGB_MigSPR();
  if (verboseTrace) rec(tf, iteration, "SPR", acc);
  if (mcmcSetup.mutRateMode == 1) {
    acc = UpdateLocusRate(mcmcSetup.finetunes.locusRate);
    if (verboseTrace) rec(tf, iteration, "LRATE", acc);
  }
  acc = UpdateTheta(mcmcSetup.finetunes.theta);
  if (verboseTrace) rec(tf, iteration, "THETA", acc);
  if (iteration > mcmcSetup.startMig) {
    acc = UpdateMigRates(mcmcSetup.finetunes.migRate);
    if (verboseTrace) rec(tf, iteration, "MIGR", acc);
  }
  UpdateTau(mcmcSetup.finetunes.taus, acceptCountArray);
  if (verboseTrace) {
    for (pop = pt->numCurPops; pop < pt->numPops; pop++) {
      char nm[32];
      snprintf(nm, sizeof nm, "TAU%d", pop);
      rec(tf, iteration, nm, acceptCountArray[pop]);
    }
    fprintf(tf, "CONFLICTS %d\n", misc_stats.rubberband_mig_conflicts);
  }
  UpdateSampleAge(mcmcSetup.finetunes.taus, acceptCountArray);
  if (verboseTrace) {
    for (pop = 0; pop < pt->numCurPops; pop++) {
      char nm[32];
      if (!pt->pops[pop]->updateSampleAge) continue;
      snprintf(nm, sizeof nm, "SAGE%d", pop);
      rec(tf, iteration, nm, acceptCountArray[pop]);
      fprintf(tf, "CONFLICTS %d\n", misc_stats.rubberband_mig_conflicts);
    }
  }
  if (mcmcSetup.doMixing) {
    acc = mixing(mcmcSetup.finetunes.mixing);
    if (verboseTrace) rec(tf, iteration, "MIX", acc);
  }
  for (gen = 0; gen < dataSetup.numLoci; gen++) {
    if (!synchronizeEvents(gen)) { fprintf(stderr, "harness: synchronizeEvents failed gen %d\n", gen); exit(3); }
  }
  /* parameters are recorded BEFORE the start-mig resampling (GPhoCS.c:1730 vs 1738) */
  if (!g_paramVals) g_paramVals = (double *)malloc(sizeof(double) * (mcmcSetup.numParameters + 1));
  recordParamVals(g_paramVals);
  if (iteration == mcmcSetup.startMig) {
    sampleMigRates(pt);
    for (gen = 0; gen < dataSetup.numLoci; gen++) {
      dataState.logLikelihood -= locus_data[gen].genLogLikelihood / dataSetup.numLoci;
      locus_data[gen].genLogLikelihood = gtreeLnLikelihood(gen);
      dataState.logLikelihood += locus_data[gen].genLogLikelihood / dataSetup.numLoci;
    }
  }
  if ((iteration + 1) % ioSetup.samplesPerLog == 0) {
    if (!checkAll()) { fprintf(stderr, "harness: checkAll failed at iteration %d\n", iteration); exit(3); }
    if (verboseTrace) rec(tf, iteration, "CHECK", 1);
  }
  return 0;
}

static void trace_line(FILE *tf, int iteration)
{
  fprintf(tf, "TRACE %d\t", iteration);
  printParamVals(g_paramVals, 0, mcmcSetup.numParameters, tf);
  fprintf(tf, "\t%.6f\t%.6f\n", dataState.logLikelihood, dataState.dataLogLikelihood);
}

static int cmd_run(int argc, char **argv)
{
  char *ctl = argv[2];
  int iters = atoi(argv[3]);
  FILE *tf = fopen(argv[4], "w");
  char *statefile = argc > 5 ? argv[5] : NULL;
  int stateIter = argc > 6 ? atoi(argv[6]) : iters - 1; /* dump after this iteration; -1 = after init */
  int withCond = argc > 7 ? atoi(argv[7]) : 0;
  int it, totalCoals, *acceptCountArray;
  if (!tf) { perror(argv[4]); return 2; }
  startup(ctl);
  acceptCountArray = (int *)calloc(dataSetup.popTree->numPops, sizeof(int));
  misc_stats.rubberband_mig_conflicts = 0;
  misc_stats.not_enough_migs = 0;
  totalCoals = initializeMCMC();
  rec(tf, -1, "INIT", totalCoals);
  if (statefile && stateIter < 0) { FILE *sf = fopen(statefile, "w"); dump_state(sf, withCond); fclose(sf); }
  for (it = 0; it < iters; it++) {
    one_iteration(tf, it, acceptCountArray, 1);
    trace_line(tf, it);
    if (statefile && stateIter == it) { FILE *sf = fopen(statefile, "w"); dump_state(sf, withCond); fclose(sf); }
  }
  fclose(tf);
  return 0;
}

/* ------------------------------------------------------------------ */
/* kernel-level fixtures: after `iters` iterations, SINGLE calls of the reference's
 * per-locus functions with deterministic arguments, every output as a hex float; each call is undone
 * (rejectEventChainChanges / revertToSaved), so the calls are independent.  The HIP engine replays the same calls
 * (gph_engine_unit) on the same chain state; a parity break is then located by one diff instead of a bisection.
 *   A g inode tnew lnLd dprior   adjustGenNodeAge + computeLocusDataLikelihood(useOld=1) + considerEventMove
 *                                (the body of UpdateGB_InternalNode, GPhoCS.c:2316-2381, with tnew = a fixed point
 *                                of the window instead of a random draw)
 *   B g lnl                      computeLocusDataLikelihood(useOld=0): full recompute
 *   C g ap d n0 n1 lik           rubberBand(pre) x3 of one ancestral population + computeLocusDataLikelihood(1)
 *                                (UpdateTau loop 1 without the migration ripple, GPhoCS.c:3705-3831) */
static int cmd_unit(int argc, char **argv)
{
  char *ctl = argv[2];
  int iters = atoi(argv[3]);
  FILE *of = fopen(argv[4], "w");
  int it, gen, inode, i, son, mig, pop, ap, n = 0, N, *acceptCountArray;
  PopulationTree *pt;
  if (!of) { perror(argv[4]); return 2; }
  startup(ctl);
  pt = dataSetup.popTree;
  n = dataSetup.numSamples; N = 2 * n - 1;
  acceptCountArray = (int *)calloc(pt->numPops, sizeof(int));
  misc_stats.rubberband_mig_conflicts = 0;
  misc_stats.not_enough_migs = 0;
  initializeMCMC();
  { FILE *nul = fopen("/dev/null", "w"); for (it = 0; it < iters; it++) one_iteration(nul, it, acceptCountArray, 1); fclose(nul); }
  for (gen = 0; gen < dataSetup.numLoci; gen++) {
    LocusData *ld = dataState.lociData[gen];
    for (inode = n; inode < N; inode++) {
      double t = getNodeAge(ld, inode), tb[2], hi, tnew, lnLd, dprior;
      pop = nodePops[gen][inode];
      tb[0] = pt->pops[pop]->age;
      tb[1] = pop != pt->rootPop ? pt->pops[pop]->father->age : OLDAGE;
      mig = findFirstMig(gen, inode, -1);
      if (mig >= 0) tb[1] = min2(tb[1], genetree_migs[gen].mignodes[mig].age);
      else if (inode != getLocusRoot(ld)) tb[1] = min2(tb[1], getNodeAge(ld, getNodeFather(ld, inode)));
      for (i = 0; i < 2; i++) {
        son = getNodeSon(ld, inode, i);
        mig = findLastMig(gen, son, -1);
        if (mig >= 0) tb[0] = max2(tb[0], genetree_migs[gen].mignodes[mig].age);
        else tb[0] = max2(tb[0], getNodeAge(ld, son));
      }
      hi = min2(tb[1], t * 1.5 + 1e-7);
      tnew = tb[0] + 0.61803 * (hi - tb[0]);
      adjustGenNodeAge(ld, inode, tnew);
      lnLd = -getLocusDataLikelihood(ld);
      lnLd += computeLocusDataLikelihood(ld, 1);
      dprior = considerEventMove(gen, 0, nodeEvents[gen][inode], pop, t, pop, tnew);
      rejectEventChainChanges(gen, 0);
      revertToSaved(ld);
      fprintf(of, "A %d %d %a %a %a\n", gen, inode, tnew, lnLd, dprior);
    }
  }
  for (ap = pt->numCurPops; ap < pt->numPops; ap++) {
    int isRoot = ap == pt->rootPop, s0 = pt->pops[ap]->sons[0]->id, s1 = pt->pops[ap]->sons[1]->id;
    double tauold = pt->pops[ap]->age, taub[2], taunew, tf[2];
    taub[0] = max2(pt->pops[s0]->age, pt->pops[s1]->age);
    taub[0] = max2(taub[0], pt->pops[s0]->sampleAge);
    taub[0] = max2(taub[0], pt->pops[s1]->sampleAge);
    taub[1] = isRoot ? OLDAGE : pt->pops[ap]->father->age;
    taunew = taub[0] + 0.55 * (min2(taub[1], tauold * 1.4) - taub[0]);
    tf[0] = (taunew - taub[0]) / (tauold - taub[0]);
    tf[1] = isRoot ? tf[0] : (taunew - taub[1]) / (tauold - taub[1]);
    for (gen = 0; gen < dataSetup.numLoci; gen++) {
      LocusData *ld = dataState.lociData[gen];
      int n0 = 0, n1 = 0;
      double d, lik = 0.0;
      if (isRoot) d = rubberBand(gen, ap, taub[0], tauold, tf[1], 0, &n1);
      else d = rubberBand(gen, ap, taub[1], tauold, tf[1], 0, &n1);
      d += rubberBand(gen, s0, taub[0], tauold, tf[0], 0, &n0);
      d += rubberBand(gen, s1, taub[0], tauold, tf[0], 0, &n0);
      if (n0 + n1) { lik = -getLocusDataLikelihood(ld); lik += computeLocusDataLikelihood(ld, 1); }
      revertToSaved(ld);
      fprintf(of, "C %d %d %a %d %d %a\n", gen, ap, d, n0, n1, lik);
    }
  }
  for (gen = 0; gen < dataSetup.numLoci; gen++) {
    LocusData *ld = dataState.lociData[gen];
    double v = computeLocusDataLikelihood(ld, 0);
    resetSaved(ld);
    fprintf(of, "B %d %a\n", gen, v);
  }
  fclose(of);
  return 0;
}

/* ------------------------------------------------------------------ */
/* kernel-level fixtures, second set (the rest of G3 / G4): after `iters` iterations, single calls of
 *   D g node target age ret lnl root     executeGenSPR(node, target, age) + computeLocusDataLikelihood(useOld=1), then
 *                                        revertToSaved: every return code (0: same root, 1: regrafted above the root,
 *                                        2: pruned from below the root), LocusDataLikelihood.c:931-1012.  For every non-root
 *                                        node the targets are: its father, its sibling, the root, and every fifth other
 *                                        node outside its subtree; the age is a fixed point of the legal window
 *   E g arg delta v2                     scaleAllNodeAges(1 + arg / 1000) (LocusDataLikelihood.c:895-917), revertToSaved,
 *                                        then a full recompute (v2 = the value before the call if the revert is complete)
 *   F g n d_do d_undo                    rubberBandRipple(do) then rubberBandRipple(undo) (patch.c:815-869) over a list of
 *                                        moved events made of every migration event's source-side event, 0.01 % older
 *   G g node res target fpop nold nnew dl0 dl1 fage lnl x y z
 *                                        traceLineage(node, 0) + traceLineage(node, 1) (patch.c:886-1331) as UpdateGB_MigSPR
 *                                        calls them (GPhoCS.c:2659-2700) + computeLocusDataLikelihood(1): the sampled
 *                                        regraft (target edge, father's new population and age), the migration events
 *                                        removed / created, both prior deltas, the locus's generator state afterwards
 *   H g n d_do d_undo                    rubberBandRipple(do / undo) over MIGRATION-BAND events (start_or_end == 1 in UpdateTau's
 *                                        list, GPhoCS.c:3708-3745): per band the MIG_BAND_START event of the target population's
 *                                        chain moved 30 % into the gap to its successor, the MIG_BAND_END event 30 % into the gap
 *                                        to its predecessor -- only models whose bands start above their target population's age
 *                                        (an ancestral end) have a START event that is not the first of its chain
 * F, G and H change the chains; they run in forked children, so that every call starts from the same state. */
#include <sys/wait.h>
#include <unistd.h>
static int u2_descends(LocusData *ld, int x, int anc)
{
  while (x >= 0) { if (x == anc) return 1; x = getNodeFather(ld, x); }
  return 0;
}
static int cmd_unit2(int argc, char **argv)
{
  char *ctl = argv[2];
  int iters = atoi(argv[3]);
  FILE *of = fopen(argv[4], "w");
  int it, gen, node, target, i, n, N, *acceptCountArray, k;
  PopulationTree *pt;
  static const int scale_args[2] = {-30, 4};
  if (!of) { perror(argv[4]); return 2; }
  startup(ctl);
  pt = dataSetup.popTree;
  n = dataSetup.numSamples; N = 2 * n - 1;
  acceptCountArray = (int *)calloc(pt->numPops, sizeof(int));
  misc_stats.rubberband_mig_conflicts = 0;
  misc_stats.not_enough_migs = 0;
  initializeMCMC();
  { FILE *nul = fopen("/dev/null", "w"); for (it = 0; it < iters; it++) one_iteration(nul, it, acceptCountArray, 1); fclose(nul); }
  /* D */
  for (gen = 0; gen < dataSetup.numLoci; gen++) {
    LocusData *ld = dataState.lociData[gen];
    int root = getLocusRoot(ld);
    for (node = 0; node < N; node++) {
      int father, sibling, grandpa;
      if (node == root) continue;
      father = getNodeFather(ld, node);
      sibling = getNodeSon(ld, father, 0) + getNodeSon(ld, father, 1) - node;
      grandpa = getNodeFather(ld, father);
      for (target = 0; target < N; target++) {
        double lo, hi, age, lnl;
        int ret, tf;
        if (target == node || u2_descends(ld, target, node)) continue;
        if (!(target == father || target == sibling || target == root || (target + node) % 5 == 0)) continue;
        if (target == father || target == sibling) {
          lo = max2(getNodeAge(ld, node), getNodeAge(ld, sibling));
          hi = grandpa >= 0 ? getNodeAge(ld, grandpa) : lo * 1.3 + 1e-6;
        } else {
          tf = getNodeFather(ld, target);
          lo = max2(getNodeAge(ld, node), getNodeAge(ld, target));
          hi = tf >= 0 ? getNodeAge(ld, tf) : lo * 1.3 + 1e-6;
        }
        if (!(hi > lo)) continue;
        age = lo + 0.37 * (hi - lo);
        ret = executeGenSPR(ld, node, target, age);
        lnl = computeLocusDataLikelihood(ld, 1);
        fprintf(of, "D %d %d %d %a %d %a %d\n", gen, node, target, age, ret, lnl, getLocusRoot(ld));
        revertToSaved(ld);
      }
    }
  }
  /* E */
  for (k = 0; k < 2; k++) {
    const double factor = 1.0 + scale_args[k] * 0.001;
    for (gen = 0; gen < dataSetup.numLoci; gen++) {
      LocusData *ld = dataState.lociData[gen];
      double d = scaleAllNodeAges(ld, factor), v2;
      revertToSaved(ld);
      v2 = computeLocusDataLikelihood(ld, 0);
      resetSaved(ld);
      fprintf(of, "E %d %d %a %a\n", gen, scale_args[k], d, v2);
    }
  }
  fflush(of);
  /* F */
  { pid_t pid = fork();
    if (pid == 0) {
      for (gen = 0; gen < dataSetup.numLoci; gen++) {
        RUBBERBAND_MIGS *rb = &locus_data[gen].rubberband_migs;
        double d1, d0;
        int nm;
        rb->num_moved_events = 0;
        for (i = 0; i < genetree_migs[gen].num_migs; i++) {
          int mig = genetree_migs[gen].living_mignodes[i], pop = genetree_migs[gen].mignodes[mig].source_pop;
          double na = genetree_migs[gen].mignodes[mig].age * 1.0001;
          double top = pop == pt->rootPop ? OLDAGE : pt->pops[pop]->father->age;
          if (!(na < top)) continue;
          rb->orig_events[rb->num_moved_events] = genetree_migs[gen].mignodes[mig].source_event;
          rb->pops[rb->num_moved_events] = pop;
          rb->new_ages[rb->num_moved_events] = na;
          rb->num_moved_events++;
        }
        nm = rb->num_moved_events;
        d1 = rubberBandRipple(gen, 1);
        d0 = rubberBandRipple(gen, 0);
        fprintf(of, "F %d %d %a %a\n", gen, nm, d1, d0);
      }
      fflush(of);
      _exit(0);
    }
    { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { fprintf(stderr, "unit2: F child failed\n"); return 3; } }
  }
  /* H */
  { pid_t pid = fork();
    if (pid == 0) {
      for (gen = 0; gen < dataSetup.numLoci; gen++) {
        RUBBERBAND_MIGS *rb = &locus_data[gen].rubberband_migs;
        double d1, d0;
        int nm, b, ev;
        rb->num_moved_events = 0;
        for (b = 0; b < pt->numMigBands; b++) {
          int tp = pt->migBands[b].targetPop;
          double age = pt->pops[tp]->age;
          for (ev = event_chains[gen].first_event[tp]; ev >= 0; ev = event_chains[gen].events[ev].next) {
            Event *e = &event_chains[gen].events[ev];
            age += e->elapsed_time;
            if (e->node_id != b) continue;
            if (e->type == MIG_BAND_START && e->next >= 0 && event_chains[gen].events[e->next].elapsed_time > 0.0) {
              rb->orig_events[rb->num_moved_events] = ev; rb->pops[rb->num_moved_events] = tp;
              rb->new_ages[rb->num_moved_events++] = age + 0.3 * event_chains[gen].events[e->next].elapsed_time;
            } else if (e->type == MIG_BAND_END && e->elapsed_time > 0.0) {
              rb->orig_events[rb->num_moved_events] = ev; rb->pops[rb->num_moved_events] = tp;
              rb->new_ages[rb->num_moved_events++] = age - 0.3 * e->elapsed_time;
            }
          }
        }
        nm = rb->num_moved_events;
        d1 = rubberBandRipple(gen, 1);
        d0 = rubberBandRipple(gen, 0);
        fprintf(of, "H %d %d %a %a\n", gen, nm, d1, d0);
      }
      fflush(of);
      _exit(0);
    }
    { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { fprintf(stderr, "unit2: H child failed\n"); return 3; } }
  }
  /* G */
  for (node = 0; node < N; node++) {
    pid_t pid = fork();
    if (pid == 0) {
      for (gen = 0; gen < dataSetup.numLoci; gen++) {
        LocusData *ld = dataState.lociData[gen];
        MIG_SPR_STATS *ms = &locus_data[gen].mig_spr_stats;
        int res, father;
        double lnl;
        if (node == getLocusRoot(ld)) continue;
        father = getNodeFather(ld, node);
        traceLineage(gen, node, 0);
        res = traceLineage(gen, node, 1);
        lnl = -getLocusDataLikelihood(ld);
        lnl += computeLocusDataLikelihood(ld, 1);
        fprintf(of, "G %d %d %d %d %d %d %d %a %a %a %a %u %u %u\n", gen, node, res, res >= 0 ? ms->target : -1,
                res >= 0 ? ms->father_pop_new : -1, ms->num_old_migs, ms->num_new_migs, ms->genetree_delta_lnLd[0],
                ms->genetree_delta_lnLd[1], getNodeAge(ld, father), lnl, RndCtx.rndu_x[gen], RndCtx.rndu_y[gen], RndCtx.rndu_z[gen]);
      }
      fflush(of);
      _exit(0);
    }
    { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { fprintf(stderr, "unit2: G child failed (node %d)\n", node); return 3; } }
  }
  fclose(of);
  return 0;
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* cpu_baseline: time `iters` full iterations after `warm` warm-up iterations */
static int cmd_time(int argc, char **argv)
{
  char *ctl = argv[2];
  int iters = atoi(argv[3]);
  int warm = argc > 4 ? atoi(argv[4]) : 2;
  int it, *acceptCountArray;
  double t0, t1;
  long evalsPerLocusIter;
  startup(ctl);
  acceptCountArray = (int *)calloc(dataSetup.popTree->numPops, sizeof(int));
  initializeMCMC();
  for (it = 0; it < warm; it++) one_iteration(NULL, it, acceptCountArray, 0);
  t0 = now_s();
  for (; it < warm + iters; it++) one_iteration(NULL, it, acceptCountArray, 0);
  t1 = now_s();
  /* nominal count of computeLocusDataLikelihood(useOld=1) calls per locus and
   * iteration: (n-1) + (2n-2) + A + mixing */
  evalsPerLocusIter = (dataSetup.numSamples - 1) + (2 * dataSetup.numSamples - 2) +
                      (dataSetup.popTree->numPops - dataSetup.popTree->numCurPops) +
                      (mcmcSetup.doMixing ? 1 : 0);
  printf("{\"loci\": %d, \"iters\": %d, \"seconds\": %.6f, \"iters_per_s\": %.6f, "
         "\"evals_per_s\": %.3f, \"nominal_evals_per_locus_iter\": %ld, \"dataLnL\": %.6f}\n",
         dataSetup.numLoci, iters, t1 - t0, iters / (t1 - t0),
         (double)evalsPerLocusIter * dataSetup.numLoci * iters / (t1 - t0), evalsPerLocusIter,
         dataState.dataLogLikelihood);
  return 0;
}

/* cpu_baseline at full size: start up once, time every thread count of the list on the running chain */
#ifdef _OPENMP
#include <omp.h>
#endif
static int cmd_timesweep(int argc, char **argv)
{
  char *ctl = argv[2], *list = argc > 5 ? argv[5] : (char *)"1", *tok;
  int iters = atoi(argv[3]), warm = atoi(argv[4]);
  int it = 0, k, *acceptCountArray;
  long evalsPerLocusIter;
  double t0, t1, ts = now_s();
  startup(ctl);
  acceptCountArray = (int *)calloc(dataSetup.popTree->numPops, sizeof(int));
  initializeMCMC();
  printf("{\"startup_seconds\": %.3f, \"loci\": %d}\n", now_s() - ts, dataSetup.numLoci);
  fflush(stdout);
  evalsPerLocusIter = (dataSetup.numSamples - 1) + (2 * dataSetup.numSamples - 2) +
                      (dataSetup.popTree->numPops - dataSetup.popTree->numCurPops) +
                      (mcmcSetup.doMixing ? 1 : 0);
  for (tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
    int threads = atoi(tok);
    if (threads < 1) continue;
#ifdef _OPENMP
    omp_set_num_threads(threads);
#else
    threads = 1;
#endif
    for (k = 0; k < warm; k++, it++) one_iteration(NULL, it, acceptCountArray, 0);
    /* every iteration timed on its own: the MEDIAN is the figure (a two-iteration mean was a single draw);
     * "threads:iters" in the list overrides the iteration count for that thread count */
    { double per[64], tmp;
      int n_it = strchr(tok, ':') ? atoi(strchr(tok, ':') + 1) : iters, a, b;
      if (n_it < 1) n_it = 1;
      if (n_it > 64) n_it = 64;
      t0 = now_s();
      for (k = 0; k < n_it; k++, it++) { double s0 = now_s(); one_iteration(NULL, it, acceptCountArray, 0); per[k] = now_s() - s0; }
      t1 = now_s();
      for (a = 0; a < n_it; a++) for (b = a + 1; b < n_it; b++) if (per[b] < per[a]) { tmp = per[a]; per[a] = per[b]; per[b] = tmp; }
      tmp = n_it % 2 ? per[n_it / 2] : 0.5 * (per[n_it / 2 - 1] + per[n_it / 2]);
      printf("{\"threads\": %d, \"loci\": %d, \"iters\": %d, \"seconds\": %.6f, \"median_iteration_seconds\": %.6f, "
             "\"min_iteration_seconds\": %.6f, \"max_iteration_seconds\": %.6f, \"iters_per_s\": %.6f, "
             "\"evals_per_s\": %.3f, \"nominal_evals_per_locus_iter\": %ld}\n",
             threads, dataSetup.numLoci, n_it, t1 - t0, tmp, per[0], per[n_it - 1], 1.0 / tmp,
             (double)evalsPerLocusIter * dataSetup.numLoci / tmp, evalsPerLocusIter);
    }
    fflush(stdout);
  }
  return 0;
}

/* ingest: wall time of the reference's own start-up path (readControlFile + readSeqFile +
 * processAlignments, GPhoCS.c:150-205): the CPU baseline of the sequence front end */
static int cmd_ingest(char *ctl)
{
  double t0 = now_s(), t1;
  long phased = 0;
  int g;
  startup(ctl);
  t1 = now_s();
  for (g = 0; g < dataSetup.numLoci; g++) phased += dataState.lociData[g]->seqData.numPatterns;
  printf("{\"loci\": %d, \"samples\": %d, \"phased_patterns\": %ld, \"seconds\": %.6f}\n", dataSetup.numLoci,
         dataSetup.numSamples, phased, t1 - t0);
  return 0;
}

static int cmd_rng(int argc, char **argv)
{
  unsigned int seed = (unsigned int)strtoul(argv[2], NULL, 10);
  int count = atoi(argv[3]), i;
  initRandomGenerator(1, seed);
  /* slot 0: rndu; slot 1 (general): interleaved rnd2normal8 / rndexp / rndnormal */
  for (i = 0; i < count; i++) printf("U %a\n", rndu(0));
  for (i = 0; i < count; i++) {
    printf("N8 %a\n", rnd2normal8(1));
    printf("E %a\n", rndexp(1, 0.37));
    printf("NN %a\n", rndnormal(1));
  }
  printf("X %u %u %u %u %u %u\n", RndCtx.rndu_x[0], RndCtx.rndu_y[0], RndCtx.rndu_z[0],
         RndCtx.rndu_x[1], RndCtx.rndu_y[1], RndCtx.rndu_z[1]);
  return 0;
}

static int cmd_reflect(void)
{
  /* deterministic table of (x,a,b) triples exercising every branch */
  static const double as[] = {0.0, 1e-5, 0.25, -3.0};
  static const double ws[] = {1e-10, 2.5e-9, 1e-6, 0.01, 1.0, 7.5};
  int ia, iw, k;
  debug = 0;
  for (ia = 0; ia < 4; ia++)
    for (iw = 0; iw < 6; iw++)
      for (k = -40; k <= 40; k++) {
        if (k == 0) continue; /* x == a with a 0.5e-9-wide window ping-pongs forever in the reference */
        double a = as[ia], b = a + ws[iw];
        double x = a + ws[iw] * (0.37 * k + 0.011 * k * k * (k % 3 - 1));
        printf("R %a %a %a %a\n", x, a, b, reflect(x, a, b));
      }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "usage: gphocs_ref pack|run|time|rng|reflect|main ...\n"); return 1; }
  if (!strcmp(argv[1], "pack") && argc >= 4) return cmd_pack(argv[2], argv[3]);
  if (!strcmp(argv[1], "run") && argc >= 5) return cmd_run(argc, argv);
  if (!strcmp(argv[1], "unit") && argc >= 5) return cmd_unit(argc, argv);
  if (!strcmp(argv[1], "unit2") && argc >= 5) return cmd_unit2(argc, argv);
  if (!strcmp(argv[1], "time") && argc >= 4) return cmd_time(argc, argv);
  if (!strcmp(argv[1], "timesweep") && argc >= 5) return cmd_timesweep(argc, argv);
  if (!strcmp(argv[1], "ingest") && argc >= 3) return cmd_ingest(argv[2]);
  if (!strcmp(argv[1], "rng") && argc >= 4) return cmd_rng(argc, argv);
  if (!strcmp(argv[1], "reflect")) return cmd_reflect();
  if (!strcmp(argv[1], "main")) return gphocs_main(argc - 1, argv + 1);
  fprintf(stderr, "bad arguments\n");
  return 1;
}
