/*
 * gphocs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the reference's per-locus hot path
 * (data likelihood P(X|G), genealogy prior P(G|M), the MCMC proposal sweeps).
 * Every function cites the reference file:line it follows.  Written from
 * scratch with its own index-based data layout; it shares no code with the
 * product (g-phocs_amd/csrc) and nothing in the product may include, link or
 * call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker.
 *
 * PARITY: pinned.  tests/test_oracle_vs_reference.py checks this restatement
 * against golden vectors generated here from the compiled real reference
 * (oracle/_ref/gphocs_ref, recipe oracle/Makefile, generator
 * tests/golden/make_goldens.sh): RNG streams, reflect(), per-proposal accept
 * counts + accumulators (hex-float, exact) and full per-locus state dumps.
 */
#ifndef GPHOCS_ORACLE_H
#define GPHOCS_ORACLE_H

#include <stdio.h>

#define GO_MAXK 40      /* populations (reference cap 2*NSPECIES-1 = 39, patch.h:19) */
#define GO_MAXB 100     /* migration bands (patch.h:17) */
#define GO_MAX_MIGS 10  /* migration events per genealogy (patch.h:18) */
#define GO_OLDAGE 999.0 /* patch.h:21 */

/* event types, same numbering as patch.h:159 */
enum { GO_COAL = 0, GO_IN_MIG, GO_OUT_MIG, GO_MIG_BAND_START, GO_MIG_BAND_END,
       GO_SAMPLES_START, GO_END_CHAIN, GO_DUMMY };

typedef struct {
  int n;                      /* haploid leaves per locus */
  int Kc, K, B, rootPop;
  int samplesPerPop[GO_MAXK];
  char popName[GO_MAXK][64];
  int popFather[GO_MAXK], popSon0[GO_MAXK], popSon1[GO_MAXK];
  double popAge[GO_MAXK], sampleAge[GO_MAXK], theta[GO_MAXK];
  int updateSampleAge[GO_MAXK];
  double thetaAlpha[GO_MAXK], thetaBeta[GO_MAXK], thetaStart[GO_MAXK];
  double ageAlpha[GO_MAXK], ageBeta[GO_MAXK], ageStart[GO_MAXK];
  unsigned char isAnc[GO_MAXK][GO_MAXK]; /* isAnc[a][d]: a is ancestral to (or equal) d */
  int bandSrc[GO_MAXB], bandTgt[GO_MAXB];
  double migRate[GO_MAXB], bandStart[GO_MAXB], bandEnd[GO_MAXB];
  double mrAlpha[GO_MAXB], mrBeta[GO_MAXB];
  /* mcmc settings */
  int seed, burnin, numSamples, sampleSkip, startMig, doMixing, samplesPerLog, mutRateMode;
  double ftCoalTime, ftMigTime, ftTheta, ftMigRate, ftMixing, ftTaus[GO_MAXK];
  double varRatesAlpha, ftLocusRate;   /* locus-mut-rate VAR <alpha>, finetune-locus-rate (mutRateMode 1) */
  int numParameters;
  double printFactors[3 * GO_MAXK + GO_MAXB];
} go_model;

typedef struct {
  int original_event, updated_event, num_lin_delta;
  int num_changed_events; int *changed_events;
  int num_pops_changed; int pops_changed[GO_MAXK];
  int num_bands_changed; int bands_changed[GO_MAXB];
  double coal_delta[GO_MAXK], mig_delta[GO_MAXB];
} go_delta;

typedef struct {
  int branch, band, target_pop, source_pop, target_event, source_event;
  double age;
} go_mignode;

typedef struct {
  /* sequence data */
  int P;                        /* phased patterns */
  unsigned char *leafcode;      /* [P][n]: 0..3 = T,C,A,G one-hot; 4 = N */
  int *numPhases, *count;       /* [P] */
  double mutRate;
  /* genealogy (current values) */
  int root;
  int *father, *left, *right;   /* [N] */
  double *age;                  /* [N] */
  int *nodePop, *nodeEvent;     /* [N] */
  /* conditionals: two buffers per node, condbit[i] selects the current one */
  double *cond[2];              /* each [N][P][4]; leaves are expanded from leafcode */
  unsigned char *condbit;       /* [N] */
  double dataLnL;
  /* saved version (value semantics of LocusDataLikelihood.c:64-75) */
  double sv_dataLnL; int sv_root, copyAll;
  unsigned char *dirty;         /* recalcConditionals [N] */
  int numChangedNodes, *changedNodes;
  int *sv_father, *sv_left, *sv_right; double *sv_age;
  int numChangedCond, *changedCond;
  /* event chain (patch.h:151-172) */
  int E;
  int *ev_type, *ev_node, *ev_next, *ev_prev, *ev_nlin; double *ev_time;
  int first_event[GO_MAXK], free_events;
  /* migration nodes (patch.h:138-148) */
  int num_migs, living[GO_MAX_MIGS];
  go_mignode mig[GO_MAX_MIGS];
  /* sufficient statistics (patch.h:48-51) */
  double coal_stats[GO_MAXK], mig_stats[GO_MAXB];
  int num_coals[GO_MAXK], num_migs_band[GO_MAXB];
  double chk_coal_stats[GO_MAXK], chk_mig_stats[GO_MAXB];
  int chk_num_coals[GO_MAXK], chk_num_migs[GO_MAXB];
  double genLnL, genDelta;
  /* pending-proposal storage (patch.h:60-117) */
  go_delta delta[2];
  int spr_father_event_old, spr_father_event_new, spr_father_pop_new, spr_target;
  int spr_num_old_migs, spr_num_new_migs;
  int spr_old_migs[GO_MAX_MIGS], spr_new_in[GO_MAX_MIGS], spr_new_out[GO_MAX_MIGS],
      spr_new_bands[GO_MAX_MIGS];
  double spr_new_ages[GO_MAX_MIGS], spr_delta_lnLd[2];
  int rb_num_moved, rb_orig[GO_MAX_MIGS + GO_MAXB], rb_new[GO_MAX_MIGS + GO_MAXB],
      rb_pops[GO_MAX_MIGS + GO_MAXB];
  double rb_new_ages[GO_MAX_MIGS + GO_MAXB];
  int mig_conflict_log;
  /* RNG slot (utils.c:401) */
  unsigned int rx, ry, rz;
} go_locus;

typedef struct {
  go_model m;
  int L;
  go_locus *loc;
  /* general RNG slot */
  unsigned int gx, gy, gz;
  /* accumulators (GPhoCS.h:35-50, patch.h:186) */
  double logLikelihood, dataLogLikelihood;
  double rateVar;              /* dataState.rateVar, GPhoCS.h:45 */
  double tot_coal_stats[GO_MAXK], tot_mig_stats[GO_MAXB];
  int tot_num_coals[GO_MAXK], tot_num_migs[GO_MAXB];
  int rubberband_mig_conflicts, not_enough_migs;
  long evals;                 /* count of useOld=1 likelihood evaluations */
  long evalNodes;             /* recomputed internal nodes (R) summed */
  long evalBytes;             /* algorithmic bytes, SURVEY.md section 8(d) */
  double *paramVals;
  int error;
} go_state;

/* ---- construction / io (gphocs_oracle_io.c) ---- */
go_state *go_load_pack(const char *path);
void go_free(go_state *s);
void go_dump_state(go_state *s, FILE *f, int withCond);

/* ---- RNG + helpers (gphocs_oracle_core.c) ---- */
void go_seed(go_state *s, unsigned int seed);
double go_rndu(unsigned int *x, unsigned int *y, unsigned int *z);
double go_rndnormal(unsigned int *x, unsigned int *y, unsigned int *z);
double go_rnd2normal8(unsigned int *x, unsigned int *y, unsigned int *z);
double go_reflect(double x, double a, double b);

/* ---- per-locus engines (gphocs_oracle_core.c) ---- */
double go_lik_compute(go_state *s, go_locus *q, int useOld);
void go_lik_reset_saved(go_state *s, go_locus *q);
void go_lik_revert(go_state *s, go_locus *q);
void go_compute_band_times(go_model *m);
int go_update_band_times(go_model *m, int b);
double go_gtree_lnl(go_state *s, go_locus *q);

/* ---- MCMC (gphocs_oracle_mcmc.c): same contract as GPhoCS.h:84-100 ---- */
int go_initialize_mcmc(go_state *s);
int go_update_internal_nodes(go_state *s, double finetune);
int go_update_migration_nodes(go_state *s, double finetune);
int go_update_mig_spr(go_state *s);
int go_update_theta(go_state *s, double finetune);
int go_update_mig_rates(go_state *s, double finetune);
void go_update_tau(go_state *s, const double *finetunes, int *accepted);
void go_update_sample_age(go_state *s, const double *finetunes, int *accepted);
int go_update_locus_rate(go_state *s, double finetune);
int go_mixing(go_state *s, double finetune);
int go_synchronize_events(go_state *s, go_locus *q);
int go_check_all(go_state *s);
/* one full iteration in performMCMC's order; tf may be NULL */
int go_iteration(go_state *s, int iteration, FILE *tf);
void go_trace_line(go_state *s, int iteration, FILE *tf);

#endif
