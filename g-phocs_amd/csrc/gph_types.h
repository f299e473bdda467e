// gph_types.h -- MI355X-native per-locus likelihood engine: data layout.
//
// HBM layout (one process per GPU, loci sharded across ranks):
//   pages  : L fixed-size "locus pages" (genealogy, event chain, migration
//            nodes, sufficient statistics, RNG slot, pending-move storage),
//            16-byte aligned, contiguous => a wavefront stages its locus into
//            LDS with perfectly coalesced 16-B/lane loads.
//   shadow : a second array of L pages written by the *evaluate* kernels of the
//            global proposals (tau rubber band, mixing); commit/revert kernels
//            read it.  Loci the serial reference would never have touched after
//            a migration conflict keep their main page.
//   cond   : per locus [2][n-1][P][4] fp64 conditional likelihoods of the internal
//            nodes (double buffer, bit `node` of the page's IS_CBIT set selects the current
//            half) -- leaves are not stored as doubles: a leaf is a base code.
//   seq    : per locus leaf codes u8[P][n], phases u16[P], counts i32[P] (read-only).
//
// Replaces the reference's per-locus heap structures: struct LOCUS_LIKELIHOOD
// (LocusDataLikelihood.c:40-104), Event / EVENT_CHAIN (patch.h:159-172),
// GENETREE_MIGS (patch.h:138-148), GENETREE_STATS (patch.h:48-51),
// Locus_SuperStruct (patch.h:108-117), RndCtx slots (utils.c:401).
#pragma once
#include <stdint.h>

// Compile-time capacities of this build of the library (variants: g-phocs_amd/__init__.py).  Up to 32 leaves / 32
// populations the per-locus code keeps one genealogy node per lane and 32-bit population sets; the largest variant
// (64 leaves, the reference's own 39 populations) uses 128-bit node sets, 64-bit population sets, 16-bit event ids and
// the list-driven forms of the functions that are lane-per-node programs below that size.
#ifndef GPH_CAP_LEAVES
#define GPH_CAP_LEAVES 24
#endif
#ifndef GPH_CAP_K
#define GPH_CAP_K 16
#endif
#ifndef GPH_CAP_B
#define GPH_CAP_B 8
#endif
#if GPH_CAP_K > 32
#define GPH_MAXK 40        // populations (reference cap 2 * NSPECIES - 1 = 39, patch.h:19)
typedef uint64_t gph_popmask;
#else
#define GPH_MAXK 32
typedef uint32_t gph_popmask;
#endif
// migration bands: 16 in every build that keeps the live-band list of a chain walk as 16 nibbles of one scalar; the
// reference's own cap (MAX_MIG_BANDS 100, patch.h:17) in the build whose band capacity exceeds 16 -- there the list sits
// in LDS (GphLds::s_live), band sets are GPH_BANDW words wide, the model no longer fits the 4-KB kernel-argument segment
// (every kernel reads it from the chain state in HBM) and the reduced row has 384 columns
#if GPH_CAP_B > 16
#define GPH_MAXB 100
#define GPH_BIG_BANDS 1
#else
#define GPH_MAXB 16
#define GPH_BIG_BANDS 0
#endif
#define GPH_BANDW ((GPH_MAXB + 31) / 32)   // 32-bit words of a set of bands
#ifndef GPH_MAX_MIGS
#define GPH_MAX_MIGS 10    // migration events per genealogy (patch.h:18); smaller only in LDS-size experiments
#endif
#if GPH_CAP_LEAVES > 32
#define GPH_BIG_TREE 1     // 2n - 1 > 64 genealogy nodes: no lane-per-node programs, node sets of GPH_NSQ 64-bit words
#define GPH_NSQ ((2 * GPH_CAP_LEAVES - 1 + 63) / 64)     // 2 up to 64 leaves, 7 at the reference's NS 200 (patch.h:22)
#else
#define GPH_BIG_TREE 0
#define GPH_NSQ 1
#endif
#define GPH_NSW (2 * GPH_NSQ)   // 32-bit page words per node set
#define GPH_OLDAGE 999.0   // patch.h:21
#define GPH_WAVE 64
#define FS_COUNT_ 5
#define IS_COUNT_ (10 + 3 * GPH_NSW)

enum { GPH_COAL = 0, GPH_IN_MIG, GPH_OUT_MIG, GPH_MIG_BAND_START, GPH_MIG_BAND_END,
       GPH_SAMPLES_START, GPH_END_CHAIN, GPH_DUMMY };

// model parameters read by the kernels: passed BY VALUE as a kernel argument so
// that every access is a scalar (SGPR) load from the kernarg segment.
struct GphModel {
  double theta[GPH_MAXK], popAge[GPH_MAXK], sampleAge[GPH_MAXK];
  double thetaInv[GPH_MAXK];           // RN(1/theta), host division: see gph_div_theta()
  // log(2 / theta[pop]) and log(migRate[band]) (gph_math.h's log: bit-identical on host and device): both are added
  // once per lineage walk / migration event of the SPR (patch.c:1325, :1268) -- a table entry instead of a division
  // and a ~50-instruction wave-uniform logarithm twice per proposal.  Kept current by gg_set_theta / gg_set_mig.
  double logTwoTheta[GPH_MAXK], logMigRate[GPH_MAXB];
  double migRate[GPH_MAXB], bandStart[GPH_MAXB], bandEnd[GPH_MAXB];
  gph_popmask isAnc[GPH_MAXK];         // bit d of isAnc[a]: a is ancestral to (or is) d
  uint32_t bandsOver[GPH_MAXK][GPH_BANDW];   // bit b of bandsOver[p]: band b's target population is p or an ancestor of p (computeMigStatsDelta's filter, patch.c:1846)
  uint32_t bandsInto[GPH_MAXK][GPH_BANDW];   // bit b of bandsInto[p]: band b's target population IS p (the bands a lineage in p can leave through, live_bands_into)
  // 32-bit entries: a scalar load cannot fetch 16 bits, and a 16-bit table would be read with vector loads
  // (a VMEM round trip on the chain's critical path for a wave-uniform value)
  int32_t popFather[GPH_MAXK], popSon0[GPH_MAXK], popSon1[GPH_MAXK], samplesPerPop[GPH_MAXK];
  int32_t bandSrc[GPH_MAXB], bandTgt[GPH_MAXB];
  int32_t postOrder[GPH_MAXK];         // populationPostOrder(rootPop), patch.c:1936
  int32_t cumSamples[GPH_MAXK];
};

// dimensions + byte offsets of the page arrays inside GphLds (uniform over loci; used by the small
// per-locus-thread kernels and the host-side dump, everything else addresses GphLds members directly)
struct GphLayout {
  int32_t n, N, K, Kc, B, E, RB, rootPop;
  // f64
  int32_t o_ev;            // event records (GphEv[E])
  int32_t o_nd, o_sv;      // node records (GphNode[N]) and their saved copies
  int32_t o_mig_age, o_coal, o_migst, o_rb_age, o_fscal;
  // i16
  int32_t o_nev;
  int32_t o_first;
  int32_t o_mig_i, o_living, o_ncoal, o_nmig, o_rb_i;
  // i32
  int32_t o_iscal;
  int32_t page_bytes;      // multiple of 16: the page part of GphLds
  int32_t Pmax;            // max phased patterns of any locus on this device
  int32_t lds_bytes;       // largest dynamic-LDS allocation of a launch (sequence block [+ terms])
  int32_t lds_sum;         // 1: the root reduction hands its per-pattern terms over through dynamic LDS (ordered_sum64_lds) for the loci whose terms fit behind their block
  int32_t cnt16;           // 1: pattern counts are stored as u16 (every count of the data set is below 65536)
  int32_t dyn_bytes;       // dynamic LDS of THIS launch (set per launch group)
  int32_t huge_P;          // a locus with more phased patterns keeps its sequence block in HBM (its own launch group; INT32_MAX: none)
};
struct GphGlobal;
// model + layout tables travel BY VALUE as the first argument of every kernel: in the kernarg segment every
// access is one scalar load off the (always live) kernarg pointer -- a __constant__ symbol costs a
// pc-relative address computation (3 scalar instructions) per access and an upload per change
struct GphKargs {
#if !GPH_BIG_BANDS
  GphModel model;        // (the many-band build: 8 KB, beyond the kernel-argument segment -- read from G->model)
#endif
  GphLayout lay;
  // exp() / log() / rndu() constants (gph_math.h, gph_libm_tables.h): [0..7] exp, [8..25] log, [26..31] 1/m and m of
  // the three Wichmann-Hill streams; and the device addresses of the two 128-entry libm tables
  double mathc[32];
  const double *log_t;
  const uint64_t *exp_t;
  // device-resident chain state (GphGlobal below): the kernels of the global proposals read the model and the
  // pending proposal from HERE (they are launched before the host knows either), the genealogy sweep reads
  // `model` above (the host knows it at every iteration boundary)
  GphGlobal *G;
};

// Sequence block of one locus (HBM block format == dynamic-LDS image), sized by the locus' OWN number of
// phased patterns P: leaf codes, 4 bits each, u8[P][(n + 1) / 2] (leaf i of pattern p: nibble i & 1 of byte
// p * ((n + 1) / 2) + i / 2; codes are 0..3 = T C A G, 4 = N) | (pad to 2) phases u16[P] | (pad to 4) pattern counts
// u16[P] (c16: every count of the data set fits, GphLayout.cnt16) or i32[P] | (pad to 16); behind it f64 terms of the
// root reduction: P of them for loci with more than one pattern per lane, up to 64 for the others when the launch
// group's dynamic LDS has the room (GphLayout.dyn_bytes).  The block is what decides how many loci are resident per CU
// next to the 4.3-KB static image (LDS comes in 1280-byte granules): a 64-pattern block is 768 bytes this way, 1424 with
// a byte per code and 32-bit counts.
// Phase counts are 16-bit words (the reference keeps an int; 8 unbroken heterozygotes in one repeated alignment column
// already give 2^8 = 256 phases, AlignmentProcessor.c:998-1158).  Round 6: a count of 2^15 or more -- always a power of two
// upstream (2^hets) -- is stored as 0x8000 | exponent, so the word covers every count an int can (GPH_PHASES decodes; only
// the generic (pattern, base) paths can meet one: such a pattern alone has more rows than a wavefront has lanes).  The same
// encoding travels through the C ABI's uint16_t arrays (gph_engine_load_loci, gph_loci_get).
#define GPH_PHASES(w) ((int)(w) < 0x8000 ? (int)(w) : (1 << ((int)(w) & 31)))
#define GPH_PHASES_ENCODE_MIN 0x8000
#define GPH_Q_LEAF 0
#define GPH_Q_NH(n) (((n) + 1) >> 1)
#define GPH_Q_PHASES(P, n) (((P) * GPH_Q_NH(n) + 1) & ~1)
#define GPH_Q_COUNT(P, n) ((GPH_Q_PHASES(P, n) + 2 * (P) + 3) & ~3)
#define GPH_Q_BYTES(P, n, c16) ((GPH_Q_COUNT(P, n) + ((c16) ? 2 : 4) * (P) + 15) & ~15)
#define GPH_Q_TERMS(P, n, c16) GPH_Q_BYTES(P, n, c16)
// delta scalars (s_di[inst])
enum { DI_ORIG = 0, DI_UPD, DI_DLIN, DI_NEV, DI_NPOPS, DI_NBANDS, DI_SRCPOP, DI_TGTPOP, DI_COUNT };   // DI_SRCPOP / DI_TGTPOP: the populations of the original / the new event
// spr scalars (register lanes, GphCtx), i16 arrays (s_spri16 + 10*k), f64 (s_sprf: new_ages[0..9], dlnLd[10..11])
enum { SI_FEV_OLD = 0, SI_FEV_NEW, SI_FPOP_NEW, SI_TARGET, SI_NOLD, SI_NNEW, SI_COUNT };
enum { SA_OLD = 0, SA_NEWIN, SA_NEWOUT, SA_NEWBAND };
// counters (s_cnt i32): evals, evalNodes, error, P, U ; (s_cntf f64): evalBytes
enum { CN_EVALS = 0, CN_NODES, CN_ERROR, CN_P, CN_NOTENOUGH, CN_RX, CN_RY, CN_RZ, CN_EMPTY, CN_NODES0, CN_QPH, CN_QCNT, CN_QTERMS, CN_SUMLDS, CN_HUGE, CN_SEQLO, CN_SEQHI, CN_COUNT };   // CN_HUGE: the locus's sequence block outgrows the launch group's LDS and is read where it lies in HBM (CN_SEQLO / CN_SEQHI: its address; GphSeq, gph_rt.h);   // CN_QPH / CN_QCNT / CN_QTERMS: byte offsets of the phases, the counts and the terms inside the locus's sequence block (GPH_Q_*: functions of P alone, derived once per kernel instead of in every evaluation); CN_SUMLDS: 1 when the locus's per-pattern terms fit behind its block in this launch's dynamic LDS (ordered_sum64_lds); CN_EMPTY: useOld evaluations that found nothing to recompute; CN_NODES0: nodes recomputed by useOld = 0 evaluations (both off the hot path: out_common derives the algorithmic bytes from them)
//   // CN_RX..: the batched generator's state after its current batch

// f64 scalars in the page (index into o_fscal)
enum { FS_DATALNL = 0, FS_SV_DATALNL, FS_GENLNL, FS_GENDELTA, FS_MUTRATE, FS_COUNT };
// i32 scalars in the page (index into o_iscal)
// IS_DIRTY / IS_CBIT / IS_SAVED: three 64-bit node sets (bit = genealogy node), two words each -- the nodes whose
// conditionals were recomputed by the pending proposal (savedVersion.recalcConditionals, LocusDataLikelihood.c:75-104),
// the half of the double buffer that holds each node's CURRENT conditionals, and the nodes whose record was saved
// (savedVersion.changedNodeIds).  While a kernel works on the locus they are three scalar registers (GphCtx).
enum { IS_ROOT = 0, IS_SV_ROOT, IS_DIRTY0, IS_CBIT0 = IS_DIRTY0 + GPH_NSW, IS_SAVED0 = IS_CBIT0 + GPH_NSW,
       IS_FREE = IS_SAVED0 + GPH_NSW, IS_NUM_MIGS, IS_RB_NUM, IS_CONFLICT_LOG, IS_RX, IS_RY, IS_RZ, IS_COUNT };
static_assert(IS_COUNT <= IS_COUNT_, "page scalars");
// i16 fields per migration node (o_mig_i + 6*mig)
enum { MG_BRANCH = 0, MG_BAND, MG_SPOP, MG_TPOP, MG_SEV, MG_TEV, MG_COUNT };

// per-locus outputs of a kernel launch (reduced over loci afterwards)
#define GPH_OUT_SLOTS 16
enum { OUT_ACCEPT = 0, OUT_DDATA, OUT_DLOG, OUT_EVALS, OUT_EVALNODES, OUT_EVALBYTES,
       OUT_NTJ0, OUT_NTJ1, OUT_CONFLICT, OUT_ERROR, OUT_GENLNL, OUT_DATALNL, OUT_NMIGS };

// ---------------------------------------------------------------------------------------
// LDS image of one locus.  Statically laid out with compile-time capacities so that every
// access is `ds_read/ds_write <constant offset>(index)` and the compiler knows that different
// arrays do not alias (loads are hoisted, paired and kept in registers) -- with run-time
// offsets every access cost an extra scalar add + v_mov and serialised behind every store.
// The HBM page is the page part of this struct verbatim (one coalesced copy in, one out).
// Capacities cover every BASELINE config (config 5: 20 leaves, 13 populations, 4 bands).
#define GPH_CAP_N (2 * GPH_CAP_LEAVES - 1)
#define GPH_CAP_E (2 * GPH_CAP_LEAVES + 4 * GPH_MAX_MIGS + 3 * GPH_CAP_B + GPH_CAP_K + 10)
#define GPH_CAP_RB (GPH_MAX_MIGS + 2 * GPH_CAP_B)

// One event = one 16-byte record (Event, patch.h:151-165): a chain walk needs next / lineages / type /
// elapsed time / node of the SAME event at every step, and one ds_read_b128 fetches them all (the
// wave-uniform access overhead -- address move, wait, readfirstlane -- is paid once per event, not per field).
struct alignas(16) GphEv {
  double time;            // elapsed_time
  int16_t next, prev;     // chain links
  int16_t node;           // node_id (genealogy node, migration node or band)
  uint8_t nlin;           // num_lineages (0 .. number of leaves <= 200: never negative)
  uint8_t type;           // EventType
};

// One genealogy node = one 16-byte record (GenericBinaryTree / LikelihoodNode, LocusDataLikelihood.c:40-104);
// the saved copy (savedVersion) is the same record, so saving a node is one 16-byte LDS copy.
struct alignas(16) GphNode {
  double age;
  int16_t father, left, right;
  int16_t npop;           // nodePops[gen][node] (unused in the saved copy)
};

#if GPH_CAP_E <= 255
typedef uint8_t gph_evid;    // event ids of the pending-delta lists
#else
typedef uint16_t gph_evid;
#endif
struct alignas(16) GphLds {
  // ---- page (mirrors the HBM page arrays, GphLayout o_*)
  GphEv ev[GPH_CAP_E];
  GphNode nd[GPH_CAP_N], sv[GPH_CAP_N];
  double mig_age[GPH_MAX_MIGS];
  double coal[GPH_CAP_K], migst[GPH_CAP_B], rb_age[GPH_CAP_RB], fscal[FS_COUNT_];
  int32_t iscal[IS_COUNT_];
  int16_t nev[GPH_CAP_N];
  int16_t first[GPH_CAP_K];
  int16_t mig_i[GPH_MAX_MIGS * 6], living[GPH_MAX_MIGS], ncoal[GPH_CAP_K], nmig[GPH_CAP_B], rb_i[3 * GPH_CAP_RB];
  // ---- LDS-only scratch: pending-proposal storage of GENETREE_STATS_DELTA x2 (patch.h:60-72),
  // MIG_SPR_STATS (patch.h:97-105), genetree_stats_check (patch.h:109), pruning work lists
  double s_dcoal[2][GPH_CAP_K], s_dmig[2][GPH_CAP_B];
  double s_cntf[8];   // s_cntf: 0 algorithmic bytes, 2..6 sweep accumulators, 7 step size
  // the SPR's scalars (sweep kernel, one proposal at a time) and the statistics being re-derived (recalcStats,
  // computeGenetreeStats, checkAll: tau / sample-age / refresh / check code, never inside an SPR) share their bytes
  union {
    struct { double s_sprf[GPH_MAX_MIGS + 2]; int16_t s_spri16[4 * GPH_MAX_MIGS]; };
    struct { double s_chkcoal[GPH_CAP_K], s_chkmig[GPH_CAP_B]; int16_t s_chknc[GPH_CAP_K], s_chknm[GPH_CAP_B]; };
  };
#if defined(GPH_STAMPS) || defined(GPH_HOSTEMU)
  double s_stamp[8];          // diagnostic cycle sums (tools/stamp_breakdown.py); not in production device builds
#endif
  int32_t s_di[2][8];   /* the SPR scalars and the counters (SI_*, CN_*) live in register lanes: GphCtx, gph_locus.h */
  uint32_t s_condptr[2];
  int16_t s_dpops[2][GPH_CAP_K], s_dbands[2][GPH_CAP_B];
  int16_t s_targets[GPH_CAP_N + 1];   // candidate edges of a regraft (getEdgesForTimePop); a per-population work list of init / check code
#if GPH_BIG_TREE || defined(GPH_HOSTEMU)
  int16_t s_ord[GPH_CAP_N + 1], s_stack[GPH_CAP_N + 1];   // the list-driven pruning (the lane-per-node builds have no lists)
#endif
  gph_evid s_dev[2][GPH_CAP_E];  // event lists of the two pending deltas
#if GPH_BIG_BANDS
  uint8_t s_live[(GPH_CAP_B + 15) & ~15];   // the live-band list of the chain walk in progress (one walk at a time: LiveList, gph_locus.h)
#endif
#if GPH_BIG_TREE
  double s_pe[GPH_CAP_N];        // edge transition probabilities of an evaluation, by child node (the smaller builds keep them in the lane of the node)
#endif
#ifdef GPH_PAD
  char s_pad[GPH_PAD];           // LDS-size sensitivity experiments only
#endif
};

static_assert(GPH_CAP_LEAVES <= 200 && GPH_CAP_K <= GPH_MAXK && GPH_CAP_B <= GPH_MAXB, "capacities beyond the reference's own caps (200 leaves, 39 populations, 100 bands: patch.h:17-22)");

// arguments of the tau-evaluate kernel (host part of UpdateTau, GPhoCS.c:3224-3461)
struct GphTauArgs {
  int32_t ap, son0, son1, isRoot, num_aff, mode;   // mode 1 = UpdateSampleAge (GPhoCS.c:4006)
  double tauold, taunew, taub0, taub1, taufactor0, taufactor1;
  int32_t aff_bands[GPH_MAXB * 2];     // 32-bit: scalar loads (see GphModel)
  int32_t start_or_end[GPH_MAXB * 2];
  double new_band_ages[GPH_MAXB * 2];
};

// what the finish (commit / revert) of a decided UpdateTau / UpdateSampleAge proposal needs, frozen by the decision stage
// (gg_tau_decide): the finish normally rides at the HEAD of the next evaluate kernel (kb_tau_eval of the next
// population, kb_mix_eval), i.e. after the stage that proposes the next move has already applied the accepted age and
// the next proposal's band times to the model -- so it reads the proposal and the chain start ages it walks from HERE
struct GphTauFin {
  int32_t ap, son0, son1, isRoot, mode, flag;   // flag 1 = accepted: commit, 0 = revert
  long long limit;                              // first conflicting locus (global index) or 1 << 62
  double tauold, taunew, taub0, taub1, taufactor0, taufactor1;
  double age_ap, age_s0, age_s1;                // popAge of the three populations as the evaluate kernel saw them
};

// ---------------------------------------------------------------------------------------
// Chain state above the loci: what the reference keeps in process-wide globals on its main thread
// (dataState GPhoCS.h:35-50, mcmcSetup finetunes MCMCcontrol.h:80-99, the population tree's parameters and
// priors PopulationTree.h:60-101, the general RNG slot utils.h:34, genetree_stats_total patch.h:121) plus
// the pending global proposal.  ONE copy lives in HBM: the decisions of the global proposals (UpdateTheta,
// UpdateMigRates, UpdateTau, UpdateSampleAge, mixing) are taken by a one-wavefront kernel (gph_global.h,
// k_global) straight from the reduced vectors, and the commit / revert kernels are predicated on the flag it
// leaves here -- an iteration is ONE stream of launches and one host synchronisation at its end.  The host
// keeps a mirror (read back at the end of every iteration) for the trace writer and the next sweep's kernarg.
// reduced vectors of one launch: section 0 = per-locus outputs (GPH_OUT_SLOTS columns), section 1 = the compact
// statistics (2K+2B columns); per section sum / min / max per column + the sticky error word.  With several
// ranks every rank's row is all-gathered (RCCL, on the engine's stream) and combined in rank order.
#if GPH_BIG_BANDS
#define GPH_RED_COLS 384      /* 2 * 39 populations + 2 * 100 bands = 278 statistics columns */
#else
#define GPH_RED_COLS 128
#endif
#define GPH_RED_STRIDE (3 * GPH_RED_COLS + 8)
#define GPH_RED_ROW (2 * GPH_RED_STRIDE)
// stages of an iteration that run above the loci (gph_global.h: gg_stage)
enum { GS_INIT_DONE = 0, GS_SWEEP_DONE, GS_TOTALS, GS_THETA, GS_TAU_PROPOSE, GS_TAU_DECIDE, GS_TAU_END, GS_SAGE_PROPOSE,
       GS_SAGE_DECIDE, GS_SAGE_END, GS_MIX_PROPOSE, GS_MIX_DECIDE, GS_STARTMIG, GS_REFRESH_DONE, GS_CHECK_DONE,
       GS_COUNT_ONLY,
       GS_THETA_ONLY, GS_MIGR_ONLY };   /* UpdateTheta / UpdateMigRates as calls of their own (gph_mcmc_update_theta / _mig_rates) */
// the parts of an iteration, one per function performMCMC calls (GPhoCS.h:84-100; gph_engine_part_)
enum { GPH_PART_SWEEP = 0, GPH_PART_LRATE, GPH_PART_THETA, GPH_PART_MIGR, GPH_PART_TAU, GPH_PART_SAGE, GPH_PART_MIX,
       GPH_PART_SYNC, GPH_PART_REFRESH, GPH_PART_CHECK };
#define GPH_REC_MAX (16 + 3 * GPH_MAXK)
struct GphRec { int32_t code, idx; int64_t acc; double dataLnL, logL; };
enum { REC_INIT = 0, REC_INT, REC_MIGN, REC_SPR, REC_LRATE, REC_THETA, REC_MIGR, REC_TAU, REC_CONFLICTS, REC_SAGE,
       REC_MIX, REC_CHECK };
struct GphApply { int32_t kind, idx; double lnc, diff; };   // kind 0: population idx (diff = 1/new - 1/old), 1: band idx (diff = new - old rate)
struct alignas(16) GphGlobal {
  GphModel model;
  GphTauArgs tau;                    // pending UpdateTau / UpdateSampleAge proposal
  long long tau_limit;               // first conflicting locus (global index) or 1 << 62
  int32_t tau_flag, mix_flag;        // 1 = accepted: the finish kernels commit, 0 = they revert
  double mix_c, mix_lnc;
  int32_t napply, pad0;
  GphApply apply[GPH_MAXK + GPH_MAXB];
  // ---- dimensions and settings
  int32_t n, K, Kc, B, rootPop, startMig, doMixing, samplesPerLog;
  double Ltot;
  double thetaAlpha[GPH_MAXK], thetaBeta[GPH_MAXK], thetaStart[GPH_MAXK];
  double ageAlpha[GPH_MAXK], ageBeta[GPH_MAXK], ageStart[GPH_MAXK], ftTaus[GPH_MAXK];
  double mrAlpha[GPH_MAXB], mrBeta[GPH_MAXB];
  int32_t updateSampleAge[GPH_MAXK];
  double ftCoalTime, ftMigTime, ftTheta, ftMigRate, ftMixing;
  // ---- running state
  uint32_t gx, gy, gz, pad1;         // general RNG slot
  double logLikelihood, dataLogLikelihood;
  double tot_coal[GPH_MAXK], tot_ncoal[GPH_MAXK], tot_mig[GPH_MAXB], tot_nmig[GPH_MAXB];
  int64_t acc[9], accTau[GPH_MAXK], rubberband_conflicts;
  int32_t iteration, error;          // error: first fatal code seen by a stage (0 = none)
  long long error_locus;             // global index of the first locus that reported it (-1: not a per-locus error)
  // pending proposal bookkeeping (between propose and decide)
  double pend_lnacc, pend_tauold, pend_taunew, pend_taufactor0, pend_taufactor1, pend_dGen;
  int32_t pend_pop, pend_kind;       // kind 0 none, 1 tau, 2 sample age: the model change applied AFTER the finish kernel
  int32_t accArr[GPH_MAXK];          // accepted[] of the running UpdateTau / UpdateSampleAge call
  // per kernel class: evaluations, recomputed nodes, algorithmic bytes, "not enough migration slots"
  double cls_evals[16], cls_nodes[16], cls_bytes[16], cnt_notenough;
  // migration rates as recordParamVals saw them (GPhoCS.c:1730 runs BEFORE sampleMigRates at iteration == start-mig)
  double migRateShown[GPH_MAXB];
  // record lines of the running iteration (printed by the host after its synchronisation)
  int32_t nrec, shownValid;
  GphRec rec[GPH_REC_MAX];
  GphTauFin fin;                     // the decided proposal whose commit / revert has not run yet
};
