// gph_input.cpp -- host front end of the engine: the reference's input formats, unchanged.
//
//   control file   readControlFile / readSecondaryControlFile / checkSettings /
//                  finalizeNumParameters        (MCMCcontrol.c:118-463, 575-1478)
//   sequence file  readSeqFile / readSeqs / processLocusAlignment / cannonizeJCpattern /
//                  computeHetSymmetryBreaks / processHetPatterns / getAllPhases
//                                               (AlignmentProcessor.c:468-1158, 1595-1894, 2242-2339)
//   rate file      readRateFile                 (GPhoCS.c:491-579)
//
// The output of gph_loci_read is exactly what gph_engine_load_loci takes: per locus the PHASED
// pattern table in the reference's order (unphased patterns by first occurrence along the
// alignment, the phases of one pattern consecutive in getAllPhases' binary-counter order).  That
// order fixes the floating-point summation order of the root reduction, so it is part of parity.
//
// Differences in HOW (results identical):
//   * no global pattern table and no linear findPattern scan (AlignmentProcessor.c:1490): a locus
//     only needs its own pattern list, de-duplicated with a hash of the canonical column;
//   * JC canonicalisation keeps the set of surviving base permutations as a 24-bit mask and
//     looks the smallest image up in a [15][15] mask table instead of looping 2 x 24 times;
//   * loci are processed in parallel on host threads (the file is scanned once, sequentially,
//     only to find the locus boundaries).
// Host only: nothing here touches the GPU.
#include "../../include/gphocs_hip.h"
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

const char BASES[] = "TCAGYWKMSRVDBHN";   // cannonizedBaseSymbols, AlignmentProcessor.c:61
enum { NO_BASE = 0, NUCLEOTIDE, PARTIAL_AMBIG, COMPLETE_AMBIG };

int base_index(char c)
{
  const char *p = c ? strchr(BASES, c) : nullptr;
  return p ? (int)(p - BASES) : -1;
}
int base_type_idx(int idx)   // getBaseType, AlignmentProcessor.c:1467-1476
{
  if (idx < 0) return NO_BASE;
  if (idx < 4) return NUCLEOTIDE;
  if (idx < 14) return PARTIAL_AMBIG;
  return COMPLETE_AMBIG;
}

// the 24 base permutations and their images on the ambiguity codes
// (initializeBaseTransformations, AlignmentProcessor.c:1518-1590)
struct JcTables {
  int8_t trans[24][15];
  uint32_t mask[15][15];   // mask[base][image] = permutations mapping base -> image
  JcTables()
  {
    int perm = 0;
    // same enumeration order as the reference's table
    static const int8_t P[24][4] = {
        {0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 2, 1}, {0, 3, 1, 2},
        {1, 0, 2, 3}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 2, 0}, {1, 3, 0, 2},
        {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 0, 1, 3}, {2, 0, 3, 1}, {2, 3, 0, 1}, {2, 3, 1, 0},
        {3, 1, 2, 0}, {3, 1, 0, 2}, {3, 0, 2, 1}, {3, 0, 1, 2}, {3, 2, 0, 1}, {3, 2, 1, 0}};
    memset(mask, 0, sizeof mask);
    for (perm = 0; perm < 24; perm++) {
      for (int b = 0; b < 4; b++) trans[perm][b] = P[perm][b];
      trans[perm][14] = 14;
      for (int b1 = 0; b1 < 4; b1++) {
        trans[perm][b1 + 10] = (int8_t)(trans[perm][b1] + 10);
        for (int b2 = b1 + 1; b2 < 4; b2++) {
          int ambig = 2 * b1 + b2 + 3;
          if (ambig == 10) ambig = 9;
          int m1 = std::min(trans[perm][b1], trans[perm][b2]), m2 = std::max(trans[perm][b1], trans[perm][b2]);
          int am = 2 * m1 + m2 + 3;
          if (am == 10) am = 9;
          trans[perm][ambig] = (int8_t)am;
        }
      }
      for (int b = 0; b < 15; b++) mask[b][trans[perm][b]] |= 1u << perm;
    }
  }
};
const JcTables &jc()
{
  static const JcTables t;
  return t;
}

// cannonizeJCpattern, AlignmentProcessor.c:1595-1652: sample by sample, the smallest image over
// the surviving permutations; permutations disagreeing with the choice die
inline void canonize(const uint8_t *colIdx, uint8_t *patIdx, int n)
{
  const JcTables &T = jc();
  uint32_t living = 0xffffffu;
  for (int s = 0; s < n; s++) {
    const int b = colIdx[s];
    int map = 0;
    while (!(living & T.mask[b][map])) map++;
    living &= T.mask[b][map];
    patIdx[s] = (uint8_t)map;
  }
}

struct Reader {   // the reference's FILE* token/line discipline over an in-memory copy
  std::string buf;
  size_t pos = 0;
  bool eof = false;
  bool load(const char *path)
  {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = sz > 0 ? fread(&buf[0], 1, (size_t)sz, f) : 0;
    buf.resize(got);
    fclose(f);
    return true;
  }
  // fscanf("%s")
  bool scan(std::string &tok)
  {
    tok.clear();
    while (pos < buf.size() && isspace((unsigned char)buf[pos])) pos++;
    if (pos >= buf.size()) { eof = true; return false; }
    size_t b = pos;
    while (pos < buf.size() && !isspace((unsigned char)buf[pos])) pos++;
    tok.assign(buf, b, pos - b);
    return true;
  }
  // fgets: the rest of the current line, newline included
  bool line(std::string &out)
  {
    out.clear();
    if (pos >= buf.size()) { eof = true; return false; }
    size_t b = pos;
    while (pos < buf.size() && buf[pos] != '\n') pos++;
    if (pos < buf.size()) pos++;
    out.assign(buf, b, pos - b);
    return true;
  }
  // getNextToken, MCMCcontrol.c:476-491: next string, lines starting with '#' skipped
  std::string next()
  {
    std::string t, dummy;
    if (!scan(t)) return "EOF";
    while (!eof && !t.empty() && t[0] == '#') {
      line(dummy);
      if (!scan(t)) break;
    }
    if (eof) return "EOF";
    return t;
  }
};

// strtokCS over " \t\n": tokens of a line up to the first comment (utils.c:695-703)
std::vector<std::string> split_cs(const std::string &line)
{
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\n')) i++;
    if (i >= line.size()) break;
    size_t b = i;
    while (i < line.size() && !(line[i] == ' ' || line[i] == '\t' || line[i] == '\n')) i++;
    if (line[b] == '#') break;
    out.emplace_back(line, b, i - b);
  }
  return out;
}

bool parse_double(const std::string &s, double *v) { return sscanf(s.c_str(), "%lf", v) == 1; }
bool parse_int(const std::string &s, int *v) { return sscanf(s.c_str(), "%d", v) == 1; }

}   // namespace

// -------------------------------------------------------------------------------------------
struct gph_control {
  // GENERAL-INFO (initGeneralInfo defaults, MCMCcontrol.c:66-111)
  std::string seqFile = "NONE", rateFile = "NONE", traceFile = "mcmc-trace.out", nodeStatsFile = "NONE",
              combStatsFile = "NONE";
  int numLoci = -1, randomSeed = -1, burnin = 0, numSamples = 10000, sampleSkip = 0, startMig = 0, doMixing = 1;
  int samplesPerLog = 100, logsPerLine = 100, mutRateMode = 0, findFinetunes = 0, ffSteps = 100, ffSamples = 100;
  int numPopPartitions = 0;
  double varRatesAlpha = 0.0;
  double gPrint = 1.0, gAlpha = -1.0, gBeta = -1.0, gMigPrint = 1.0, gMigAlpha = -1.0, gMigBeta = -1.0, gFtTaus = -1.0;
  double ftCoalTime = -1.0, ftMigTime = -1.0, ftTheta = -1.0, ftMigRate = -1.0, ftLocusRate = -1.0, ftMixing = -1.0;
  // population tree
  int Kc = 0, K = 0, B = 0, rootPop = -1;
  std::vector<std::string> popName;
  std::vector<int32_t> father, son0, son1, updateSampleAge, samplesPerPop;
  std::vector<double> sampleAge, thetaAlpha, thetaBeta, thetaStart, ageAlpha, ageBeta, ageStart, ftTaus;
  std::vector<std::vector<uint8_t>> anc;   // isAncestralTo
  std::vector<int32_t> bandSrc, bandTgt;
  std::vector<double> mrAlpha, mrBeta, printFactors;
  int numParameters = 0;
  // samples (haploid slots; the second slot of a diploid has an empty name)
  std::vector<std::string> sampleNames;
  std::vector<uint8_t> isDiploid;
  bool pops_read = false;
  int pop_by_name(const std::string &nm) const
  {
    for (int p = 0; p < (int)popName.size(); p++)
      if (popName[p] == nm) return p;
    return -1;
  }
};

namespace {

int expect_token(Reader &r, const char *expected, std::string &tok)   // expectNextToken, MCMCcontrol.c:502-527
{
  int errs = 0;
  while ((tok = r.next()) != expected) {
    if (r.eof) break;
    errs++;
    fprintf(stderr, "Error: unexpected token '%s' before '%s'. Will Ignore this.\n", tok.c_str(), expected);
  }
  return errs;
}

int count_tokens(Reader &r, const char *countTok, const char *endTok)   // countTokens, MCMCcontrol.c:539-560
{
  size_t save = r.pos;
  bool save_eof = r.eof;
  int count = 0;
  while (!r.eof) {
    std::string t = r.next();
    if (t == endTok) break;
    if (t == countTok) count++;
  }
  r.pos = save;
  r.eof = save_eof;
  return count;
}

#define BADVAL(what, kind) do { fprintf(stderr, "Error: value for %s should be %s, got %s.\n", what, kind, v.c_str()); errs++; } while (0)

int read_general(gph_control &c, Reader &r)   // readGeneralInfo, MCMCcontrol.c:575-789
{
  int errs = 0;
  std::string tok, ln;
  errs += expect_token(r, "GENERAL-INFO-START", tok);
  if (errs) return errs;
  for (;;) {
    tok = r.next();
    if (tok == "GENERAL-INFO-END") break;
    if (!r.line(ln)) {
      fprintf(stderr, "Error: unexpected end of file or other error inside GENERAL-INFO module.\n");
      return errs + 1;
    }
    std::vector<std::string> t = split_cs(ln);
    if (t.empty()) { fprintf(stderr, "Error: unable to read value for %s.\n", tok.c_str()); errs++; continue; }
    const std::string &v = t[0];
    if (tok == "seq-file") c.seqFile = v;
    else if (tok == "trace-file") c.traceFile = v;
    else if (tok == "coal-stats-file") c.nodeStatsFile = v;
    else if (tok == "comb-stats-file") c.combStatsFile = v;
    else if (tok == "num-pop-partitions") { if (!parse_int(v, &c.numPopPartitions) || c.numPopPartitions <= 0) BADVAL("num-pop-partitions", "positive integer"); }
    else if (tok == "num-loci") { if (!parse_int(v, &c.numLoci) || c.numLoci <= 0) BADVAL("num-loci", "positive integer"); }
    else if (tok == "random-seed") { if (!parse_int(v, &c.randomSeed)) BADVAL("random-seed", "integer"); }
    else if (tok == "burn-in") { if (!parse_int(v, &c.burnin) || c.burnin < 0) BADVAL("burnin", "non-negative integer"); }
    else if (tok == "mcmc-iterations") { if (!parse_int(v, &c.numSamples) || c.numSamples <= 0) BADVAL("mcmc-iterations", "positive integer"); }
    else if (tok == "mcmc-sample-skip") { if (!parse_int(v, &c.sampleSkip) || c.sampleSkip < 0) BADVAL("mcmc-sample-skip", "non-negative integer"); }
    else if (tok == "start-mig") { if (!parse_int(v, &c.startMig) || c.startMig < 0) BADVAL("start-mig", "non-negative integer"); }
    else if (tok == "no-mixing") c.doMixing = 0;
    else if (tok == "iterations-per-log") { if (!parse_int(v, &c.samplesPerLog)) BADVAL("iterations-per-log", "integer"); }
    else if (tok == "logs-per-line") { if (!parse_int(v, &c.logsPerLine)) BADVAL("logs-per-line", "integer"); }
    else if (tok == "tau-theta-print") { if (!parse_double(v, &c.gPrint)) BADVAL("tau-theta-print", "floating point number"); }
    else if (tok == "tau-theta-alpha") { if (!parse_double(v, &c.gAlpha)) BADVAL("tau-theta-alpha", "floating point number"); }
    else if (tok == "tau-theta-beta") { if (!parse_double(v, &c.gBeta)) BADVAL("tau-theta-beta", "floating point number"); }
    else if (tok == "mig-rate-print") { if (!parse_double(v, &c.gMigPrint)) BADVAL("mig-rate-print", "floating point number"); }
    else if (tok == "mig-rate-alpha") { if (!parse_double(v, &c.gMigAlpha)) BADVAL("mig-rate-alpha", "floating point number"); }
    else if (tok == "mig-rate-beta") { if (!parse_double(v, &c.gMigBeta)) BADVAL("mig-rate-beta", "floating point number"); }
    else if (tok == "locus-mut-rate") {
      if (v == "CONST") c.mutRateMode = 0;
      else if (v == "FIXED") {
        if (t.size() < 2) { fprintf(stderr, "Error: unable to read filename for fixed locus mutation rates.\n"); errs++; continue; }
        c.rateFile = t[1];
        c.mutRateMode = 2;
      } else if (v == "VAR") {
        if (t.size() < 2 || !parse_double(t[1], &c.varRatesAlpha)) {
          fprintf(stderr, "Error: unable to read floating point alpha parameter for Dirichlet prior of mutation rate variation in locus-mut-rate.\n");
          errs++;
        }
        c.mutRateMode = 1;
      } else { fprintf(stderr, "Error: value of const-rate should be CONST, FIXED, or VAR, got %s.\n", v.c_str()); errs++; }
    }
    else if (tok == "finetune-coal-time") { if (!parse_double(v, &c.ftCoalTime)) BADVAL("finetune-coal-time", "floating point number"); }
    else if (tok == "finetune-mig-time") { if (!parse_double(v, &c.ftMigTime)) BADVAL("finetune-mig-time", "floating point number"); }
    else if (tok == "finetune-theta") { if (!parse_double(v, &c.ftTheta)) BADVAL("finetune-theta", "floating point number"); }
    else if (tok == "finetune-mig-rate") { if (!parse_double(v, &c.ftMigRate)) BADVAL("finetune-mig-rate", "floating point number"); }
    else if (tok == "finetune-tau") { if (!parse_double(v, &c.gFtTaus)) BADVAL("finetune-tau", "floating point number"); }
    else if (tok == "finetune-locus-rate") { if (!parse_double(v, &c.ftLocusRate)) BADVAL("finetune-locus-rate", "floating point number"); }
    else if (tok == "finetune-mixing") { if (!parse_double(v, &c.ftMixing)) BADVAL("finetune-mixing", "floating point number"); }
    else if (tok == "find-finetunes") {
      if (v == "TRUE") c.findFinetunes = 1;
      else if (v != "FALSE") { fprintf(stderr, "Error: value of find-finetunes should be TRUE or FALSE, got '%s'.\n", v.c_str()); errs++; }
    }
    else if (tok == "find-finetunes-num-steps") { if (!parse_int(v, &c.ffSteps) || c.ffSteps <= 0) BADVAL("find-finetunes-num-steps", "positive integer"); }
    else if (tok == "find-finetunes-samples-per-step") { if (!parse_int(v, &c.ffSamples) || c.ffSamples <= 0) BADVAL("find-finetunes-samples-per-step", "positive integer"); }
    else { fprintf(stderr, "Error: argument '%s' is not accepted in GENERAL-INFO module.\n", tok.c_str()); errs++; }
  }
  return errs;
}

// readSampleLine, MCMCcontrol.c:1271-1362: "name h|d" pairs; a diploid takes two haploid slots,
// the second one nameless
int read_sample_line(gph_control &c, const std::string &ln, int pop)
{
  int errs = 0;
  if (c.samplesPerPop[pop] > 0) { fprintf(stderr, "Error: observed two sample lists for pop %d.\n", pop + 1); return 1; }
  std::vector<std::string> t = split_cs(ln);
  if (t.size() % 2) { fprintf(stderr, "Error: uneven terms in sample line for pop %d.\n", pop + 1); errs++; }
  for (size_t i = 0; i + 1 < t.size(); i += 2) {
    const std::string &fmt = t[i + 1];
    if (fmt.size() != 1 || (fmt[0] != 'h' && fmt[0] != 'd')) {
      fprintf(stderr, "Error: faulty format %s for sample pop %d. Expected h or d.\n", fmt.c_str(), pop + 1);
      errs++;
    }
    c.sampleNames.push_back(t[i]);
    c.samplesPerPop[pop]++;
    if (fmt[0] == 'd') { c.sampleNames.push_back(""); c.samplesPerPop[pop]++; }
  }
  if (c.samplesPerPop[pop] < 1) { fprintf(stderr, "Error: no samples provided for pop %d.\n", pop + 1); errs++; }
  return errs;
}

void alloc_pops(gph_control &c, int Kc)   // createPopTree, PopulationTree.c:60-140
{
  c.Kc = Kc;
  c.K = 2 * Kc - 1;
  const int K = c.K;
  c.popName.assign(K, "");
  c.father.assign(K, -1); c.son0.assign(K, -1); c.son1.assign(K, -1);
  c.updateSampleAge.assign(K, 0);
  c.samplesPerPop.assign(Kc, 0);
  c.sampleAge.assign(K, 0.0);
  c.thetaAlpha.assign(K, 0.0); c.thetaBeta.assign(K, 0.0); c.thetaStart.assign(K, 0.0);
  c.ageAlpha.assign(K, 0.0); c.ageBeta.assign(K, 0.0); c.ageStart.assign(K, 0.0);
  c.ftTaus.assign(K, -1.0);
  c.anc.assign(K, std::vector<uint8_t>(K, 0));
  c.printFactors.assign(3 * Kc - 2, 0.0);
}

int read_current_pops(gph_control &c, Reader &r)   // readCurrentPops, MCMCcontrol.c:800-937
{
  int errs = 0;
  std::string tok, ln;
  errs += expect_token(r, "CURRENT-POPS-START", tok);
  if (r.eof) { fprintf(stderr, "Error: unexpected end of file before CURRENT-POPS-START.\n"); return errs + 1; }
  int curPops = count_tokens(r, "POP-START", "CURRENT-POPS-END");
  if (curPops <= 0) { fprintf(stderr, "Error: could not find any POP items in CURRENT-POPS module.\n"); return errs + 1; }
  alloc_pops(c, curPops);
  for (int pop = 0; pop < c.Kc; pop++) {
    errs += expect_token(r, "POP-START", tok);
    if (r.eof) { fprintf(stderr, "Error: unexpected end of file before POP-START.\n"); return errs + 1; }
    c.thetaAlpha[pop] = c.gAlpha;
    c.thetaBeta[pop] = c.gBeta;
    c.anc[pop][pop] = 1;
    c.printFactors[pop] = c.gPrint;
    for (;;) {
      tok = r.next();
      if (tok == "POP-END") break;
      if (!r.line(ln)) { fprintf(stderr, "Error: unexpected end of file or other error inside POP module.\n"); return errs + 1; }
      if (tok == "samples") { errs += read_sample_line(c, ln, pop); continue; }
      std::vector<std::string> t = split_cs(ln);
      if (t.empty()) { fprintf(stderr, "Error: unable to read value for %s.\n", tok.c_str()); errs++; continue; }
      const std::string &v = t[0];
      if (tok == "name") c.popName[pop] = v;
      else if (tok == "theta-print") { if (!parse_double(v, &c.printFactors[pop])) BADVAL("POP theta-print", "floating point number"); }
      else if (tok == "theta-alpha") { if (!parse_double(v, &c.thetaAlpha[pop])) BADVAL("POP theta-alpha", "floating point number"); }
      else if (tok == "theta-beta") { if (!parse_double(v, &c.thetaBeta[pop])) BADVAL("POP theta-beta", "floating point number"); }
      else if (tok == "age") {
        if (!parse_double(v, &c.sampleAge[pop])) BADVAL("POP age", "floating point number");
        const std::string f = t.size() > 1 ? t[1] : "";
        if (f.size() != 1 || (f[0] != 'f' && f[0] != 'e')) {
          fprintf(stderr, "Error: POP age can be set to fixed (f) or estimated (e), not %s.\n", f.c_str());
          errs++;
        }
        if (!f.empty() && f[0] == 'f') {
          c.updateSampleAge[pop] = 0;
          if (c.sampleAge[pop] != 0.0) c.doMixing = 0;   /* MCMCcontrol.c:905-907 */
        } else {
          c.updateSampleAge[pop] = 1;
        }
      } else {
        fprintf(stderr, "Error: argument '%s' is not accepted in POP module of CURRENT-POPS.\n", tok.c_str());
        errs++;
      }
    }
    if (c.popName[pop].empty()) { fprintf(stderr, "Error: no name is given for current pop %d.\n", pop + 1); errs++; }
  }
  errs += expect_token(r, "CURRENT-POPS-END", tok);
  if (r.eof) { fprintf(stderr, "Error: unexpected end of file before CURRENT-POPS-END.\n"); errs++; }
  // parseSampleNames, MCMCcontrol.c:1373-1470 (no admixture): a sample may be listed once
  for (size_t s = 0; s < c.sampleNames.size(); s++) {
    if (c.sampleNames[s].empty()) continue;
    for (size_t s1 = s + 1; s1 < c.sampleNames.size(); s1++)
      if (c.sampleNames[s] == c.sampleNames[s1]) {
        fprintf(stderr, "Error: admixture is not allowed, so sample %s cannot appear more than once in control file.\n",
                c.sampleNames[s].c_str());
        errs++;
        break;
      }
  }
  c.pops_read = true;
  return errs;
}

int read_ancestral_pops(gph_control &c, Reader &r)   // readAncestralPops, MCMCcontrol.c:945-1113
{
  int errs = 0;
  std::string tok, ln;
  errs += expect_token(r, "ANCESTRAL-POPS-START", tok);
  if (r.eof) { fprintf(stderr, "Error: unexpected end of file before ANCESTRAL-POPS-START.\n"); return errs + 1; }
  for (int pop = 0; pop < c.K; pop++) c.ftTaus[pop] = c.gFtTaus;
  for (int pop = c.Kc; pop < c.K; pop++) {
    errs += expect_token(r, "POP-START", tok);
    if (r.eof) {
      fprintf(stderr, "Error: could not find module POP for ancestral population %d (expecting %d populations total).\n", pop + 1, c.K);
      return errs + 1;
    }
    c.thetaAlpha[pop] = c.gAlpha; c.thetaBeta[pop] = c.gBeta;
    c.ageAlpha[pop] = c.gAlpha; c.ageBeta[pop] = c.gBeta; c.ageStart[pop] = -1;
    c.anc[pop][pop] = 1;
    c.printFactors[pop] = c.gPrint;
    c.printFactors[pop + c.Kc - 1] = c.gPrint;
    for (;;) {
      tok = r.next();
      if (tok == "POP-END") break;
      if (!r.line(ln)) { fprintf(stderr, "Error: unexpected end of file or other error inside POP module.\n"); return errs + 1; }
      std::vector<std::string> t = split_cs(ln);
      if (t.empty()) { fprintf(stderr, "Error: unable to read value for %s.\n", tok.c_str()); errs++; continue; }
      const std::string &v = t[0];
      if (tok == "name") c.popName[pop] = v;
      else if (tok == "children") {
        for (int son = 0; son < 2; son++) {
          if ((int)t.size() <= son) {
            if (son < 2 && (int)t.size() < 2) { fprintf(stderr, "Error: second child of ancestral pop %d is missing.\n", pop + 1); errs++; }
            break;
          }
          int p1 = c.pop_by_name(t[son]);
          if (p1 < 0) {
            fprintf(stderr, "Error: pop child name '%s' unrecognized for ancestral pop %d.\n", t[son].c_str(), pop + 1);
            errs++;
            continue;
          }
          (son ? c.son1 : c.son0)[pop] = p1;
          if (c.father[p1] >= 0) {
            fprintf(stderr, "Error: population %s already has a parent defined already (%d in addition to %d).\n",
                    c.popName[p1].c_str(), c.father[p1] + 1, pop + 1);
            errs++;
          } else c.father[p1] = pop;
          for (int p2 = 0; p2 < c.K; p2++)
            if (c.anc[p1][p2]) c.anc[pop][p2] = 1;
        }
      }
      else if (tok == "theta-print") { if (!parse_double(v, &c.printFactors[pop])) BADVAL("POP theta-print", "floating point number"); }
      else if (tok == "theta-alpha") { if (!parse_double(v, &c.thetaAlpha[pop])) BADVAL("POP theta-alpha", "floating point number"); }
      else if (tok == "theta-beta") { if (!parse_double(v, &c.thetaBeta[pop])) BADVAL("POP theta-beta", "floating point number"); }
      else if (tok == "tau-print") { if (!parse_double(v, &c.printFactors[c.Kc + pop - 1])) BADVAL("POP tau-print", "floating point number"); }
      else if (tok == "tau-alpha") { if (!parse_double(v, &c.ageAlpha[pop])) BADVAL("POP tau-alpha", "floating point number"); }
      else if (tok == "tau-beta") { if (!parse_double(v, &c.ageBeta[pop])) BADVAL("POP tau-beta", "floating point number"); }
      else if (tok == "tau-initial") { if (!parse_double(v, &c.ageStart[pop])) BADVAL("POP tau-initial", "floating point number"); }
      else if (tok == "finetune-tau") { if (!parse_double(v, &c.ftTaus[pop])) BADVAL("POP finetune-tau", "floating point number"); }
      else { fprintf(stderr, "Error: argument '%s' is not accepted in POP module of ANCESTRAL-POPS.\n", tok.c_str()); errs++; }
    }
    if (c.popName[pop].empty()) { fprintf(stderr, "Error: no name is given for ancestral pop %d.\n", pop + 1); errs++; }
    if (c.son0[pop] < 0) { fprintf(stderr, "Error: son #1 is not set for ancestral pop %d.\n", pop + 1); errs++; }
    if (c.son1[pop] < 0) { fprintf(stderr, "Error: son #2 is not set for ancestral pop %d.\n", pop + 1); errs++; }
  }
  c.rootPop = c.K - 1;
  errs += expect_token(r, "ANCESTRAL-POPS-END", tok);
  if (r.eof) { fprintf(stderr, "Error: unexpected end of file before ANCESTRAL-POPS-END.\n"); errs++; }
  return errs;
}

int read_mig_bands(gph_control &c, Reader &r)   // readMigrationBands, MCMCcontrol.c:1124-1256
{
  int errs = 0;
  std::string tok, ln;
  errs += expect_token(r, "MIG-BANDS-START", tok);
  if (r.eof) return errs;   /* the module is optional */
  c.B = count_tokens(r, "BAND-START", "MIG-BANDS-END");
  c.bandSrc.assign(c.B, -1); c.bandTgt.assign(c.B, -1);
  c.mrAlpha.assign(c.B, 0.0); c.mrBeta.assign(c.B, 0.0);
  c.printFactors.resize(3 * c.Kc - 2 + c.B, 0.0);
  for (int b = 0; b < c.B; b++) {
    errs += expect_token(r, "BAND-START", tok);
    if (r.eof) { fprintf(stderr, "Error: unexpected end of file before satrt of mig band %d.\n", b); return errs; }
    c.mrAlpha[b] = c.gMigAlpha; c.mrBeta[b] = c.gMigBeta;
    c.printFactors[3 * c.Kc - 2 + b] = c.gMigPrint;
    for (;;) {
      tok = r.next();
      if (tok == "BAND-END") break;
      if (!r.line(ln)) { fprintf(stderr, "Error: unexpected end of file or other error inside BAND module.\n"); return errs + 1; }
      std::vector<std::string> t = split_cs(ln);
      if (t.empty()) { fprintf(stderr, "Error: unable to read value for %s.\n", tok.c_str()); errs++; continue; }
      const std::string &v = t[0];
      if (tok == "source") {
        int p = c.pop_by_name(v);
        if (p < 0) { fprintf(stderr, "Error: invalid name '%s' for source pop of mig-band %d.\n", v.c_str(), b + 1); errs++; }
        else c.bandSrc[b] = p;
      } else if (tok == "target") {
        int p = c.pop_by_name(v);
        if (p < 0) { fprintf(stderr, "Error: invalid name '%s' for target pop of mig-band %d.\n", v.c_str(), b + 1); errs++; }
        else c.bandTgt[b] = p;
      }
      else if (tok == "mig-rate-print") { if (!parse_double(v, &c.printFactors[3 * c.Kc - 2 + b])) BADVAL("mig-rate-print", "floating point number"); }
      else if (tok == "mig-rate-alpha") { if (!parse_double(v, &c.mrAlpha[b])) BADVAL("mig-rate-alpha", "floating point number"); }
      else if (tok == "mig-rate-beta") { if (!parse_double(v, &c.mrBeta[b])) BADVAL("mig-rate-beta", "floating point number"); }
      else { fprintf(stderr, "Error: argument '%s' is not accepted in BAND module.\n", tok.c_str()); errs++; }
    }
    if (c.bandSrc[b] == -1) { fprintf(stderr, "Error: source population for migration band %d was not defined.\n", b + 1); return errs + 1; }
    if (c.bandTgt[b] == -1) { fprintf(stderr, "Error: target population for migration band %d was not defined.\n", b + 1); return errs + 1; }
    if (c.anc[c.bandSrc[b]][c.bandTgt[b]]) {
      fprintf(stderr, "Error: source pop for migration band %d is an ancestor of its target pop.\n\t\tMigration bands can only be placed between two populations which may have co-occured.\n", b + 1);
      return errs + 1;
    }
    if (c.anc[c.bandTgt[b]][c.bandSrc[b]]) {
      fprintf(stderr, "Error: target pop for migration band %d is an ancestor of its source pop.\n\t\tMigration bands can only be placed between two populations which may have co-occured.\n", b + 1);
      return errs + 1;
    }
  }
  errs += expect_token(r, "MIG-BANDS-END", tok);
  if (r.eof) { fprintf(stderr, "Error: unexpected end of file before MIG-BANDS-END.\n"); errs++; }
  tok = r.next();
  while (!r.eof) {
    errs++;
    fprintf(stderr, "Error: ignoring token '%s' after MIG-BANDS.\n", tok.c_str());
    tok = r.next();
  }
  return errs;
}

int check_settings(gph_control &c)   // checkSettings, MCMCcontrol.c:219-358
{
  int errs = 0;
  if (c.seqFile == "NONE") { fprintf(stderr, "Error: Sequences file ('seq-file') is not defined in the control file.\n"); errs++; }
  if (c.nodeStatsFile == "NONE" && c.numPopPartitions > 0) {
    fprintf(stderr, "Error: number of population partitions is %d, but no stats file name was specified.\n", c.numPopPartitions);
    errs++;
  }
  if (!c.findFinetunes) {
    if (c.ftCoalTime < 0.0) { fprintf(stderr, "Error: positive finetune for coal-time should be specified.\n"); errs++; }
    if (c.ftMigTime < 0.0) { fprintf(stderr, "Error: positive finetune for mig-time should be specified.\n"); errs++; }
    if (c.ftTheta < 0.0) { fprintf(stderr, "Error: positive finetune for theta should be specified.\n"); errs++; }
    if (c.ftMigRate < 0.0) { fprintf(stderr, "Error: positive finetune for mig-rate should be specified.\n"); errs++; }
    if (c.mutRateMode == 1 && c.ftLocusRate < 0.0) { fprintf(stderr, "Error: positive finetune for locus-rate should be specified.\n"); errs++; }
    if (c.ftMixing < 0.0) { fprintf(stderr, "Error: positive finetune for mixing should be specified.\n"); errs++; }
  }
  if (c.samplesPerLog <= 0) { fprintf(stderr, "Warning: samples-per-log must be 1 or greater, adjusting to 100.\n"); c.samplesPerLog = 100; }
  if (c.logsPerLine <= 0) { fprintf(stderr, "Warning: logs-per-line must be 1 or greater, adjusting to 100.\n"); c.logsPerLine = 100; }
  for (int pop = c.Kc; pop < c.K; pop++)
    if (c.ageStart[pop] <= 0) c.ageStart[pop] = c.ageAlpha[pop] / c.ageBeta[pop];
  for (int pop = 0; pop < c.K; pop++) {
    const char *nm = c.popName[pop].c_str();
    if (c.thetaAlpha[pop] < 0) { fprintf(stderr, "Error: gamma prior alpha parameter not set for theta of pop %s (%d).\n", nm, pop + 1); errs++; }
    if (c.thetaBeta[pop] < 0) { fprintf(stderr, "Error: gamma prior beta argument not set for theta of pop %s (%d).\n", nm, pop + 1); errs++; }
    c.thetaStart[pop] = c.thetaAlpha[pop] / c.thetaBeta[pop];
    if (pop >= c.Kc) {
      if (c.ageAlpha[pop] < 0) { fprintf(stderr, "Error: gamma prior alpha parameter not set for tau of ancestral pop %s (%d).\n", nm, pop + 1); errs++; }
      if (c.ageBeta[pop] < 0) { fprintf(stderr, "Error: gamma prior beta parameter not set for tau of ancestral pop %s (%d).\n", nm, pop + 1); errs++; }
      if (!c.findFinetunes && c.ftTaus[pop] < 0.0) { fprintf(stderr, "Error: finetune for tau of ancestral pop %s (%d) is not set.\n", nm, pop + 1); errs++; }
      if (c.rootPop != pop && c.father[pop] >= 0) {
        const int f = c.father[pop];
        if (c.ageAlpha[f] / c.ageBeta[f] < c.ageAlpha[pop] / c.ageBeta[pop]) {
          fprintf(stderr, "\nError:Conflicting prior for ancestral population ages found for pop %s, and parent pop %s.\n", nm, c.popName[f].c_str());
          errs++;
        }
        if (c.ageStart[f] < c.ageStart[pop]) {
          fprintf(stderr, "\nError:Conflicting initalization settings for ancestral population ages found for pop %s, and parent pop %s.\n", nm, c.popName[f].c_str());
          errs++;
        }
      }
    } else if (c.father[pop] >= 0) {
      const int f = c.father[pop];
      if (c.ageAlpha[f] / c.ageBeta[f] < c.sampleAge[pop]) {
        fprintf(stderr, "\nError:Conflicting prior for ancestral population age for parent pop %s and sample age for pop %s.\n", c.popName[f].c_str(), nm);
        errs++;
      }
      if (c.ageStart[f] < c.sampleAge[pop]) {
        fprintf(stderr, "\nError:Conflicting initialization for ancestral population age for parent pop %s and sample age for pop %s (%g,%g).\n",
                c.popName[f].c_str(), nm, c.ageStart[f], c.sampleAge[pop]);
        errs++;
      }
    }
  }
  for (int b = 0; b < c.B; b++) {
    if (c.mrAlpha[b] < 0) { fprintf(stderr, "Error: gamma prior alpha argument not set for mig-rate of mig-band (#%d).\n", b + 1); errs++; }
    if (c.mrBeta[b] < 0) { fprintf(stderr, "Error: gamma prior beta argument not set for mig-rate of mig-band (#%d).\n", b + 1); errs++; }
  }
  return errs;
}

void finalize_parameters(gph_control &c)   // finalizeNumParameters, MCMCcontrol.c:428-463
{
  int numAncient = 0;
  for (int p = 0; p < c.Kc; p++)
    if (c.updateSampleAge[p] || c.sampleAge[p] > 0.0) numAncient++;
  const int base = 2 * c.K - c.Kc + c.B;
  c.numParameters = base + numAncient + (c.mutRateMode == 1);
  c.printFactors.resize(c.numParameters, 0.0);
  for (int p = base; p < base + numAncient; p++) c.printFactors[p] = c.gPrint;
  for (int p = base + numAncient; p < c.numParameters; p++) c.printFactors[p] = 1.0;
}

}   // namespace

// -------------------------------------------------------------------------------------------
struct gph_loci {
  int64_t L = 0;
  int32_t n = 0;
  std::vector<int64_t> offsets;
  std::vector<uint8_t> leafcodes;
  std::vector<uint16_t> numPhases;
  std::vector<int32_t> counts, unphased;   // unphased[g] = patterns before phasing
  std::vector<double> mutRates;
  std::vector<std::string> names;
};

namespace {

struct RawLocus {
  std::string name;
  int seqLength = 0;
  std::vector<const char *> seq;   // per haploid slot, nullptr = absent
  std::string error;
};

struct LocusOut {
  std::vector<uint8_t> leaf;
  std::vector<uint16_t> phases;
  std::vector<int32_t> counts;
  int unphased = 0;
  std::string error;
};

// computeHetSymmetryBreaks, AlignmentProcessor.c:1706-1894: each diploid is phased arbitrarily at
// no more than one (singleton) column of the locus; greedy by score 2^(live hets)
void het_symmetry_breaks(const std::vector<std::vector<uint8_t>> &pat, const std::vector<int32_t> &cnt, int n,
                         std::vector<std::vector<uint8_t>> &breaks)
{
  const int np = (int)pat.size();
  std::vector<double> score(np, -1.0);
  std::vector<std::vector<int>> liveHets(np);
  std::vector<int> livePatterns, liveIndex(np, -1);
  breaks.assign(np, std::vector<uint8_t>(n, 0));
  double maxScore = -1.0;
  int chosen = -1;
  for (int p = 0; p < np; p++) {
    if (cnt[p] > 1) continue;
    for (int s = 0; s < n; s++) {
      if (base_type_idx(pat[p][s]) == PARTIAL_AMBIG) {
        liveHets[p].push_back(s);
        score[p] *= 2;
        if (liveHets[p].size() <= 1) {
          liveIndex[p] = (int)livePatterns.size();
          livePatterns.push_back(p);
          score[p] = 2.0;
        }
      }
    }
    if (maxScore < score[p]) { chosen = p; maxScore = score[p]; }
  }
  int numLive = (int)livePatterns.size();
  while (maxScore > 0.0) {
    const int sample = liveHets[chosen].back();
    liveHets[chosen].pop_back();
    breaks[chosen][sample] = 1;
    if (liveHets[chosen].empty()) score[chosen] = -1.0;
    else score[chosen] /= 2;
    maxScore = score[chosen];
    for (int i = 0; i < numLive;) {
      const int p1 = livePatterns[i];
      std::vector<int> &lh = liveHets[p1];
      for (size_t k = 0; k < lh.size(); k++)
        if (lh[k] == sample) { lh[k] = lh.back(); lh.pop_back(); break; }
      if (!lh.empty()) i++;
      else {
        numLive--;
        livePatterns[liveIndex[p1]] = livePatterns[numLive];
        liveIndex[livePatterns[liveIndex[p1]]] = liveIndex[p1];
        liveIndex[p1] = -1;
        score[p1] = -1.0;
      }
      if (maxScore < score[p1]) { maxScore = score[p1]; chosen = p1; }
    }
  }
}

// translateAmbiguity, AlignmentProcessor.c:2298-2339 (indices into BASES; 14 = N)
inline void translate_ambiguity(int idx, uint8_t *out)
{
  switch (idx) {
  case 4: out[0] = 0; out[1] = 1; break;    /* Y = T C */
  case 6: out[0] = 0; out[1] = 3; break;    /* K = T G */
  case 5: out[0] = 0; out[1] = 2; break;    /* W = T A */
  case 8: out[0] = 1; out[1] = 3; break;    /* S = C G */
  case 7: out[0] = 2; out[1] = 1; break;    /* M = A C */
  case 9: out[0] = 2; out[1] = 3; break;    /* R = A G */
  case 0: case 1: case 2: case 3: out[0] = out[1] = (uint8_t)idx; break;
  default: out[0] = out[1] = 14; break;
  }
}

void process_locus(const RawLocus &raw, int n, const std::vector<uint8_t> &isDiploid,
                   const std::vector<std::string> &sampleNames, LocusOut &out)
{
  // character checks of readSeqs, AlignmentProcessor.c:808-842
  for (int s = 0; s < n; s++) {
    if (!raw.seq[s]) continue;
    for (int site = 0; site < raw.seqLength; site++) {
      const char ch = (char)toupper((unsigned char)raw.seq[s][site]);
      char msg[256];
      if (isspace((unsigned char)ch)) {
        snprintf(msg, sizeof msg, "Whitespace found in site %d for sample %s. No whitespaces (tab, space, etc.) permitted inside sequences.", site + 1, sampleNames[s].c_str());
        out.error = msg;
        return;
      }
      const int bi = base_index(ch);
      if (bi < 0) {
        snprintf(msg, sizeof msg, "Illegal base type '%c' found in site %d of sample %s.", ch, site + 1, sampleNames[s].c_str());
        out.error = msg;
        return;
      }
      if (bi >= 4 && bi < 14 && !isDiploid[s]) {
        snprintf(msg, sizeof msg, "Ambiguity character '%c' found in site %d of haploid sample %s.", ch, site + 1, sampleNames[s].c_str());
        out.error = msg;
        return;
      }
    }
  }
  // processLocusAlignment, AlignmentProcessor.c:871-983
  std::vector<std::vector<uint8_t>> pats;
  std::vector<int32_t> cnt;
  std::unordered_map<std::string, int> seen;
  std::vector<uint8_t> col(n), pat(n);
  std::string key(n, '\0');
  for (int site = 0; site < raw.seqLength; site++) {
    bool notAllNs = false;
    for (int s = 0; s < n; s++) {
      if (!raw.seq[s]) col[s] = 14;
      else {
        col[s] = (uint8_t)base_index((char)toupper((unsigned char)raw.seq[s][site]));
        if (col[s] != 14) notAllNs = true;
      }
    }
    if (!notAllNs) continue;
    canonize(col.data(), pat.data(), n);
    key.assign((const char *)pat.data(), n);
    auto it = seen.find(key);
    if (it != seen.end()) cnt[it->second]++;
    else {
      seen.emplace(key, (int)pats.size());
      pats.push_back(pat);
      cnt.push_back(1);
    }
  }
  out.unphased = (int)pats.size();
  // processHetPatterns, AlignmentProcessor.c:998-1158 (breakSymmetries = 1)
  std::vector<std::vector<uint8_t>> breaks;
  het_symmetry_breaks(pats, cnt, n, breaks);
  std::vector<uint8_t> hap(n + 1), perturb(n + 2), cur(n + 1), nxt(n + 1);
  for (size_t p = 0; p < pats.size(); p++) {
    for (int s = 0; s < n; s++) {
      if (!isDiploid[s]) {
        perturb[s] = 0;
        hap[s] = pats[p][s];
      } else {
        perturb[s] = 0;
        translate_ambiguity(pats[p][s], &hap[s]);
        perturb[s + 1] = (base_type_idx(pats[p][s]) == PARTIAL_AMBIG && !breaks[p][s]) ? 1 : 0;
        s++;
      }
    }
    // getAllPhases, AlignmentProcessor.c:2242-2287: binary counter over the marked pairs
    int64_t numPhases = 1;
    for (int h = 0; h < n; h++)
      if (perturb[h]) numPhases *= 2;
    /* a count of 2^15 or more travels as 0x8000 | exponent in the 16-bit word (gph_types.h: GPH_PHASES; always a power of two
     * here).  Beyond 2^24 phased rows of ONE pattern the reference itself asks malloc for >= 64 N x 2^24 bytes (76 GB at 36
     * leaves, AlignmentProcessor.c:1068-1090): refused with the pattern named */
    if (numPhases > ((int64_t)1 << 24)) {
      char msg[200];
      snprintf(msg, sizeof msg, "pattern %zu has %lld phases (more than 24 unbroken heterozygotes in one repeated column): beyond 2^24 "
               "phased patterns for one column", p + 1, (long long)numPhases);
      out.error = msg;
      return;
    }
    const size_t row0 = out.phases.size();
    for (int h = 0; h < n; h++) cur[h] = hap[h];
    for (int64_t phase = 0; phase < numPhases; phase++) {
      if (phase > 0) {
        bool flip = true;
        for (int h = 0; h < n; h++) nxt[h] = cur[h];
        for (int h = 0; h < n; h++) {
          if (flip && perturb[h] > 0) {
            nxt[h] = cur[h - 1];
            nxt[h - 1] = cur[h];
            if (perturb[h] == 1) { perturb[h] = 2; flip = false; }
            else perturb[h] = 1;
          }
        }
        /* a flipped pair reads the PREVIOUS phase's values (cur), as the reference's
         * phasedColumns[phase-1] */
        cur.swap(nxt);
      }
      for (int h = 0; h < n; h++) {
        const uint8_t b = cur[h];
        out.leaf.push_back(b < 4 ? b : 4);   /* T,C,A,G = 0..3; everything else is N (LDL.c:1321) */
      }
      out.phases.push_back(0);
      out.counts.push_back(0);
    }
    {
      uint16_t w = (uint16_t)numPhases;
      if (numPhases >= 0x8000) { int ex = 0; while (((int64_t)1 << ex) < numPhases) ex++; w = (uint16_t)(0x8000 | ex); }
      out.phases[row0] = w;
    }
    out.counts[row0] = cnt[p];
  }
}

void set_err(char *err, int errlen, const std::string &msg)
{
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg.c_str());
}

}   // namespace

extern "C" {

int gph_control_read(const char *path, const char *secondary, gph_control **out)
{
  if (!path || !out) return GPH_EARG;
  *out = nullptr;
  gph_control *c = new gph_control();
  Reader r;
  if (!r.load(path)) { fprintf(stderr, "Error: Could not open control file '%s'.\n", path); delete c; return GPH_EARG; }
  int errs = read_general(*c, r);
  const char *where = "GENERAL-INFO";
  if (!errs) { errs += read_current_pops(*c, r); where = "CURRENT-POPS"; }
  if (!errs) { errs += read_ancestral_pops(*c, r); where = "ANCESTRAL-POPS"; }
  if (!errs) { errs += read_mig_bands(*c, r); where = "MIG-BANDS"; }
  if (errs) {
    fprintf(stderr, "Found %d errors when parsing %s in control file %s.\n", errs, where, path);
    delete c;
    return GPH_EARG;
  }
  if (secondary) {   // readSecondaryControlFile, MCMCcontrol.c:178-210: GENERAL-INFO then MIG-BANDS
    Reader r2;
    if (!r2.load(secondary)) { fprintf(stderr, "Error: Could not open secondary control file '%s'.\n", secondary); delete c; return GPH_EARG; }
    errs = read_general(*c, r2);
    if (!errs) errs += read_mig_bands(*c, r2);
    if (errs) { fprintf(stderr, "Found %d errors when parsing secondary control file %s.\n", errs, secondary); delete c; return GPH_EARG; }
  }
  errs = check_settings(*c);
  finalize_parameters(*c);
  if (errs) { fprintf(stderr, "Found %d errors when processing control settings.\n", errs); delete c; return GPH_EARG; }
  const int n = (int)c->sampleNames.size();
  c->isDiploid.assign(n, 0);   // initAlignmentData, AlignmentProcessor.c:226-235
  for (int s = 1; s < n; s++)
    if (c->sampleNames[s].empty()) c->isDiploid[s - 1] = c->isDiploid[s] = 1;
  *out = c;
  return GPH_OK;
}

void gph_control_free(gph_control *c) { delete c; }

int gph_control_get(const gph_control *c, gph_config *cfg, gph_mcmc_config *mc, gph_control_info *info)
{
  if (!c) return GPH_EARG;
  if (cfg) {
    memset(cfg, 0, sizeof *cfg);
    cfg->n = (int32_t)c->sampleNames.size();
    cfg->Kc = c->Kc; cfg->K = c->K; cfg->B = c->B; cfg->rootPop = c->rootPop;
    cfg->samplesPerPop = c->samplesPerPop.data();
    cfg->popFather = c->father.data(); cfg->popSon0 = c->son0.data(); cfg->popSon1 = c->son1.data();
    cfg->bandSrc = c->bandSrc.data(); cfg->bandTgt = c->bandTgt.data();
    cfg->L_total = c->numLoci;
  }
  if (mc) {
    memset(mc, 0, sizeof *mc);
    mc->thetaAlpha = c->thetaAlpha.data(); mc->thetaBeta = c->thetaBeta.data(); mc->thetaStart = c->thetaStart.data();
    mc->ageAlpha = c->ageAlpha.data(); mc->ageBeta = c->ageBeta.data(); mc->ageStart = c->ageStart.data();
    mc->sampleAge = c->sampleAge.data(); mc->updateSampleAge = c->updateSampleAge.data();
    mc->mrAlpha = c->mrAlpha.data(); mc->mrBeta = c->mrBeta.data();
    mc->ftCoalTime = c->ftCoalTime; mc->ftMigTime = c->ftMigTime; mc->ftTheta = c->ftTheta;
    mc->ftMigRate = c->ftMigRate; mc->ftMixing = c->ftMixing;
    mc->ftTaus = c->ftTaus.data();
    mc->seed = c->randomSeed; mc->startMig = c->startMig; mc->doMixing = c->doMixing;
    mc->samplesPerLog = c->samplesPerLog; mc->numParameters = c->numParameters;
    mc->printFactors = c->printFactors.data();
    mc->mutRateMode = c->mutRateMode; mc->varRatesAlpha = c->varRatesAlpha; mc->ftLocusRate = c->ftLocusRate;
  }
  if (info) {
    memset(info, 0, sizeof *info);
    info->seqFile = c->seqFile.c_str(); info->traceFile = c->traceFile.c_str(); info->rateFile = c->rateFile.c_str();
    info->numLoci = c->numLoci; info->burnin = c->burnin; info->numSamples = c->numSamples;
    info->sampleSkip = c->sampleSkip; info->logsPerLine = c->logsPerLine; info->mutRateMode = c->mutRateMode;
    info->findFinetunes = c->findFinetunes; info->findFinetunesNumSteps = c->ffSteps;
    info->findFinetunesSamplesPerStep = c->ffSamples; info->numSampleSlots = (int32_t)c->sampleNames.size();
    info->varRatesAlpha = c->varRatesAlpha; info->ftLocusRate = c->ftLocusRate;
  }
  return GPH_OK;
}

const char *gph_control_pop_name(const gph_control *c, int32_t pop)
{
  return c && pop >= 0 && pop < c->K ? c->popName[pop].c_str() : nullptr;
}
const char *gph_control_sample_name(const gph_control *c, int32_t slot)
{
  return c && slot >= 0 && slot < (int32_t)c->sampleNames.size() ? c->sampleNames[slot].c_str() : nullptr;
}

int gph_loci_read(const gph_control *c, const char *seq_path, int32_t threads, gph_loci **out, char *err, int32_t errlen)
{
  if (!c || !out) return GPH_EARG;
  *out = nullptr;
  const char *path = seq_path ? seq_path : c->seqFile.c_str();
  const int n = (int)c->sampleNames.size();
  Reader r;
  if (!r.load(path)) {
    set_err(err, errlen, std::string("Could not find sequence file '") + path + "'");
    return GPH_EARG;
  }
  // ---- sequential scan: readSeqFile / readSeqs, AlignmentProcessor.c:468-860
  std::string ln, tok;
  std::vector<std::string> t;
  do {
    if (!r.line(ln)) { set_err(err, errlen, "Unexpected End of File when trying to read number of loci from seq file"); return GPH_EARG; }
    t = split_cs(ln);
  } while (false);
  int numLoci = 0;
  if (t.empty()) { set_err(err, errlen, "Unexpected End of File when trying to read number of loci from seq file"); return GPH_EARG; }
  if (!parse_int(t[0], &numLoci)) { set_err(err, errlen, "Expected number of loci when reading sequence file, got " + t[0]); return GPH_EARG; }
  if (numLoci <= 0) { set_err(err, errlen, "At least one locus must be specified in the sequence file"); return GPH_EARG; }
  if (c->numLoci > 0 && c->numLoci < numLoci) numLoci = c->numLoci;   /* AlignmentProcessor.c:543-548 */
  std::vector<RawLocus> raw((size_t)numLoci);
  std::vector<uint8_t> sampleSeen(n, 0);
  for (int s = 0; s < n; s++)
    if (c->sampleNames[s].empty()) sampleSeen[s] = 1;
  const std::string &B = r.buf;
  for (int locus = 0; locus < numLoci; locus++) {
    RawLocus &q = raw[locus];
    t.clear();
    while (r.line(ln)) {
      t = split_cs(ln);
      if (!t.empty()) break;
    }
    char where[64];
    snprintf(where, sizeof where, " (locus %d)", locus + 1);
    if (t.empty()) {
      char msg[200];
      snprintf(msg, sizeof msg, "Sequence file says to use %d loci, but the sequence file only contains %d loci", numLoci, locus);
      set_err(err, errlen, msg);
      return GPH_EARG;
    }
    q.name = t[0];
    int numLocusSamples = 0;
    if (t.size() < 2 || !parse_int(t[1], &numLocusSamples)) { set_err(err, errlen, std::string("Expected number of locus samples") + where); return GPH_EARG; }
    if (numLocusSamples <= 0) { set_err(err, errlen, std::string("Every Locus must have one or more samples") + where); return GPH_EARG; }
    if (t.size() < 3 || !parse_int(t[2], &q.seqLength)) { set_err(err, errlen, std::string("Expected sequence length") + where); return GPH_EARG; }
    q.seq.assign(n, nullptr);
    for (int sq = 0; sq < numLocusSamples; sq++) {
      if (!r.scan(tok)) { set_err(err, errlen, std::string("Encountered unexpected EOF while reading sequences") + where); return GPH_EARG; }
      int idx = -1;
      for (int s = 0; s < n; s++)
        if (tok == c->sampleNames[s] && !tok.empty()) { idx = s; break; }
      if (idx < 0) { r.line(ln); continue; }   /* unknown sample: skipped, AlignmentProcessor.c:791-795 */
      while (r.pos < B.size() && isspace((unsigned char)B[r.pos])) r.pos++;
      const size_t b = r.pos;
      if (b + (size_t)q.seqLength > B.size()) { set_err(err, errlen, std::string("Unexpected EOF while reading sequence of sample ") + tok + where); return GPH_EARG; }
      {
        const void *nl = memchr(B.data() + b, '\n', (size_t)q.seqLength);
        if (nl) {
          char msg[256];
          snprintf(msg, sizeof msg, "Sequence for sample %s contained only %d bases instead of the expected %d bases as defined in the sequence file%s",
                   tok.c_str(), (int)((const char *)nl - (B.data() + b)), q.seqLength, where);
          set_err(err, errlen, msg);
          return GPH_EARG;
        }
      }
      r.pos = b + (size_t)q.seqLength;
      if (r.pos >= B.size() || !isspace((unsigned char)B[r.pos])) {   /* EOF right after the last base is an error upstream too */
        char msg[256];
        snprintf(msg, sizeof msg, "Sequence for sample %s might be too long than specified (%d bases). Found character %c at position %d%s",
                 tok.c_str(), q.seqLength, r.pos < B.size() ? B[r.pos] : '?', q.seqLength + 1, where);
        set_err(err, errlen, msg);
        return GPH_EARG;
      }
      if (r.pos < B.size() && B[r.pos] != '\n') r.line(ln);
      else if (r.pos < B.size()) r.pos++;
      q.seq[idx] = B.data() + b;
      sampleSeen[idx] = 1;
    }
  }
  for (int s = 0; s < n; s++)
    if (!sampleSeen[s]) {
      set_err(err, errlen, "Sample name '" + c->sampleNames[s] + "' was defined in the control file, but no samples for this name exist in the sequence file");
      return GPH_EARG;
    }
  // ---- per-locus processing on host threads
  std::vector<LocusOut> outs((size_t)numLoci);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, numLoci));
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int b = next.fetch_add(64);
      if (b >= numLoci) break;
      const int e = std::min(numLoci, b + 64);
      for (int g = b; g < e; g++) process_locus(raw[g], n, c->isDiploid, c->sampleNames, outs[g]);
    }
  };
  if (nt == 1) work();
  else {
    std::vector<std::thread> th;
    for (int i = 0; i < nt; i++) th.emplace_back(work);
    for (auto &x : th) x.join();
  }
  gph_loci *Lc = new gph_loci();
  Lc->L = numLoci;
  Lc->n = n;
  Lc->offsets.assign((size_t)numLoci + 1, 0);
  for (int g = 0; g < numLoci; g++) {
    if (!outs[g].error.empty()) {
      char msg[320];
      snprintf(msg, sizeof msg, "locus %d: %s", g + 1, outs[g].error.c_str());
      set_err(err, errlen, msg);
      delete Lc;
      return GPH_EARG;
    }
    Lc->offsets[g + 1] = Lc->offsets[g] + (int64_t)outs[g].phases.size();
  }
  const int64_t Ptot = Lc->offsets[numLoci];
  Lc->leafcodes.resize((size_t)Ptot * n);
  Lc->numPhases.resize((size_t)Ptot);
  Lc->counts.resize((size_t)Ptot);
  Lc->unphased.resize((size_t)numLoci);
  Lc->names.resize((size_t)numLoci);
  for (int g = 0; g < numLoci; g++) {
    const int64_t o = Lc->offsets[g];
    if (!outs[g].phases.empty()) {
      memcpy(&Lc->leafcodes[(size_t)o * n], outs[g].leaf.data(), outs[g].leaf.size());
      memcpy(&Lc->numPhases[(size_t)o], outs[g].phases.data(), outs[g].phases.size() * sizeof(uint16_t));
      memcpy(&Lc->counts[(size_t)o], outs[g].counts.data(), outs[g].counts.size() * sizeof(int32_t));
    }
    Lc->unphased[g] = outs[g].unphased;
    Lc->names[g] = raw[g].name;
  }
  // ---- readRateFile, GPhoCS.c:491-579 (locus-mut-rate FIXED): normalised to mean 1.  The message bodies are upstream's
  // ("Error: " + body on stderr there, followed by "Error: Unable to reading rate file '<name>'. Aborting !!", GPhoCS.c:1149-1154)
  Lc->mutRates.assign((size_t)numLoci, 1.0);
  if (c->mutRateMode == 2) {
    char m[192];
    FILE *f = fopen(c->rateFile.c_str(), "r");
    if (!f) { set_err(err, errlen, "Could not find/read rate file " + c->rateFile + "."); delete Lc; return GPH_EARG; }
    double sum = 0.0, tmp;
    for (int g = 0; g < numLoci; g++) {
      if (fscanf(f, "%lf", &Lc->mutRates[g]) != 1) { snprintf(m, sizeof m, "Cannot read rate for locus %d.", g + 1); set_err(err, errlen, m); fclose(f); delete Lc; return GPH_EARG; }
      sum += Lc->mutRates[g];
      if (Lc->mutRates[g] <= 0.0) { snprintf(m, sizeof m, "Locus %d has non-positive (%g) rate.", g + 1, Lc->mutRates[g]); set_err(err, errlen, m); fclose(f); delete Lc; return GPH_EARG; }
    }
    if (fscanf(f, "%lf", &tmp) == 1) {
      snprintf(m, sizeof m, "Rate file contains more than the %d loci specified in the sequence file.", numLoci);
      set_err(err, errlen, m); fclose(f); delete Lc; return GPH_EARG;
    }
    fclose(f);
    sum /= numLoci;
    for (int g = 0; g < numLoci; g++) Lc->mutRates[g] = Lc->mutRates[g] / sum;
  }
  *out = Lc;
  return GPH_OK;
}

void gph_loci_free(gph_loci *l) { delete l; }

int gph_loci_arrays(const gph_loci *l, int64_t *L, int32_t *n, const int64_t **pattern_offsets, const uint8_t **leafcodes,
                    const uint16_t **numPhases, const int32_t **counts, const double **mutRates, const int32_t **unphased)
{
  if (!l) return GPH_EARG;
  if (L) *L = l->L;
  if (n) *n = l->n;
  if (pattern_offsets) *pattern_offsets = l->offsets.data();
  if (leafcodes) *leafcodes = l->leafcodes.data();
  if (numPhases) *numPhases = l->numPhases.data();
  if (counts) *counts = l->counts.data();
  if (mutRates) *mutRates = l->mutRates.data();
  if (unphased) *unphased = l->unphased.data();
  return GPH_OK;
}

}   // extern "C"
