/*
 * gphocs_oracle_mcmc.c -- TEST INFRASTRUCTURE ONLY (see gphocs_oracle.h).
 *
 * The MCMC proposal functions of the reference restated serially over loci
 * (same order of RNG draws and of floating-point accumulation as the serial
 * reference build).  Citations are file:line under /root/reference/src.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "gphocs_oracle.h"
#include "gphocs_oracle_int.h"

#define GRND(s) go_rndu(&(s)->gx, &(s)->gy, &(s)->gz)
#define LRND(q) go_rndu(&(q)->rx, &(q)->ry, &(q)->rz)
#define max2(a, b) ((a) > (b) ? (a) : (b))
#define min2(a, b) ((a) < (b) ? (a) : (b))

/* samplePopParameters, PopulationTree.c:339-403 */
static void sample_pop_parameters(go_state *s)
{
  go_model *m = &s->m;
  int queue[GO_MAXK], head = 0, tail = 0, pop, b;
  double mean;
  queue[tail++] = m->rootPop;
  while (head < tail) {
    pop = queue[head++];
    mean = m->thetaStart[pop];
    m->theta[pop] = mean * (0.9 + 0.2 * GRND(s));
    if (m->popSon0[pop] >= 0) {
      mean = m->ageStart[pop];
      m->popAge[pop] = mean * (0.9 + 0.2 * GRND(s));
      if (m->popFather[pop] >= 0 && m->popAge[m->popFather[pop]] < m->popAge[pop]) {
        m->popAge[pop] = max2(m->sampleAge[m->popSon0[pop]], m->sampleAge[m->popSon1[pop]]);
        m->popAge[pop] += (m->popAge[m->popFather[pop]] - m->popAge[pop]) * (0.93 + 0.004 * GRND(s));
      }
      queue[tail++] = m->popSon0[pop];
      queue[tail++] = m->popSon1[pop];
    }
  }
  for (b = 0; b < m->B; b++) m->migRate[b] = 0.0;
  go_compute_band_times(m);
}

/* sampleMigRates, PopulationTree.c:414-429 */
static void sample_mig_rates(go_state *s)
{
  go_model *m = &s->m;
  int b;
  double mean;
  for (b = 0; b < m->B; b++) {
    mean = m->mrAlpha[b] / m->mrBeta[b];
    m->migRate[b] = mean * (0.9 + 0.2 * GRND(s));
  }
}

/* Coalescence1Pop, patch.c:279-358 */
static int coalescence_1pop(go_state *s, go_locus *q, int pop, int *living, const int *cum, int *nextId)
{
  go_model *m = &s->m;
  int num, node1, node2, choice;
  double t, T;
  if (pop < m->Kc) {
    node1 = pop > 0 ? cum[pop - 1] : 0;
    num = cum[pop] - node1;
    for (node2 = 0; node2 < num; ++node2) {
      living[node2] = node1 + node2;
      q->nodePop[node1 + node2] = pop;
      q->nodeEvent[node1 + node2] = -1;
      q->left[node1 + node2] = -1;
      q->right[node1 + node2] = -1;
      q->father[node1 + node2] = -1;
      q->age[node1 + node2] = m->sampleAge[pop];
    }
  } else {
    num = coalescence_1pop(s, q, m->popSon0[pop], living, cum, nextId);
    num += coalescence_1pop(s, q, m->popSon1[pop], living + num, cum, nextId);
  }
  T = m->popAge[pop];
  if (pop < m->Kc) T = m->sampleAge[pop];
  for (; num > 1; num--, (*nextId)++) {
    t = -(m->theta[pop] / (num * (num - 1.))) * log(LRND(q));
    T += t;
    if (pop != m->rootPop && T > m->popAge[m->popFather[pop]]) break;
    choice = (int)(num * LRND(q));
    node1 = living[choice];
    living[choice] = living[num - 1];
    choice = (int)((num - 1) * LRND(q));
    node2 = living[choice];
    living[choice] = *nextId;
    q->right[*nextId] = node1;
    q->left[*nextId] = node2;
    q->father[*nextId] = -1;
    q->age[*nextId] = T;
    q->father[node1] = *nextId;
    q->father[node2] = *nextId;
    q->nodePop[*nextId] = pop;
  }
  return num;
}

/* computeTotalStats, patch.c:2134-2165 */
static void compute_total_stats(go_state *s)
{
  go_model *m = &s->m;
  int pop, b, g;
  for (pop = 0; pop < m->K; pop++) { s->tot_coal_stats[pop] = 0; s->tot_num_coals[pop] = 0; }
  for (b = 0; b < m->B; b++) { s->tot_mig_stats[b] = 0; s->tot_num_migs[b] = 0; }
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    for (pop = 0; pop < m->K; pop++) {
      s->tot_coal_stats[pop] += q->coal_stats[pop];
      s->tot_num_coals[pop] += q->num_coals[pop];
    }
    for (b = 0; b < m->B; b++) {
      s->tot_mig_stats[b] += q->mig_stats[b];
      s->tot_num_migs[b] += q->num_migs_band[b];
    }
  }
}

/* initializeMCMC, GPhoCS.c:1122-1225 */
int go_initialize_mcmc(go_state *s)
{
  go_model *m = &s->m;
  int g, pop, totalCoals = 0, cum[GO_MAXK], nextId;
  int *living = (int *)malloc(sizeof(int) * m->n);
  sample_pop_parameters(s);
  /* locus-specific mutation rates, GPhoCS.c:1137-1178: CONST = 1, FIXED = as loaded (readRateFile),
   * VAR = 0.8 + 0.4 u from the locus's own stream, normalised to mean 1 */
  s->rateVar = 0.0;
  if (m->mutRateMode == 1) {
    double total = 0.0, r;
    for (g = 0; g < s->L; g++) {
      r = 0.8 + 0.4 * LRND(&s->loc[g]);
      total += r;
      s->loc[g].mutRate = r;
    }
    total /= s->L;
    for (g = 0; g < s->L; g++) {
      r = s->loc[g].mutRate / total;
      s->loc[g].mutRate = r;
      s->rateVar += (r - 1) * (r - 1);
    }
    s->rateVar /= s->L;
  } else if (m->mutRateMode == 2) {
    /* readRateFile, GPhoCS.c:565-572 */
    for (g = 0; g < s->L; g++) s->rateVar += (s->loc[g].mutRate - 1.0) * (s->loc[g].mutRate - 1.0);
    s->rateVar /= s->L;
  }
  s->logLikelihood = 0.0;
  s->dataLogLikelihood = 0.0;
  cum[0] = m->samplesPerPop[0];
  for (pop = 1; pop < m->Kc; pop++) cum[pop] = cum[pop - 1] + m->samplesPerPop[pop];
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    totalCoals += m->n - 1;
    nextId = m->n;
    coalescence_1pop(s, q, m->rootPop, living, cum, &nextId);
    q->root = nextId - 1;
    go_construct_event_chain(s, q);
    go_compute_genetree_stats(s, q);
    q->genLnL = go_gtree_lnl(s, q);
    s->logLikelihood += q->genLnL;
    s->dataLogLikelihood += go_lik_compute(s, q, 0);
    go_lik_reset_saved(s, q);
  }
  free(living);
  compute_total_stats(s);
  s->logLikelihood = (s->logLikelihood + s->dataLogLikelihood) / s->L;
  return totalCoals;
}

/* UpdateGB_InternalNode, GPhoCS.c:2287-2429 */
int go_update_internal_nodes(go_state *s, double finetune)
{
  go_model *m = &s->m;
  int accepted = 0, g;
  if (finetune <= 0.0) return 0;
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    int pop, inode, i, son, mig, acc = 0;
    double t, tnew, lnacc, lnLd, dgen, tb[2], dData = 0, dLog = 0;
    for (inode = m->n; inode < 2 * m->n - 1; inode++) {
      t = q->age[inode];
      pop = q->nodePop[inode];
      tb[0] = m->popAge[pop];
      if (pop != m->rootPop) tb[1] = m->popAge[m->popFather[pop]];
      else tb[1] = GO_OLDAGE;
      mig = go_find_first_mig(q, inode, -1);
      if (mig >= 0) tb[1] = min2(tb[1], q->mig[mig].age);
      else if (inode != q->root) tb[1] = min2(tb[1], q->age[q->father[inode]]);
      for (i = 0; i < 2; i++) {
        son = i == 0 ? q->left[inode] : q->right[inode];
        mig = go_find_last_mig(q, son, -1);
        if (mig >= 0) tb[0] = max2(tb[0], q->mig[mig].age);
        else tb[0] = max2(tb[0], q->age[son]);
      }
      tnew = t + finetune * go_rnd2normal8(&q->rx, &q->ry, &q->rz);
      tnew = go_reflect(tnew, tb[0], tb[1]);
      if (fabs(tnew - t) < 1e-15) { acc++; continue; }
      go_lik_adjust_age(q, inode, tnew);
      lnLd = -q->dataLnL;
      lnLd += go_lik_compute(s, q, 1);
      dgen = go_consider_event_move(s, q, 0, q->nodeEvent[inode], pop, t, pop, tnew);
      lnacc = dgen + lnLd;
      if (lnacc >= 0 || LRND(q) < exp(lnacc)) {
        acc++;
        q->genLnL += dgen;
        dData += lnLd;
        dLog += (dgen + lnLd) / s->L;
        go_accept_event_chain_changes(s, q, 0);
        go_lik_reset_saved(s, q);
      } else {
        go_reject_event_chain_changes(s, q, 0);
        go_lik_revert(s, q);
      }
    }
    s->dataLogLikelihood += dData;
    s->logLikelihood += dLog;
    accepted += acc;
  }
  return accepted;
}

/* UpdateGB_MigrationNode, GPhoCS.c:2439-2590 */
int go_update_migration_nodes(go_state *s, double finetune)
{
  go_model *m = &s->m;
  int g, accepted = 0;
  if (finetune <= 0.0) return 0;
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    int mi, mignode, pop_s, pop_t, ev_s, ev_t, below, mig_below, mig_above, father, acc = 0;
    double t, tnew, tb[2], dgen, lnacc, dLog = 0;
    for (mi = 0; mi < q->num_migs; mi++) {
      mignode = q->living[mi];
      t = q->mig[mignode].age;
      pop_s = q->mig[mignode].source_pop;
      pop_t = q->mig[mignode].target_pop;
      ev_s = q->mig[mignode].source_event;
      ev_t = q->mig[mignode].target_event;
      below = q->mig[mignode].branch;
      tb[0] = m->bandStart[q->mig[mignode].band];
      tb[1] = m->bandEnd[q->mig[mignode].band];
      mig_below = go_find_last_mig(q, below, t);
      mig_above = go_find_first_mig(q, below, t);
      if (mig_below >= 0) tb[0] = max2(tb[0], q->mig[mig_below].age);
      else tb[0] = max2(tb[0], q->age[below]);
      if (mig_above >= 0) tb[1] = min2(tb[1], q->mig[mig_above].age);
      else {
        father = q->father[below];
        if (father < 0) tb[1] = min2(tb[1], GO_OLDAGE);
        else tb[1] = min2(tb[1], q->age[father]);
      }
      tnew = t + finetune * go_rnd2normal8(&q->rx, &q->ry, &q->rz);
      tnew = go_reflect(tnew, tb[0], tb[1]);
      if (fabs(tnew - t) < 1e-15) { acc++; continue; }
      dgen = go_consider_event_move(s, q, 0, ev_s, pop_s, t, pop_s, tnew);
      dgen += go_consider_event_move(s, q, 1, ev_t, pop_t, t, pop_t, tnew);
      lnacc = dgen;
      if (lnacc >= 0 || LRND(q) < exp(lnacc)) {
        acc++;
        q->genLnL += dgen;
        dLog += dgen / s->L;
        go_accept_event_chain_changes(s, q, 0);
        go_accept_event_chain_changes(s, q, 1);
        q->mig[mignode].age = tnew;
      } else {
        go_reject_event_chain_changes(s, q, 0);
        go_reject_event_chain_changes(s, q, 1);
      }
    }
    s->logLikelihood += dLog;
    accepted += acc;
  }
  return accepted;
}

/* UpdateGB_MigSPR, GPhoCS.c:2598-2948 (no admixture) */
int go_update_mig_spr(go_state *s)
{
  go_model *m = &s->m;
  int accepted = 0, g;
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    int node, res, father, father_pop_old, sibling, b, i, mig, ev, target, pop, acc = 0;
    double lnLd, lnacc, t_new;
    for (node = 0; node < 2 * m->n - 1; node++) {
      if (node == q->root) continue;
      father = q->father[node];
      father_pop_old = q->nodePop[father];
      sibling = q->left[father] + q->right[father] - node;
      go_trace_lineage(s, q, node, 0);
      res = go_trace_lineage(s, q, node, 1);
      lnLd = -q->dataLnL;
      lnLd += go_lik_compute(s, q, 1);
      lnacc = lnLd;
      if (res >= 0 && (lnacc >= 0 || LRND(q) < exp(lnacc))) {
        acc++;
        q->genLnL += (q->spr_delta_lnLd[1] - q->spr_delta_lnLd[0]);
        s->dataLogLikelihood += lnLd;
        s->logLikelihood += (lnLd - q->spr_delta_lnLd[0] + q->spr_delta_lnLd[1]) / s->L;
        target = q->spr_target;
        t_new = q->age[father];
        for (i = 0; i < q->num_migs; i++) {
          mig = q->living[i];
          if (q->mig[mig].branch == father) q->mig[mig].branch = sibling;
          if (target == father) target = sibling;
          if (q->mig[mig].branch == target && q->mig[mig].age >= t_new) q->mig[mig].branch = father;
        }
        go_remove_event(q, q->spr_father_event_old);
        q->ev_type[q->spr_father_event_new] = GO_COAL;
        q->ev_node[q->spr_father_event_new] = father;
        q->nodeEvent[father] = q->spr_father_event_new;
        if (q->spr_father_pop_new != father_pop_old) {
          q->nodePop[father] = q->spr_father_pop_new;
          q->num_coals[father_pop_old]--;
          s->tot_num_coals[father_pop_old]--;
          q->num_coals[q->spr_father_pop_new]++;
          s->tot_num_coals[q->spr_father_pop_new]++;
        }
        go_replace_mig_nodes(s, q, node);
        for (i = 0; i < q->delta[1].num_changed_events; ++i) {
          ev = q->delta[1].changed_events[i];
          q->ev_nlin[ev]++;
        }
        for (b = 0; b < m->B; ++b) {
          q->mig_stats[b] += (q->delta[1].mig_delta[b] - q->delta[0].mig_delta[b]);
          s->tot_mig_stats[b] += (q->delta[1].mig_delta[b] - q->delta[0].mig_delta[b]);
        }
        for (pop = 0; pop < m->K; pop++) {
          q->coal_stats[pop] += q->delta[1].coal_delta[pop] - q->delta[0].coal_delta[pop];
          s->tot_coal_stats[pop] += (q->delta[1].coal_delta[pop] - q->delta[0].coal_delta[pop]);
        }
        go_lik_reset_saved(s, q);
      } else {
        if (res >= 0) go_remove_event(q, q->spr_father_event_new);
        for (i = 0; i < q->spr_num_new_migs; i++) {
          go_remove_event(q, q->spr_new_in[i]);
          go_remove_event(q, q->spr_new_out[i]);
        }
        for (i = 0; i < q->delta[0].num_changed_events; ++i) {
          ev = q->delta[0].changed_events[i];
          q->ev_nlin[ev]++;
        }
        go_lik_revert(s, q);
      }
    }
    accepted += acc;
  }
  return accepted;
}

/* UpdateTheta, GPhoCS.c:3037-3107 */
int go_update_theta(go_state *s, double finetune)
{
  go_model *m = &s->m;
  int pop, g, accepted = 0;
  double thetaold, thetanew, c, lnc, lnacc, dLL;
  if (finetune <= 0.0) return 0;
  for (pop = 0; pop < m->K; pop++) {
    thetaold = m->theta[pop];
    lnc = finetune * go_rnd2normal8(&s->gx, &s->gy, &s->gz);
    c = exp(lnc);
    thetanew = thetaold * c;
    lnacc = lnc + lnc * (m->thetaAlpha[pop] - 1) - (thetanew - thetaold) * m->thetaBeta[pop];
    dLL = -(lnc * s->tot_num_coals[pop] + (1 / thetanew - 1 / thetaold) * s->tot_coal_stats[pop]);
    lnacc += dLL;
    if (lnacc >= 0 || GRND(s) < exp(lnacc)) {
      accepted++;
      for (g = 0; g < s->L; g++) {
        go_locus *q = &s->loc[g];
        q->genLnL -= (lnc * q->num_coals[pop] + (1 / thetanew - 1 / thetaold) * q->coal_stats[pop]);
      }
      s->logLikelihood += dLL / s->L;
      m->theta[pop] = thetanew;
    }
  }
  return accepted;
}

/* UpdateMigRates, GPhoCS.c:3115-3213 */
int go_update_mig_rates(go_state *s, double finetune)
{
  go_model *m = &s->m;
  int b, g, accepted = 0;
  double old_rate, new_rate, c, lnc, lnacc, dLL;
  if (finetune <= 0.0) return 0;
  for (b = 0; b < m->B; b++) {
    old_rate = m->migRate[b];
    lnc = finetune * go_rnd2normal8(&s->gx, &s->gy, &s->gz);
    c = exp(lnc);
    new_rate = old_rate * c;
    if (new_rate < 0.00001) continue;
    lnacc = lnc + lnc * (m->mrAlpha[b] - 1) - (new_rate - old_rate) * m->mrBeta[b];
    dLL = (lnc * s->tot_num_migs[b] - (new_rate - old_rate) * s->tot_mig_stats[b]);
    lnacc += dLL;
    if (lnacc >= 0 || GRND(s) < exp(lnacc)) {
      accepted++;
      for (g = 0; g < s->L; g++) {
        go_locus *q = &s->loc[g];
        q->genLnL += (lnc * q->num_migs_band[b] - (new_rate - old_rate) * q->mig_stats[b]);
      }
      m->migRate[b] = new_rate;
      s->logLikelihood += dLL / s->L;
    }
  }
  return accepted;
}

/* adjustRootEvents, patch.c:1808-1825 */
static void adjust_root_events(go_state *s)
{
  go_model *m = &s->m;
  int g, ev;
  double age;
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    ev = q->first_event[m->rootPop];
    age = m->popAge[m->rootPop];
    while (q->ev_next[ev] >= 0) { age += q->ev_time[ev]; ev = q->ev_next[ev]; }
    q->ev_time[ev] = GO_OLDAGE - age;
  }
}

/* UpdateTau, GPhoCS.c:3224-3994 */
void go_update_tau(go_state *s, const double *finetunes, int *accepted)
{
  go_model *m = &s->m;
  int k, ap, g, ntj[2], num_aff, aff_bands[GO_MAXB], start_or_end[GO_MAXB], b, src, tgt, sons[2];
  int isRoot, res, mig_conflict;
  double tauold, taunew, taub[2], taufactor[2], lnacc, new_band_ages[GO_MAXB], dData, dGen;
  memset(start_or_end, 0, sizeof start_or_end);
  memset(aff_bands, 0, sizeof aff_bands);
  for (ap = m->Kc; ap < m->K; ++ap) {
    accepted[ap] = 0;
    isRoot = (ap == m->rootPop);
    tauold = m->popAge[ap];
    sons[0] = m->popSon0[ap];
    sons[1] = m->popSon1[ap];
    taub[0] = max2(m->popAge[sons[0]], m->popAge[sons[1]]);
    taub[0] = max2(taub[0], m->sampleAge[sons[0]]);
    taub[0] = max2(taub[0], m->sampleAge[sons[1]]);
    if (isRoot) taub[1] = GO_OLDAGE;
    else taub[1] = m->popAge[m->popFather[ap]];
    for (b = 0; b < m->B; b++) {
      src = m->bandSrc[b];
      tgt = m->bandTgt[b];
      if (src == ap || tgt == ap) taub[1] = min2(taub[1], m->bandEnd[b]);
      else if (src == sons[0] || src == sons[1] || tgt == sons[0] || tgt == sons[1])
        taub[0] = max2(taub[0], m->bandStart[b]);
    }
    taunew = tauold + finetunes[ap] * go_rnd2normal8(&s->gx, &s->gy, &s->gz);
    taunew = go_reflect(taunew, taub[0], taub[1]);
    m->popAge[ap] = taunew; /* temporarily, restored below (GPhoCS.c:3302, 3444) */
    for (k = 0; k < 2; k++) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
    if (isRoot) taufactor[1] = taufactor[0];
    num_aff = 0;
    for (b = 0; b < m->B; b++) {
      src = m->bandSrc[b];
      tgt = m->bandTgt[b];
      res = go_update_band_times(m, b);
      if ((src == sons[0] && tgt == sons[1]) || (src == sons[1] && tgt == sons[0])) {
        /* bands between the two sons: not affected */
      } else if (tgt == ap) {
        if (m->bandEnd[b] < taub[1]) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 0;
          new_band_ages[num_aff] = taub[1] + (m->bandEnd[b] - taub[1]) / taufactor[1];
          num_aff++;
        }
        if (m->bandStart[b] < taub[1] && m->popAge[src] > min2(tauold, taunew)) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 1;
          new_band_ages[num_aff] = taub[1] + (m->bandStart[b] - taub[1]) / taufactor[1];
          if (new_band_ages[num_aff] < tauold) new_band_ages[num_aff] = tauold;
          num_aff++;
        }
      } else if (tgt == sons[0] || tgt == sons[1]) {
        if (m->bandStart[b] > taub[0]) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 1;
          new_band_ages[num_aff] = taub[0] + (m->bandStart[b] - taub[0]) / taufactor[0];
          num_aff++;
        }
        if (m->bandEnd[b] > taub[0] && m->popAge[m->popFather[src]] < max2(tauold, taunew)) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 0;
          new_band_ages[num_aff] = taub[0] + (m->bandEnd[b] - taub[0]) / taufactor[0];
          num_aff++;
        }
      } else if (res && src == ap) {
        aff_bands[num_aff] = b;
        start_or_end[num_aff] = 1;
        new_band_ages[num_aff] = m->bandStart[b];
        num_aff++;
      } else if (res && (src == sons[0] || src == sons[1])) {
        aff_bands[num_aff] = b;
        start_or_end[num_aff] = 0;
        new_band_ages[num_aff] = m->bandEnd[b];
        num_aff++;
      }
    }
    m->popAge[ap] = tauold;
    lnacc = log(taunew / tauold) * (m->ageAlpha[ap] - 1) - (taunew - tauold) * m->ageBeta[ap];
    dData = 0.0;
    dGen = 0.0;
    mig_conflict = 0;
    ntj[0] = ntj[1] = 0;

    /* loop 1: evaluate (GPhoCS.c:3491-3833) */
    for (g = 0; g < s->L; g++) {
      go_locus *q = &s->loc[g];
      double age_mt, new_age = 0.0, dGen_l = 0, dData_l = 0;
      int srcP, tgtP, fatherNode, inode, inORout = -1, ev = -1, n1[2] = {0, 0}, i, mig, mig1, band, pop;
      q->mig_conflict_log = 0;
      if (mig_conflict == 0) {
        q->mig_conflict_log = 1;
        q->rb_num_moved = 0;
        ev = -1;
        new_age = 0.0;
        for (i = 0; i < q->num_migs; i++) {
          if (mig_conflict == 0) {
            pop = -1;
            mig = q->living[i];
            band = q->mig[mig].band;
            srcP = q->mig[mig].source_pop;
            tgtP = q->mig[mig].target_pop;
            age_mt = q->mig[mig].age;
            if (age_mt < taub[0] || age_mt > taub[1]) continue;
            if ((srcP == sons[0] && tgtP == sons[1]) || (srcP == sons[1] && tgtP == sons[0])) {
              n1[0]++;
            } else if (srcP == ap) {
              inORout = 1;
              ev = q->mig[mig].target_event;
              pop = tgtP;
              new_age = taub[1] + taufactor[1] * (age_mt - taub[1]);
              n1[1]++;
            } else if (tgtP == ap) {
              inORout = 0;
              ev = q->mig[mig].source_event;
              pop = srcP;
              new_age = taub[1] + taufactor[1] * (age_mt - taub[1]);
              n1[1]++;
            } else if ((srcP == sons[0] || srcP == sons[1]) && q->mig[mig].age > taub[0]) {
              inORout = 1;
              ev = q->mig[mig].target_event;
              pop = tgtP;
              new_age = taub[0] + taufactor[0] * (age_mt - taub[0]);
              n1[0]++;
            } else if ((tgtP == sons[0] || tgtP == sons[1]) && q->mig[mig].age > taub[0]) {
              inORout = 0;
              ev = q->mig[mig].source_event;
              pop = srcP;
              new_age = taub[0] + taufactor[0] * (age_mt - taub[0]);
              n1[0]++;
            }
            if (ev >= 0) {
              inode = q->mig[mig].branch;
              if (new_age >= m->bandEnd[band]) mig_conflict = 1;
              else if (new_age <= m->bandStart[band]) mig_conflict = 1;
              else if (inORout == 0 && new_age > age_mt) {
                fatherNode = q->father[inode];
                mig1 = go_find_first_mig(q, inode, q->mig[mig].age);
                if (mig1 >= 0 && q->mig[mig1].source_pop != ap && q->mig[mig1].source_pop != sons[0] &&
                    q->mig[mig1].source_pop != sons[1] && new_age >= q->mig[mig1].age)
                  mig_conflict = 1;
                else if (fatherNode >= 0 && new_age >= q->age[fatherNode])
                  mig_conflict = 1;
              } else if (inORout == 1 && new_age < age_mt) {
                mig1 = go_find_last_mig(q, inode, q->mig[mig].age);
                if (mig1 >= 0 && q->mig[mig1].target_pop != ap && q->mig[mig1].target_pop != sons[0] &&
                    q->mig[mig1].target_pop != sons[1] && new_age <= q->mig[mig1].age)
                  mig_conflict = 1;
                else if (new_age <= q->age[inode])
                  mig_conflict = 1;
              }
              if (mig_conflict != 1) {
                q->rb_orig[q->rb_num_moved] = ev;
                q->rb_pops[q->rb_num_moved] = pop;
                q->rb_new_ages[q->rb_num_moved] = new_age;
                q->rb_num_moved++;
                ev = -1;
              }
            }
          }
        }
        if (mig_conflict) {
          q->rb_num_moved = 0;
        } else {
          for (i = 0; i < num_aff; i++) {
            band = aff_bands[i];
            tgtP = m->bandTgt[band];
            for (ev = q->first_event[tgtP]; ev >= 0; ev = q->ev_next[ev]) {
              if (q->ev_node[ev] == band &&
                  ((q->ev_type[ev] == GO_MIG_BAND_START && start_or_end[i]) ||
                   q->ev_type[ev] == GO_MIG_BAND_END))
                break;
            }
            if (ev < 0) go_fatal(s, 74, "UpdateTau: band event not found");
            q->rb_orig[q->rb_num_moved] = ev;
            q->rb_pops[q->rb_num_moved] = tgtP;
            q->rb_new_ages[q->rb_num_moved] = new_band_ages[i];
            q->rb_num_moved++;
          }
          q->genDelta = go_rubber_band_ripple(s, q, 1);
          if (isRoot) q->genDelta += go_rubber_band(s, q, ap, taub[0], tauold, taufactor[1], 0, &n1[1]);
          else q->genDelta += go_rubber_band(s, q, ap, taub[1], tauold, taufactor[1], 0, &n1[1]);
          q->genDelta += go_rubber_band(s, q, sons[0], taub[0], tauold, taufactor[0], 0, &n1[0]);
          q->genDelta += go_rubber_band(s, q, sons[1], taub[0], tauold, taufactor[0], 0, &n1[0]);
          dGen_l += q->genDelta;
          ntj[0] += n1[0];
          ntj[1] += n1[1];
          if (n1[0] + n1[1]) {
            dData_l -= q->dataLnL;
            dData_l += go_lik_compute(s, q, 1);
          }
          dGen += dGen_l;
          dData += dData_l;
        }
      }
    }
    lnacc += dData + dGen + ntj[0] * log(taufactor[0]) + ntj[1] * log(taufactor[1]);

    if (!mig_conflict && (lnacc >= 0 || GRND(s) < exp(lnacc))) {
      accepted[ap]++;
      s->dataLogLikelihood += dData;
      s->logLikelihood += (dData + dGen) / s->L;
      /* loop 2: commit (GPhoCS.c:3882-3936) */
      for (g = 0; g < s->L; g++) {
        go_locus *q = &s->loc[g];
        int dummy = 0, i, mig;
        q->genLnL += q->genDelta;
        if (isRoot) go_rubber_band(s, q, ap, taub[0], tauold, taufactor[1], 1, &dummy);
        else go_rubber_band(s, q, ap, taub[1], tauold, taufactor[1], 1, &dummy);
        go_rubber_band(s, q, sons[0], taub[0], tauold, taufactor[0], 1, &dummy);
        go_rubber_band(s, q, sons[1], taub[0], tauold, taufactor[0], 1, &dummy);
        go_lik_reset_saved(s, q);
        for (i = 0; i < q->rb_num_moved; i++) {
          mig = q->ev_node[q->rb_new[i]];
          if (q->ev_type[q->rb_new[i]] == GO_IN_MIG) {
            q->mig[mig].target_event = q->rb_new[i];
            q->mig[mig].age = q->rb_new_ages[i];
          } else if (q->ev_type[q->rb_new[i]] == GO_OUT_MIG) {
            q->mig[mig].source_event = q->rb_new[i];
          }
          go_remove_event(q, q->rb_orig[i]);
        }
        q->rb_num_moved = 0;
      }
      m->popAge[ap] = taunew;
      if (isRoot) adjust_root_events(s);
    } else {
      go_compute_band_times(m);
      if (mig_conflict) {
        s->rubberband_mig_conflicts++;
        for (g = 0; g < s->L; g++) {
          go_locus *q = &s->loc[g];
          if (q->mig_conflict_log == 1) {
            go_lik_revert(s, q);
            go_rubber_band_ripple(s, q, 0);
          }
        }
      } else {
        for (g = s->L - 1; g >= 0; --g) {
          go_locus *q = &s->loc[g];
          go_lik_revert(s, q);
          go_rubber_band_ripple(s, q, 0);
        }
      }
    }
  }
}

/* UpdateSampleAge, GPhoCS.c:4006-4584: the sample age of a current population with an
 * estimated ("e") age is moved inside [0, father age]; the population's single chain is
 * rubber-banded around the old sample age (below it with factor[0], above with factor[1]) */
void go_update_sample_age(go_state *s, const double *finetunes, int *accepted)
{
  go_model *m = &s->m;
  int k, pop, g, ntj[2], num_aff, aff_bands[2 * GO_MAXB], start_or_end[2 * GO_MAXB], b, tgt, mig_conflict;
  double tauold, taunew, taub[2], taufactor[2], lnacc, new_band_ages[2 * GO_MAXB], dData, dGen, age;
  for (pop = 0; pop < m->Kc; pop++) {
    accepted[pop] = 0;
    if (!m->updateSampleAge[pop]) continue;
    tauold = m->sampleAge[pop];
    taub[0] = 0.0;
    taub[1] = m->popAge[m->popFather[pop]];
    taunew = tauold + finetunes[pop] * go_rnd2normal8(&s->gx, &s->gy, &s->gz);
    taunew = go_reflect(taunew, taub[0], taub[1]);
    for (k = 0; k < 2; ++k) taufactor[k] = (taunew - taub[k]) / (tauold - taub[k]);
    num_aff = 0;
    for (b = 0; b < m->B; ++b) {
      tgt = m->bandTgt[b];
      if (tgt == pop) {
        if (m->bandEnd[b] < taub[1] && m->bandEnd[b] > taub[0]) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 0;
          age = m->bandEnd[b];
          new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
          ++num_aff;
        }
        if (m->bandStart[b] < taub[1] && m->bandStart[b] > taub[0]) {
          aff_bands[num_aff] = b;
          start_or_end[num_aff] = 1;
          age = m->bandStart[b];
          new_band_ages[num_aff] = taub[age > taunew] + (age - taub[age > taunew]) / taufactor[age > taunew];
          if (new_band_ages[num_aff] < tauold) new_band_ages[num_aff] = tauold;
          ++num_aff;
        }
      }
    }
    /* the model keeps the OLD sample age during the evaluation (GPhoCS.c:4116) */
    lnacc = log(taunew / tauold) * (m->ageAlpha[pop] - 1) - (taunew - tauold) * m->ageBeta[pop];
    dData = 0.0;
    dGen = 0.0;
    mig_conflict = 0;
    ntj[0] = ntj[1] = 0;
    for (g = 0; g < s->L; ++g) {
      go_locus *q = &s->loc[g];
      double age_mt, new_age = 0.0, dGen_l = 0, dData_l = 0;
      int srcP, tgtP, fatherNode, inode, inORout = -1, ev = -1, n1[2] = {0, 0}, i, mig, mig1, band, migPop = -1;
      q->mig_conflict_log = 0;
      if (mig_conflict == 0) {
        q->mig_conflict_log = 1;
        q->rb_num_moved = 0;
        for (i = 0; i < q->num_migs; ++i) {
          if (mig_conflict == 0) {
            mig = q->living[i];
            band = q->mig[mig].band;
            srcP = q->mig[mig].source_pop;
            tgtP = q->mig[mig].target_pop;
            age_mt = q->mig[mig].age;
            if (age_mt < taub[0] || age_mt > taub[1]) continue;
            if (srcP == pop) {
              inORout = 1;
              ev = q->mig[mig].target_event;
              migPop = tgtP;
              new_age = taub[age_mt > tauold] + taufactor[age_mt > tauold] * (age_mt - taub[age_mt > tauold]);
              ++n1[age_mt > tauold];
            } else if (tgtP == pop) {
              inORout = 0;
              ev = q->mig[mig].source_event;
              migPop = srcP;
              new_age = taub[age_mt > tauold] + taufactor[age_mt > tauold] * (age_mt - taub[age_mt > tauold]);
              ++n1[age_mt > tauold];
            }
            if (ev >= 0) {
              inode = q->mig[mig].branch;
              if (new_age >= m->bandEnd[band]) mig_conflict = 1;
              else if (new_age <= m->bandStart[band]) mig_conflict = 1;
              else if (inORout == 0 && new_age > age_mt) {
                fatherNode = q->father[inode];
                mig1 = go_find_first_mig(q, inode, q->mig[mig].age);
                if (mig1 >= 0 && pop != q->mig[mig1].source_pop && new_age >= q->mig[mig1].age) mig_conflict = 1;
                else if (fatherNode >= 0 && new_age >= q->age[fatherNode]) mig_conflict = 1;
              } else if (inORout == 1 && new_age < age_mt) {
                mig1 = go_find_last_mig(q, inode, q->mig[mig].age);
                if (mig1 >= 0 && pop != q->mig[mig1].target_pop && new_age <= q->mig[mig1].age) mig_conflict = 1;
                else if (new_age <= q->age[inode]) mig_conflict = 1;
              }
              if (mig_conflict == 0) {
                q->rb_orig[q->rb_num_moved] = ev;
                q->rb_pops[q->rb_num_moved] = migPop;
                q->rb_new_ages[q->rb_num_moved] = new_age;
                q->rb_num_moved++;
                ev = -1;
              }
            }
          }
        }
        if (mig_conflict) {
          q->rb_num_moved = 0;
        } else {
          for (i = 0; i < num_aff; ++i) {
            band = aff_bands[i];
            tgtP = m->bandTgt[band];
            for (ev = q->first_event[tgtP]; ev >= 0; ev = q->ev_next[ev]) {
              if (q->ev_node[ev] == band &&
                  ((start_or_end[i] && q->ev_type[ev] == GO_MIG_BAND_START) || q->ev_type[ev] == GO_MIG_BAND_END))
                break;
            }
            if (ev < 0) go_fatal(s, 174, "UpdateSampleAge: band event not found");
            q->rb_orig[q->rb_num_moved] = ev;
            q->rb_pops[q->rb_num_moved] = tgtP;
            q->rb_new_ages[q->rb_num_moved] = new_band_ages[i];
            q->rb_num_moved++;
          }
          q->genDelta = go_rubber_band_ripple(s, q, 1);
          q->genDelta += go_rubber_band(s, q, pop, taub[1], tauold, taufactor[1], 0, &n1[1]);
          q->genDelta += go_rubber_band(s, q, pop, taub[0], tauold, taufactor[0], 0, &n1[0]);
          dGen_l += q->genDelta;
          ntj[0] += n1[0];
          ntj[1] += n1[1];
          dData_l -= q->dataLnL;
          dData_l += go_lik_compute(s, q, 1);
          dData += dData_l;
          dGen += dGen_l;
        }
      }
    }
    lnacc += dData + dGen + ntj[0] * log(taufactor[0]) + ntj[1] * log(taufactor[1]);
    if (!mig_conflict && (lnacc >= 0 || GRND(s) < exp(lnacc))) {
      ++accepted[pop];
      s->dataLogLikelihood += dData;
      s->logLikelihood += (dData + dGen) / s->L;
      for (g = 0; g < s->L; g++) {
        go_locus *q = &s->loc[g];
        int dummy = 0, i, mig, nw;
        q->genLnL += q->genDelta;
        go_rubber_band(s, q, pop, taub[1], tauold, taufactor[1], 1, &dummy);
        go_rubber_band(s, q, pop, taub[0], tauold, taufactor[0], 1, &dummy);
        go_lik_reset_saved(s, q);
        for (i = 0; i < q->rb_num_moved; ++i) {
          nw = q->rb_new[i];
          mig = q->ev_node[nw];
          if (q->ev_type[nw] == GO_IN_MIG) {
            q->mig[mig].target_event = nw;
            q->mig[mig].age = q->rb_new_ages[i];
          } else if (q->ev_type[nw] == GO_OUT_MIG) {
            q->mig[mig].source_event = nw;
          }
          go_remove_event(q, q->rb_orig[i]);
        }
        q->rb_num_moved = 0;
      }
      m->sampleAge[pop] = taunew;
    } else {
      if (mig_conflict) {
        s->rubberband_mig_conflicts++;
        for (g = 0; g < s->L; ++g) {
          go_locus *q = &s->loc[g];
          if (q->mig_conflict_log == 1) {
            go_lik_revert(s, q);
            go_rubber_band_ripple(s, q, 0);
          }
        }
      } else {
        for (g = s->L - 1; g >= 0; --g) {
          go_locus *q = &s->loc[g];
          go_lik_revert(s, q);
          go_rubber_band_ripple(s, q, 0);
        }
      }
    }
  }
}

/* mixing, GPhoCS.c:4688-4912 */
/* UpdateLocusRate, GPhoCS.c:4598-4680: for every locus but the reference one (genRateRef = 0,
 * :1177) shift its rate and the reference locus's rate in opposite directions (mean rate stays 1),
 * full recomputation of both data likelihoods, Dirichlet(alpha) prior ratio.  Serial over loci by
 * construction: the reference locus's rate carries every earlier decision. */
int go_update_locus_rate(go_state *s, double finetune)
{
  go_model *m = &s->m;
  int accepted = 0, g;
  const int ref = 0;
  go_locus *qr = &s->loc[ref];
  double lnacc, lnLd, rold, rnew, rrefold, rrefnew;
  if (finetune <= 0.0) return 0;
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    if (g == ref) continue;
    rrefold = qr->mutRate;
    rold = q->mutRate;
    rnew = rold + finetune * go_rnd2normal8(&q->rx, &q->ry, &q->rz);
    rnew = go_reflect(rnew, 0, rold + rrefold);
    q->mutRate = rnew;
    rrefnew = rrefold + rold - rnew;
    qr->mutRate = rrefnew;
    lnacc = (m->varRatesAlpha - 1) * log((rnew * rrefnew) / (rold * rrefold));
    lnLd = -(q->dataLnL + qr->dataLnL);
    lnLd += go_lik_compute(s, q, 0);
    lnLd += go_lik_compute(s, qr, 0);
    lnacc += lnLd;
    if (lnacc >= 0 || LRND(q) < exp(lnacc)) {
      accepted++;
      s->dataLogLikelihood += lnLd;
      s->logLikelihood += lnLd / s->L;
      go_lik_reset_saved(s, q);
      go_lik_reset_saved(s, qr);
      s->rateVar += (rnew * rnew + rrefnew * rrefnew - rold * rold - rrefold * rrefold) / s->L;
    } else {
      q->mutRate = rold;
      qr->mutRate = rrefold;
      go_lik_revert(s, q);
      go_lik_revert(s, qr);
    }
  }
  return accepted;
}

int go_mixing(go_state *s, double finetune)
{
  go_model *m = &s->m;
  double xold, xnew, c, lnc, lnacc, dData, dGen;
  int g, b, pop, num_events;
  if (finetune <= 0.0) return 0;
  lnc = finetune * go_rnd2normal8(&s->gx, &s->gy, &s->gz);
  c = exp(lnc);
  num_events = 0;
  for (pop = 0; pop < m->K; pop++) num_events += s->tot_num_coals[pop];
  for (b = 0; b < m->B; b++) num_events += s->tot_num_migs[b];
  lnacc = lnc * (2 * m->K - m->Kc - m->B + num_events);
  dData = 0.0;
  dGen = 0.0;
  for (pop = 0; pop < m->K; pop++) {
    xold = m->theta[pop];
    m->theta[pop] = xnew = xold * c;
    lnacc += lnc * (m->thetaAlpha[pop] - 1) - (xnew - xold) * m->thetaBeta[pop];
    dGen -= lnc * s->tot_num_coals[pop];
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
    dGen -= lnc * s->tot_num_migs[b];
  }
  for (g = 0; g < s->L; g++) {
    double d = go_lik_scale_ages(s, &s->loc[g], c);
    dData += d;
  }
  lnacc += (dData + dGen);
  if (lnacc >= 0 || GRND(s) < exp(lnacc)) {
    for (g = 0; g < s->L; g++) {
      go_locus *q = &s->loc[g];
      int i;
      go_lik_reset_saved(s, q);
      for (i = 0; i < q->num_migs; i++) q->mig[q->living[i]].age *= c;
      q->genLnL -= lnc * (m->n - 1 + q->num_migs);
      for (pop = 0; pop < m->K; pop++) q->coal_stats[pop] *= c;
      for (b = 0; b < m->B; b++) q->mig_stats[b] *= c;
      for (i = 0; i < q->E; i++)
        if (q->ev_time[i] > 0) q->ev_time[i] *= c;
    }
    for (pop = 0; pop < m->K; pop++) s->tot_coal_stats[pop] *= c;
    for (b = 0; b < m->B; b++) s->tot_mig_stats[b] *= c;
    s->dataLogLikelihood += dData;
    s->logLikelihood += (dData + dGen) / s->L;
    adjust_root_events(s);
    return 1;
  }
  for (g = 0; g < s->L; g++) go_lik_revert(s, &s->loc[g]);
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
  return 0;
}

/* the state-mutating part of checkGtreeStructure, patch.c:2978-3380: per-locus
 * statistics are recomputed from the chain and overwrite the stored ones */
static int check_gtree_structure(go_state *s, go_locus *q)
{
  go_model *m = &s->m;
  int i, n, pop, b, ev, id, queue[GO_MAXK], lins_in[GO_MAXK], live[GO_MAXB], nlive, res = 1;
  double age, dt, PREC = 0.0000000001;
  for (pop = 0; pop < m->K; pop++) lins_in[pop] = 0;
  go_pop_post_order(m, m->rootPop, queue);
  for (i = 0; i < m->K; i++) {
    pop = queue[i];
    q->chk_coal_stats[pop] = 0.0;
    q->chk_num_coals[pop] = 0;
    n = lins_in[pop];
    age = m->popAge[pop];
    nlive = 0;
    for (ev = q->first_event[pop]; ev >= 0; ev = q->ev_next[ev]) {
      if (q->ev_nlin[ev] != n) res = 0;
      if (q->ev_next[ev] >= 0 && ev != q->ev_prev[q->ev_next[ev]]) res = 0;
      id = q->ev_node[ev];
      dt = q->ev_time[ev];
      age += dt;
      q->chk_coal_stats[pop] += n * (n - 1) * dt;
      for (b = 0; b < nlive; b++) q->chk_mig_stats[live[b]] += n * dt;
      switch (q->ev_type[ev]) {
      case GO_SAMPLES_START:
        n += m->samplesPerPop[pop];
        if (fabs(m->sampleAge[pop] - age) > PREC) res = 0;
        break;
      case GO_COAL:
        q->chk_num_coals[pop]++;
        n--;
        if (fabs(q->age[id] - age) > PREC) res = 0;
        if (q->nodePop[id] != pop || q->nodeEvent[id] != ev) res = 0;
        break;
      case GO_IN_MIG:
        q->chk_num_migs[q->mig[id].band]++;
        n--;
        if (fabs(q->mig[id].age - age) > PREC || q->mig[id].target_event != ev) res = 0;
        break;
      case GO_OUT_MIG:
        n++;
        if (fabs(q->mig[id].age - age) > PREC || q->mig[id].source_event != ev) res = 0;
        break;
      case GO_MIG_BAND_START:
        live[nlive++] = id;
        q->chk_num_migs[id] = 0;
        q->chk_mig_stats[id] = 0.0;
        if (fabs(m->bandStart[id] - age) > PREC) res = 0;
        break;
      case GO_MIG_BAND_END:
        for (b = 0; b < nlive; b++) if (live[b] == id) break;
        if (b == nlive) res = 0;
        else live[b] = live[--nlive];
        if (fabs(m->bandEnd[id] - age) > PREC) res = 0;
        break;
      case GO_END_CHAIN:
        if (id != pop || nlive != 0 || q->ev_next[ev] >= 0) res = 0;
        if (pop != m->rootPop) {
          lins_in[m->popFather[pop]] += n;
          if (fabs(m->popAge[m->popFather[pop]] - age) > PREC) res = 0;
        }
        break;
      default: res = 0; break;
      }
    }
  }
  for (pop = 0; pop < m->K; pop++) {
    if (fabs(q->chk_coal_stats[pop] - q->coal_stats[pop]) > PREC) res = 0;
    q->coal_stats[pop] = q->chk_coal_stats[pop];
    if (q->chk_num_coals[pop] != q->num_coals[pop]) res = 0;
  }
  for (b = 0; b < m->B; b++) {
    if (fabs(q->chk_mig_stats[b] - q->mig_stats[b]) > PREC) res = 0;
    q->mig_stats[b] = q->chk_mig_stats[b];
    if (q->chk_num_migs[b] != q->num_migs_band[b]) res = 0;
  }
  return res;
}

/* checkAll, patch.c:2745-2884 (checks + accumulator resynchronisation) */
int go_check_all(go_state *s)
{
  go_model *m = &s->m;
  int g, pop, b, res = 1, nc[GO_MAXK], nm[GO_MAXB];
  double PREC = 0.0000001, lnLd_gen, genLnLd = 0.0, dataLnLd = 0.0, cs[GO_MAXK], ms[GO_MAXB];
  for (pop = 0; pop < m->K; pop++) { nc[pop] = 0; cs[pop] = 0.0; }
  for (b = 0; b < m->B; b++) { nm[b] = 0; ms[b] = 0.0; }
  for (g = 0; g < s->L; g++) {
    go_locus *q = &s->loc[g];
    if (!check_gtree_structure(s, q)) { fprintf(stderr, "oracle: checkGtreeStructure failed, locus %d\n", g); return 0; }
    /* checkLocusDataLikelihood, LocusDataLikelihood.c:717-758 */
    go_lik_compute(s, q, 0);
    if (!(q->dataLnL == q->sv_dataLnL || fabs(1 - q->dataLnL / q->sv_dataLnL) < 0.000000001)) {
      fprintf(stderr, "oracle: data likelihood check failed, locus %d\n", g);
      go_lik_reset_saved(s, q);
      return 0;
    }
    go_lik_reset_saved(s, q);
    lnLd_gen = go_gtree_lnl(s, q);
    if (fabs(q->genLnL - lnLd_gen) > PREC && fabs(1 - q->genLnL / lnLd_gen) > PREC) {
      fprintf(stderr, "oracle: genealogy likelihood check failed, locus %d (%g vs %g)\n", g, q->genLnL, lnLd_gen);
      return 0;
    }
    q->genLnL = lnLd_gen;
    dataLnLd += q->dataLnL;
    genLnLd += lnLd_gen;
    for (pop = 0; pop < m->K; pop++) { nc[pop] += q->num_coals[pop]; cs[pop] += q->coal_stats[pop]; }
    for (b = 0; b < m->B; b++) { nm[b] += q->num_migs_band[b]; ms[b] += q->mig_stats[b]; }
  }
  for (pop = 0; pop < m->K; pop++) {
    if (fabs(cs[pop] - s->tot_coal_stats[pop]) > PREC && fabs(1 - cs[pop] / s->tot_coal_stats[pop]) > PREC) res = 0;
    s->tot_coal_stats[pop] = cs[pop];
    if (nc[pop] != s->tot_num_coals[pop]) res = 0;
  }
  for (b = 0; b < m->B; b++) {
    if (fabs(ms[b] - s->tot_mig_stats[b]) > PREC && fabs(1 - ms[b] / s->tot_mig_stats[b]) > PREC) res = 0;
    s->tot_mig_stats[b] = ms[b];
    if (nm[b] != s->tot_num_migs[b]) res = 0;
  }
  if (fabs(s->dataLogLikelihood - dataLnLd) > PREC && fabs(1 - s->dataLogLikelihood / dataLnLd) > PREC) res = 0;
  s->dataLogLikelihood = dataLnLd;
  dataLnLd = (genLnLd + dataLnLd) / s->L;
  if (fabs(1 - s->logLikelihood / dataLnLd) > PREC) res = 0;
  s->logLikelihood = dataLnLd;
  return res;
}

/* recordParamVals, GPhoCS.c:802-849 */
static void record_param_vals(go_state *s)
{
  go_model *m = &s->m;
  int pop, b, ind = 0;
  if (!s->paramVals) s->paramVals = (double *)malloc(sizeof(double) * (m->numParameters + 4));
  for (pop = 0; pop < m->K; pop++) s->paramVals[ind++] = m->theta[pop];
  for (pop = m->Kc; pop < m->K; pop++) s->paramVals[ind++] = m->popAge[pop];
  for (b = 0; b < m->B; b++) s->paramVals[ind++] = m->migRate[b];
  for (pop = 0; pop < m->Kc; pop++)
    if (m->updateSampleAge[pop] || m->sampleAge[pop] > 0.0) s->paramVals[ind++] = m->sampleAge[pop];
  if (m->mutRateMode == 1) s->paramVals[ind++] = sqrt(s->rateVar);
}

static void rec(go_state *s, FILE *tf, int it, const char *what, int acc)
{
  if (tf) fprintf(tf, "IT %d %s %d %a %a\n", it, what, acc, s->dataLogLikelihood, s->logLikelihood);
}

/* one iteration of performMCMC, GPhoCS.c:1476-1821 (genetreeSamples == 1, no
 * find-finetunes, no admixture) */
int go_iteration(go_state *s, int iteration, FILE *tf)
{
  go_model *m = &s->m;
  int acc, pop, g, accArr[GO_MAXK];
  acc = go_update_internal_nodes(s, m->ftCoalTime);
  rec(s, tf, iteration, "INT", acc);
  acc = go_update_migration_nodes(s, m->ftMigTime);
  rec(s, tf, iteration, "MIGN", acc);
  acc = go_update_mig_spr(s);
  rec(s, tf, iteration, "SPR", acc);
  if (m->mutRateMode == 1) {
    acc = go_update_locus_rate(s, m->ftLocusRate);
    rec(s, tf, iteration, "LRATE", acc);
  }
  acc = go_update_theta(s, m->ftTheta);
  rec(s, tf, iteration, "THETA", acc);
  if (iteration > m->startMig) {
    acc = go_update_mig_rates(s, m->ftMigRate);
    rec(s, tf, iteration, "MIGR", acc);
  }
  go_update_tau(s, m->ftTaus, accArr);
  for (pop = m->Kc; pop < m->K; pop++) {
    char nm[32];
    snprintf(nm, sizeof nm, "TAU%d", pop);
    rec(s, tf, iteration, nm, accArr[pop]);
  }
  if (tf) fprintf(tf, "CONFLICTS %d\n", s->rubberband_mig_conflicts);
  go_update_sample_age(s, m->ftTaus, accArr);
  for (pop = 0; pop < m->Kc; pop++) {
    char nm[32];
    if (!m->updateSampleAge[pop]) continue;
    snprintf(nm, sizeof nm, "SAGE%d", pop);
    rec(s, tf, iteration, nm, accArr[pop]);
    if (tf) fprintf(tf, "CONFLICTS %d\n", s->rubberband_mig_conflicts);
  }
  if (m->doMixing) {
    acc = go_mixing(s, m->ftMixing);
    rec(s, tf, iteration, "MIX", acc);
  }
  for (g = 0; g < s->L; g++)
    if (!go_synchronize_events(s, &s->loc[g])) { fprintf(stderr, "oracle: synchronizeEvents failed locus %d\n", g); return -1; }
  record_param_vals(s);
  if (iteration == m->startMig) {
    sample_mig_rates(s);
    for (g = 0; g < s->L; g++) {
      go_locus *q = &s->loc[g];
      s->logLikelihood -= q->genLnL / s->L;
      q->genLnL = go_gtree_lnl(s, q);
      s->logLikelihood += q->genLnL / s->L;
    }
  }
  if ((iteration + 1) % m->samplesPerLog == 0) {
    if (!go_check_all(s)) { fprintf(stderr, "oracle: checkAll failed at iteration %d\n", iteration); return -2; }
    rec(s, tf, iteration, "CHECK", 1);
  }
  return 0;
}

/* trace line, GPhoCS.c:746-754, 1763-1769 */
void go_trace_line(go_state *s, int iteration, FILE *tf)
{
  int i;
  fprintf(tf, "TRACE %d\t", iteration);
  for (i = 0; i < s->m.numParameters; i++) fprintf(tf, "%8.5f\t", s->paramVals[i] * s->m.printFactors[i]);
  fprintf(tf, "\t%.6f\t%.6f\n", s->logLikelihood, s->dataLogLikelihood);
}
