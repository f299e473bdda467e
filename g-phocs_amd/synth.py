"""Synthetic workloads for the benchmark (no network, no datasets): builds a Pack -- model plus
processed loci, i.e. what the reference holds after readControlFile + processAlignments --
directly, without writing a 1.6 GB sequence file.

Recipe: per locus a genealogy is simulated under the config's population tree
at the prior means (theta = 1e-4, tau = tau-initial, no migration); JC69 mutations are dropped
on its branches for a `seqlen`-bp locus; haplotypes are paired into diploid samples, so a site
where a sample's two haplotypes differ becomes an unphased het pattern with 2^h phases
(LocusDataLikelihood.c:466-479 averages over them); `nmask` of the genotypes are N.
Patterns are stored first-occurrence-style (invariant pattern first) with JC-canonical labels
(first leaf's base = T).  Shapes follow BASELINE.json `configs`."""
import math

import numpy as np

CONFIGS = {  # diploid samples per current pop, migration bands (src, tgt), ancient pop
    1: dict(pops=[1, 1, 1, 1], bands=[(3, 1)]),
    2: dict(pops=[2, 1, 1], bands=[]),
    3: dict(pops=[2, 2, 2], bands=[(0, 1), (2, 1)]),
    4: dict(pops=[2, 2, 2, 1, 1], bands=[(0, 1), (1, 0), (3, 2), (4, 3)]),
    5: dict(pops=[2, 2, 2, 1, 1, 1, 1], bands=[(0, 1), (1, 0), (3, 2), (4, 3)], ancient=6),
    # estimated ("e") sample ages: UpdateSampleAge is live and mixing stays on (tools/gen_synth.py configs 6, 7)
    6: dict(pops=[2, 2, 2, 1], bands=[(0, 1), (3, 2), (2, 3)], ancient=3, ancient_est=True),
    7: dict(pops=[2, 1, 2], bands=[(0, 1), (1, 0), (2, 1)], ancient=1, ancient_est=True),
    # more than 16 diploid samples (an ordinary G-PhoCS data set): 20 diploids = 40 leaves over 5 populations -- library
    # variant h (128-bit node sets); and many populations: 12 current (23 in all), 24 leaves
    10: dict(pops=[4, 4, 4, 4, 4], bands=[(0, 1), (1, 0), (3, 2), (4, 3)]),
    11: dict(pops=[1] * 12, bands=[(0, 1), (1, 0), (3, 2), (4, 3), (6, 5), (8, 7)]),
    # beyond the nibble list / 128-bit node sets (tools/gen_synth.py configs 12, 13): 20 migration bands (variant b); 72 leaves
    # (variant n); and 136 leaves with 132 of them in ONE population: event records with more than 127 lineages
    12: dict(pops=[1] * 6, bands=[(i, i + 1) for i in range(5)] + [(i + 1, i) for i in range(5)] + [(i, i + 2) for i in range(4)] +
             [(i + 2, i) for i in range(4)] + [(0, 3), (3, 0)]),
    13: dict(pops=[6] * 6, bands=[(0, 1), (1, 0), (3, 2), (4, 3)]),
    14: dict(pops=[66, 2], bands=[(0, 1)]),
}


def _taus(kc):
    taus, t = [], 5e-6
    for i in range(kc - 1):
        taus.append(t)
        t *= 2.0 if i < kc - 3 else 5.0 if i == kc - 3 else 1.0
    return taus


def make_var_rates(p, alpha, finetune):
    """switch a synthetic pack to `locus-mut-rate VAR alpha` (UpdateLocusRate live): one more recorded parameter"""
    p.mutRateMode, p.varRatesAlpha, p.ftLocusRate = 1, float(alpha), float(finetune)
    p.numParameters += 1
    p.printFactors = np.concatenate([p.printFactors, [1.0]])
    return p


def make_model(pack, config, seed=12345, samples_per_log=100, start_mig=0, do_mixing=None, mig_beta=0.00001):
    """fills the model / prior / finetune fields of `pack` like sample-control-file.ctl:12-27"""
    cfg = CONFIGS[config]
    kc = len(cfg["pops"])
    K, B = 2 * kc - 1, len(cfg["bands"])
    p = pack
    p.n = 2 * sum(cfg["pops"])
    p.Kc, p.K, p.B, p.rootPop = kc, K, B, K - 1
    p.samplesPerPop = np.array([2 * d for d in cfg["pops"]], np.int32)
    p.popFather = np.full(K, -1, np.int32)
    p.popSon0 = np.full(K, -1, np.int32)
    p.popSon1 = np.full(K, -1, np.int32)
    prev = 0
    for i in range(kc - 1):          # caterpillar ((((A,B),C),D),...)
        a = kc + i
        p.popSon0[a], p.popSon1[a] = prev, i + 1
        p.popFather[prev] = p.popFather[i + 1] = a
        prev = a
    p.sampleAge = np.zeros(K)
    p.updateSampleAge = np.zeros(K, np.int32)
    if cfg.get("ancient") is not None:
        p.sampleAge[cfg["ancient"]] = 0.000002
        p.updateSampleAge[cfg["ancient"]] = int(bool(cfg.get("ancient_est")))
    p.thetaAlpha, p.thetaBeta, p.thetaStart = np.full(K, 1.0), np.full(K, 10000.0), np.full(K, 1e-4)
    taus = _taus(kc)
    p.ageAlpha, p.ageBeta, p.ageStart = np.zeros(K), np.zeros(K), np.zeros(K)
    for i in range(kc - 1):
        p.ageAlpha[kc + i], p.ageBeta[kc + i], p.ageStart[kc + i] = 1.0, 20000.0, taus[i]
    p.bandSrc = np.array([s for s, _ in cfg["bands"]] or [0], np.int32)
    p.bandTgt = np.array([t for _, t in cfg["bands"]] or [0], np.int32)
    p.mrAlpha, p.mrBeta = np.full(max(B, 1), 0.002), np.full(max(B, 1), mig_beta)
    p.seed, p.burnin, p.numSamplesMcmc, p.sampleSkip, p.startMig = seed, 0, 1000, 0, start_mig
    # a FIXED non-zero sample age switches mixing off (MCMCcontrol.c:905-907); an estimated one does not
    p.doMixing = int(cfg.get("ancient") is None or bool(cfg.get("ancient_est"))) if do_mixing is None else int(do_mixing)
    p.samplesPerLog, p.mutRateMode = samples_per_log, 0
    p.ftCoalTime, p.ftMigTime, p.ftTheta, p.ftMigRate, p.ftMixing = 0.01, 0.3, 0.04, 0.02, 0.003
    p.ftTaus = np.full(K, 0.0000008)
    p.ftTaus[K - 1] = 0.00000286
    n_anc = 1 if cfg.get("ancient") is not None else 0
    p.numParameters = K + (kc - 1) + B + n_anc
    p.printFactors = np.concatenate([np.full(K + kc - 1, 10000.0), np.full(B, 0.1), np.full(n_anc, 10000.0)])
    p.popName = [chr(65 + i) for i in range(kc)] + [f"anc{i}" for i in range(kc - 1)]
    return taus


def make_synthetic_pack(Pack, config, L, seqlen=1000, nmask=0.002, mut_scale=1.0, data_seed=None,
                        mcmc_seed=12345, samples_per_log=100, mig_beta=0.00001):
    cfg = CONFIGS[config]
    rng = np.random.default_rng(20261002 + config if data_seed is None else data_seed)
    p = Pack()
    taus = make_model(p, config, seed=mcmc_seed, samples_per_log=samples_per_log, mig_beta=mig_beta)
    p.L = p.numLoci = L
    n, kc = p.n, p.Kc
    theta = 1e-4 * mut_scale
    taus = [t * mut_scale for t in taus]
    anc_age = 0.000002 * mut_scale
    nd = n // 2
    leaf_pop = np.repeat(np.arange(kc), p.samplesPerPop)
    offs = [0]
    leaf_rows, phase_rows, count_rows = [], [], []
    expo = rng.exponential
    for g in range(L):
        # --- genealogy (structured coalescent on the caterpillar, no migration)
        age = [0.0] * n
        under = [1 << i for i in range(n)]   # bitmask of leaves under each node
        parent_len = []                       # (node mask, branch length) collected on the way
        node_age = list(age)
        if cfg.get("ancient") is not None:
            for i in range(n):
                if leaf_pop[i] == cfg["ancient"]:
                    node_age[i] = anc_age
        lins = []

        def coalesce(ls, t0, t1):
            t = t0
            while len(ls) > 1:
                k = len(ls)
                t += expo(theta / (k * (k - 1.0)))
                if t1 is not None and t > t1:
                    break
                i, j = rng.choice(k, 2, replace=False)
                a, b = ls[i], ls[j]
                parent_len.append((under[a], t - node_age[a]))
                parent_len.append((under[b], t - node_age[b]))
                under.append(under[a] | under[b])
                node_age.append(t)
                ls = [x for q, x in enumerate(ls) if q != i and q != j] + [len(under) - 1]
            return ls

        start = 0
        per_pop = []
        for c in range(kc):
            k = int(p.samplesPerPop[c])
            per_pop.append(list(range(start, start + k)))
            start += k
        t0 = anc_age if cfg.get("ancient") == 0 else 0.0
        lins = coalesce(per_pop[0], t0, taus[0])
        for c in range(1, kc):
            t0 = anc_age if cfg.get("ancient") == c else 0.0
            l2 = coalesce(per_pop[c], t0, taus[c - 1])
            lins = coalesce(lins + l2, taus[c - 1], taus[c] if c < kc - 1 else None)
        # --- mutations -> patterns
        pats = {}
        for mask, blen in parent_len:
            pm = 0.75 * (1.0 - math.exp(-4.0 * blen / 3.0)) * seqlen
            k = rng.poisson(pm) if pm > 0 else 0
            if k:
                pats[mask] = pats.get(mask, 0) + int(k)
        nvar = sum(pats.values())
        masked = rng.binomial(seqlen, nmask, size=nd)
        rows, phs, cnts = [], [], []
        inv = max(seqlen - nvar - int(masked.sum()), 1)
        rows.append(np.zeros(n, np.uint8)); phs.append(1); cnts.append(inv)
        for d in range(nd):
            if masked[d]:
                r = np.zeros(n, np.uint8)
                r[2 * d] = r[2 * d + 1] = 4
                rows.append(r); phs.append(1); cnts.append(int(masked[d]))
        for mask, cnt in pats.items():
            bits = np.array([(mask >> i) & 1 for i in range(n)], np.uint8)
            if bits[0]:
                bits = 1 - bits                 # JC-canonical: first leaf carries T
            hets = [d for d in range(nd) if bits[2 * d] != bits[2 * d + 1]][:3]
            nph = 1 << len(hets)
            for ph in range(nph):
                r = bits.copy()
                for q, d in enumerate(hets):
                    if (ph >> q) & 1:
                        r[2 * d], r[2 * d + 1] = r[2 * d + 1], r[2 * d]
                rows.append(r)
                phs.append(nph if ph == 0 else 0)
                cnts.append(cnt if ph == 0 else 0)
        leaf_rows.append(np.stack(rows))
        phase_rows.extend(phs)
        count_rows.extend(cnts)
        offs.append(offs[-1] + len(rows))
    p.pattern_offsets = np.array(offs, np.int64)
    p.leafcodes = np.concatenate(leaf_rows).astype(np.uint8)
    p.numPhases = np.array(phase_rows, np.uint16)
    p.counts = np.array(count_rows, np.int32)
    p.mutRates = np.ones(L)
    return p


def replicate_pack(Pack, base, L):
    """tile a pack's loci up to L loci (cheap way to reach 100k+ loci from a few thousand
    genuinely distinct ones; every locus still evolves independently only through its data --
    all loci share one RNG seed in the reference too, utils.c:421-426)"""
    reps = (L + base.L - 1) // base.L
    p = Pack()
    p.__dict__.update(base.__dict__)
    P = np.diff(base.pattern_offsets)
    P_all = np.tile(P, reps)[:L]
    p.pattern_offsets = np.concatenate([[0], np.cumsum(P_all)]).astype(np.int64)
    tot = int(p.pattern_offsets[-1])
    p.leafcodes = np.tile(base.leafcodes, (reps, 1))[:tot]
    p.numPhases = np.tile(base.numPhases, reps)[:tot]
    p.counts = np.tile(base.counts, reps)[:tot]
    p.mutRates = np.ones(L)
    p.L = p.numLoci = L
    return p


def write_pack(p, path):
    """text pack (same format oracle/ref_harness.c writes) so the oracle can run the same input"""
    code = "TCAGN"
    with open(path, "w") as f:
        f.write("GPHOCS-PACK 1\n")
        f.write(f"numLoci {p.L}\nnumSamples {p.n}\nnumCurPops {p.Kc}\nnumPops {p.K}\nnumMigBands {p.B}\nrootPop {p.rootPop}\n")
        f.write("samplesPerPop " + " ".join(str(int(x)) for x in p.samplesPerPop) + "\n")
        for k in range(p.K):
            f.write(f"pop {k} {p.popName[k]} {p.popFather[k]} {p.popSon0[k]} {p.popSon1[k]} "
                    f"{float(p.sampleAge[k]).hex()} {int(getattr(p, 'updateSampleAge', [0] * p.K)[k])} {float(p.thetaAlpha[k]).hex()} {float(p.thetaBeta[k]).hex()} "
                    f"{float(p.thetaStart[k]).hex()} {float(p.ageAlpha[k]).hex()} {float(p.ageBeta[k]).hex()} "
                    f"{float(p.ageStart[k]).hex()}\n")
        for b in range(p.B):
            f.write(f"band {b} {p.bandSrc[b]} {p.bandTgt[b]} {float(p.mrAlpha[b]).hex()} {float(p.mrBeta[b]).hex()}\n")
        f.write(f"mcmc {p.seed} {p.burnin} {p.numSamplesMcmc} {p.sampleSkip} {p.startMig} {p.doMixing} "
                f"{p.samplesPerLog} {p.mutRateMode}\n")
        f.write("finetunes " + " ".join(float(x).hex() for x in
                                        [p.ftCoalTime, p.ftMigTime, p.ftTheta, p.ftMigRate, p.ftMixing] + list(p.ftTaus)) + "\n")
        if p.mutRateMode == 1:
            f.write(f"locusrate {float(p.varRatesAlpha).hex()} {float(p.ftLocusRate).hex()}\n")
        f.write(f"printFactors {p.numParameters} " + " ".join(float(x).hex() for x in p.printFactors) + "\n")
        for g in range(p.L):
            o0, o1 = int(p.pattern_offsets[g]), int(p.pattern_offsets[g + 1])
            f.write(f"locus {g} {o1 - o0} {float(p.mutRates[g]).hex()}\n")
            for r in range(o0, o1):
                w = int(p.numPhases[r])      # the ABI's 16-bit word: 0x8000 | exponent for counts of 2^15 or more
                f.write("".join(code[c] for c in p.leafcodes[r]) + f" {w if w < 0x8000 else 1 << (w & 31)} {int(p.counts[r])}\n")
        f.write("end\n")
