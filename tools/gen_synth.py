#!/usr/bin/env python3
"""Synthetic G-PhoCS inputs (own code, seeded, deterministic): a control file in the
reference's unchanged format (MCMCcontrol.c:575-1256) plus a sequence file in the
reference's unchanged format (AlignmentProcessor.c:468-860).

Recipe = SURVEY.md section 8(d): per locus, simulate a genealogy under the config's
population tree at the prior means (theta = alpha/beta, tau = tau-initial, no migration),
drop JC69 mutations (mu = 1) on a `seqlen`-bp sequence, pair haplotypes into diploids
(IUPAC het codes Y R M K S W), mask `nmask` of genotypes as N.

Configs mirror BASELINE.json `configs` (index 1..5); --loci overrides L so the same
shapes can be produced at fixture size.
"""
import argparse
import math
import os
import sys

import numpy as np

BASES = "TCAG"
IUPAC = {frozenset("CT"): "Y", frozenset("AG"): "R", frozenset("AC"): "M",
         frozenset("GT"): "K", frozenset("CG"): "S", frozenset("AT"): "W"}

# name -> (diploid samples per current pop, #bands, ancient-pop index or None)
CONFIGS = {
    1: dict(pops=[1, 1, 1, 1], bands=[("D", "B")], loci=1000, sample_ctl=True),
    2: dict(pops=[2, 1, 1], bands=[], loci=10000),
    3: dict(pops=[2, 2, 2], bands=[(0, 1), (2, 1)], loci=40000),
    4: dict(pops=[2, 2, 2, 1, 1], bands=[(0, 1), (1, 0), (3, 2), (4, 3)], loci=100000),
    5: dict(pops=[2, 2, 2, 1, 1, 1, 1], bands=[(0, 1), (1, 0), (3, 2), (4, 3)], loci=200000,
            ancient=6),
    # estimated ("e") sample ages: UpdateSampleAge is live, mixing stays on
    6: dict(pops=[2, 2, 2, 1], bands=[(0, 1), (3, 2), (2, 3)], loci=1000, ancient=3, ancient_est=True),
    7: dict(pops=[2, 1, 2], bands=[(0, 1), (1, 0), (2, 1)], loci=1000, ancient=1, ancient_est=True),
    # the engine's hard caps: 32 leaves (16 diploids, one per population), 31 populations, 16 migration bands
    8: dict(pops=[1] * 16, bands=[(i, i + 1) for i in range(8)] + [(i + 1, i) for i in range(8)], loci=1000, tau_factor=1.3),
    # the reference's own population cap (NSPECIES 20, patch.h:19): 20 current populations (39 in all), 40 leaves, 16 bands
    9: dict(pops=[1] * 20, bands=[(i, i + 1) for i in range(8)] + [(i + 1, i) for i in range(8)], loci=1000, tau_factor=1.25),
    # more than 16 migration bands (the reference allows MAX_MIG_BANDS 100, patch.h:17): 6 current populations, 12 leaves, 20 bands
    12: dict(pops=[1] * 6, bands=[(i, i + 1) for i in range(5)] + [(i + 1, i) for i in range(5)] + [(i, i + 2) for i in range(4)] +
             [(i + 2, i) for i in range(4)] + [(0, 3), (3, 0)], loci=1000, tau_factor=1.6),
    # more than 64 leaves (the reference allows NS 200, patch.h:22): 36 diploids over 6 populations = 72 leaves, 4 bands
    13: dict(pops=[6] * 6, bands=[(0, 1), (1, 0), (3, 2), (4, 3)], loci=1000, tau_factor=1.6),
    # --- population trees that are NOT caterpillars, migration bands with ANCESTRAL endpoints (round 6).  `tree` is a nested
    # tuple over the current populations' names; an ancestral population is named by the sorted names below it ("root" at the top);
    # its tau-initial is 5e-6 * 2^(height-1), cousins spread by 7 % so that no two splits coincide.  These shapes reach what a
    # caterpillar with leaf-to-leaf bands cannot: a band whose start time moves with a tau (UpdateTau types 1b / 2a / 3,
    # GPhoCS.c:3353-3431), tau bounds set by two ancestral sons (GPhoCS.c:3266-3267), rubberBandRipple with start_or_end == 1
    # (patch.c:815-869)
    20: dict(pops=[1, 1, 1, 1, 1], tree=((("A", "B"), ("C", "D")), "E"),
             bands=[("AB", "CD"), ("CD", "AB"), ("C", "AB"), ("E", "ABCD")], loci=1000),
    21: dict(pops=[1, 1, 1, 1, 1, 1], tree=(("A", ("B", "C")), (("D", "E"), "F")),
             bands=[("A", "BC"), ("BC", "A"), ("DE", "F"), ("F", "DE"), ("ABC", "DEF"), ("DEF", "ABC"), ("B", "C"), ("D", "BC")],
             loci=1000, tau_base=4e-6),
    22: dict(pops=[1, 1, 2, 1], tree=(("A", "B"), ("C", "D")), bands=[("AB", "CD"), ("CD", "AB"), ("D", "C"), ("A", "B")],
             loci=1000, ancient=2, ancient_est=True),
}


def build_tree(cfg):
    """the population tree of a configuration: current names, and per ancestral population (in the order of the control
    file: children before parents, root last) its name, two children names and tau-initial.  Without a `tree` entry: the
    caterpillar ((((A,B),C),D),...) with tau-initial doubling per level.  Returns (cur, anc, taus, children)."""
    kc = len(cfg["pops"])
    cur = [chr(ord("A") + i) for i in range(kc)]
    if "tree" not in cfg:
        anc, children = [], []
        name, prev = cur[0], cur[0]
        for i in range(1, kc):
            name = name + cur[i]
            anc.append(name if i < kc - 1 else "root")
            children.append((prev, cur[i]))
            prev = anc[-1]
        taus = []
        t = 5e-6
        for i in range(kc - 1):
            taus.append(t)
            t *= cfg.get("tau_factor", 2.0) if i < kc - 3 else 5.0 if i == kc - 3 else 1.0
        return cur, anc, taus, children
    anc, taus, children = [], [], []
    base = cfg.get("tau_base", 5e-6)
    explicit = cfg.get("taus", {})

    def walk(t):      # -> (name, leaves below, height)
        if isinstance(t, str):
            assert t in cur, t
            return t, [t], 0
        assert len(t) == 2
        a, la, ha = walk(t[0])
        b, lb, hb = walk(t[1])
        leaves = sorted(la + lb)
        h = max(ha, hb) + 1
        nm = "".join(leaves)
        anc.append(nm)
        children.append((a, b))
        taus.append(explicit.get(nm, base * 2.0 ** (h - 1) * (1.0 + 0.07 * (len(anc) - 1))))
        return nm, leaves, h
    _, leaves, _ = walk(cfg["tree"])
    assert leaves == cur, "the tree must name every current population once"
    anc[-1] = "root"
    if "root" in explicit:
        taus[-1] = explicit["root"]
    return cur, anc, taus, children


def ancient_pops(cfg):
    a = cfg.get("ancient")
    return [] if a is None else list(a) if isinstance(a, (list, tuple)) else [a]


def write_ctl(path, cfg, seqfile, tracefile, loci, seed, iters, samples_per_log, no_mixing=False,
              start_mig=0, mig_beta=0.00001, var_rates=None, fixed_rates=None):
    cur, anc, taus, children = build_tree(cfg)
    out = []
    out.append("GENERAL-INFO-START\n")
    out.append(f"\tseq-file            {seqfile}")
    out.append(f"\ttrace-file          {tracefile}")
    out.append("\tlocus-mut-rate          " + (f"FIXED {fixed_rates}" if fixed_rates else
                                            "CONST" if var_rates is None else f"VAR {var_rates[0]}"))
    out.append(f"\tnum-loci            {loci}")
    out.append(f"\trandom-seed         {seed}")
    out.append(f"\tmcmc-iterations\t  {iters}")
    out.append(f"\titerations-per-log  {samples_per_log}")
    out.append("\tlogs-per-line       10")
    if start_mig:
        out.append(f"\tstart-mig           {start_mig}")
    if no_mixing:
        out.append("\tno-mixing           1")
    out.append("")
    out.append("\tfind-finetunes\t\tFALSE")
    out.append("\tfinetune-coal-time\t0.01\t\t")
    out.append("\tfinetune-mig-time\t0.3\t\t")
    out.append("\tfinetune-theta\t\t0.04")
    out.append("\tfinetune-mig-rate\t0.02")
    out.append("\tfinetune-tau\t\t0.0000008")
    out.append("\tfinetune-mixing\t\t0.003")
    if var_rates is not None:
        out.append(f"\tfinetune-locus-rate\t{var_rates[1]}")
    out.append("")
    out.append("\ttau-theta-print\t\t10000.0")
    out.append("\ttau-theta-alpha\t\t1.0")
    out.append("\ttau-theta-beta\t\t10000.0")
    out.append("")
    out.append("\tmig-rate-print\t\t0.001")
    out.append("\tmig-rate-alpha\t\t0.002")
    out.append(f"\tmig-rate-beta\t\t{mig_beta:.10f}")
    out.append("\nGENERAL-INFO-END\n")
    out.append("CURRENT-POPS-START\t\n")
    sid = 0
    for i, nm in enumerate(cur):
        out.append("\tPOP-START")
        out.append(f"\t\tname\t\t{nm}")
        samples = " ".join(f"s{sid + j} d" for j in range(cfg["pops"][i]))
        sid += cfg["pops"][i]
        out.append(f"\t\tsamples\t\t{samples}")
        if i in ancient_pops(cfg):
            out.append("\t\tage\t\t0.000002 " + ("e" if cfg.get("ancient_est") else "f"))
        out.append("\tPOP-END\n")
    out.append("CURRENT-POPS-END\n")
    out.append("ANCESTRAL-POPS-START\n")
    for i, nm in enumerate(anc):
        out.append("\tPOP-START")
        out.append(f"\t\tname\t\t\t{nm}")
        out.append(f"\t\tchildren\t\t{children[i][0]}\t\t{children[i][1]}")
        out.append(f"\t\ttau-initial\t{taus[i]:.9f}")
        out.append("\t\ttau-beta\t\t20000.0\t")
        ft = 0.0000008 if i < len(anc) - 1 else 0.00000286
        out.append(f"\t\tfinetune-tau\t\t\t{ft:.8f}")
        out.append("\tPOP-END\n")
    out.append("ANCESTRAL-POPS-END\n")
    if cfg["bands"]:
        out.append("MIG-BANDS-START\t")
        for (s, t) in cfg["bands"]:
            sn = s if isinstance(s, str) else cur[s]
            tn = t if isinstance(t, str) else cur[t]
            out.append("\tBAND-START\t\t")
            out.append(f"       source  {sn}")
            out.append(f"       target  {tn}")
            out.append("       mig-rate-print 0.1")
            out.append("\tBAND-END\n")
        out.append("MIG-BANDS-END")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def simulate_locus(rng, cfg, taus, theta, seqlen, ancient_age):
    """returns list of haploid sequences (np.uint8 arrays of base indices), leaf order =
    population order then sample order, two haploids per diploid sample"""
    kc = len(cfg["pops"])
    # nodes: list of (age, left, right); leaves first
    nleaf = 2 * sum(cfg["pops"])
    age = [0.0] * nleaf
    left = [-1] * nleaf
    right = [-1] * nleaf
    leaf = 0
    pop_lineages = {}
    cur, anc, _, children = build_tree(cfg)
    for i in range(kc):
        k = 2 * cfg["pops"][i]
        a0 = ancient_age if i in ancient_pops(cfg) else 0.0
        for j in range(k):
            age[leaf + j] = a0
        pop_lineages[cur[i]] = (list(range(leaf, leaf + k)), a0)
        leaf += k

    def coalesce(lins, t0, t1):
        t = t0
        lins = list(lins)
        while len(lins) > 1:
            k = len(lins)
            t += rng.exponential(theta / (k * (k - 1.0)))
            if t1 is not None and t > t1:
                break
            a, b = rng.choice(k, size=2, replace=False)
            na, nb = lins[a], lins[b]
            age.append(t)
            left.append(na)
            right.append(nb)
            new = len(age) - 1
            lins = [x for idx, x in enumerate(lins) if idx not in (a, b)] + [new]
        return lins

    # populations in the order of the control file (children before parents); a population's lineages coalesce from its
    # start (sample age / its tau) to its father's tau.  On a caterpillar this is the order A, B, AB, C, ABC, ...
    tau_of = dict(zip(anc, taus))
    top_of = {}
    for nm, (a, b) in zip(anc, children):
        top_of[a] = top_of[b] = tau_of[nm]

    def run_pop(nm):
        if nm in pop_lineages:
            l0, t0 = pop_lineages[nm]
        else:
            a, b = children[anc.index(nm)]
            l0, t0 = run_pop(a) + run_pop(b), tau_of[nm]
        return coalesce(l0, t0, top_of.get(nm))
    lins = run_pop(anc[-1])
    root = lins[0]
    seqs = {}
    seqs[root] = rng.integers(0, 4, size=seqlen, dtype=np.uint8)
    stack = [root]
    while stack:
        nd = stack.pop()
        for ch in (left[nd], right[nd]):
            if ch < 0:
                continue
            bl = age[nd] - age[ch]
            p = 0.75 * (1.0 - math.exp(-4.0 * bl / 3.0))
            s = seqs[nd].copy()
            hit = rng.random(seqlen) < p
            nh = int(hit.sum())
            if nh:
                s[hit] = (s[hit] + rng.integers(1, 4, size=nh, dtype=np.uint8)) % 4
            seqs[ch] = s
            stack.append(ch)
    return [seqs[i] for i in range(nleaf)]


def genotype_strings(haps, rng, nmask):
    nd = len(haps) // 2
    out = []
    for d in range(nd):
        a, b = haps[2 * d], haps[2 * d + 1]
        chars = np.array(list(BASES))[a].copy()
        het = a != b
        for i in np.nonzero(het)[0]:
            chars[i] = IUPAC[frozenset((BASES[a[i]], BASES[b[i]]))]
        mask = rng.random(len(a)) < nmask
        chars[mask] = "N"
        out.append("".join(chars))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS))
    ap.add_argument("--model-json", default=None,
                    help="a configuration as a JSON object instead of --config (keys as in CONFIGS; `tree` as nested lists; "
                         "`data_seed` seeds the sequences): what tools/random_models.py writes")
    ap.add_argument("--loci", type=int, default=None)
    ap.add_argument("--seqlen", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--per-log", type=int, default=100)
    ap.add_argument("--mcmc-seed", type=int, default=12345)
    ap.add_argument("--nmask", type=float, default=0.002)
    ap.add_argument("--no-mixing", action="store_true")
    ap.add_argument("--start-mig", type=int, default=0)
    ap.add_argument("--mig-beta", type=float, default=0.00001,
                    help="mig-rate-beta (prior mean = 0.002/beta); small beta = many migration events")
    ap.add_argument("--mut-scale", type=float, default=1.0,
                    help="scale branch lengths when dropping mutations (more patterns)")
    ap.add_argument("--var-rates", type=float, nargs=2, default=None, metavar=("ALPHA", "FINETUNE"),
                    help="locus-mut-rate VAR <ALPHA> with finetune-locus-rate <FINETUNE> (UpdateLocusRate is live)")
    ap.add_argument("--fixed-rates", action="store_true",
                    help="locus-mut-rate FIXED <out>.rates: a rate file with one rate per locus, spread over 0.2 .. 5 "
                         "(readRateFile, GPhoCS.c:491-579, normalises them to mean 1)")
    ap.add_argument("--out", required=True, help="output prefix: <out>.ctl, <out>.seq")
    a = ap.parse_args()
    if a.model_json:
        import json
        cfg = json.load(open(a.model_json))

        def tup(t):
            return t if isinstance(t, str) else tuple(tup(x) for x in t)
        if "tree" in cfg:
            cfg["tree"] = tup(cfg["tree"])
        cfg["bands"] = [tuple(b) for b in cfg.get("bands", [])]
        a.config = int(cfg.get("data_seed", 0))
    else:
        assert a.config is not None, "--config or --model-json"
        cfg = CONFIGS[a.config]
    L = a.loci or cfg["loci"]
    rng = np.random.default_rng(20261002 + a.config)
    cur, anc, taus, _ = build_tree(cfg)
    theta = 1e-4
    seqfile = os.path.basename(a.out) + ".seq"
    write_ctl(a.out + ".ctl", cfg, seqfile, os.path.basename(a.out) + ".trace", L, a.mcmc_seed,
              a.iters, a.per_log, no_mixing=a.no_mixing, start_mig=a.start_mig, mig_beta=a.mig_beta,
              var_rates=a.var_rates, fixed_rates=os.path.basename(a.out) + ".rates" if a.fixed_rates else None)
    if a.fixed_rates:
        # log-uniform over 0.2 .. 5, mixed layout (several per line, tabs, an exponent form): the reader is fscanf("%lf")
        rr = np.exp(np.random.default_rng(77 + a.config).uniform(np.log(0.2), np.log(5.0), L))
        with open(a.out + ".rates", "w") as f:
            for g in range(L):
                f.write(("%.6e" % rr[g]) if g % 5 == 3 else ("%.5f" % rr[g]))
                f.write("\n" if g % 4 == 3 else "\t" if g % 2 else " ")
            f.write("\n")
    nd = sum(cfg["pops"])
    with open(a.out + ".seq", "w") as f:
        f.write(f"{L}\n\n")
        for g in range(L):
            haps = simulate_locus(rng, cfg, [t * a.mut_scale for t in taus], theta * a.mut_scale,
                                  a.seqlen, 0.000002 * a.mut_scale)
            gts = genotype_strings(haps, rng, a.nmask)
            f.write(f"locus{g + 1} {nd} {a.seqlen}\n")
            for d in range(nd):
                f.write(f"s{d}\t{gts[d]}\n")
            f.write("\n")
    print(f"wrote {a.out}.ctl {a.out}.seq  L={L} samples={nd} diploid", file=sys.stderr)


if __name__ == "__main__":
    main()
