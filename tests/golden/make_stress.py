#!/usr/bin/env python3
"""Stress input for the sequence front end: heavy heterozygosity (2-way and 3-way IUPAC codes),
haploid + diploid samples, N runs, all-N columns, lower case, samples missing from some loci,
sequences of samples the control file does not name, blank lines.  Writes stress.ctl / stress.seq;
make_goldens.sh turns them into stress.gpk with the REAL reference (oracle/_ref/gphocs_ref pack)."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
import gen_synth  # noqa: E402

rng = random.Random(20240611)
cfg = dict(pops=[2, 2, 2], bands=[(0, 1)], loci=40)
gen_synth.write_ctl(os.path.join(HERE, "stress.ctl"), cfg, "stress.seq", "stress.trace", 40, 4242, 10, 5)
# turn one diploid of pop B into two haploids and add a haploid to pop C
txt = open(os.path.join(HERE, "stress.ctl")).read()
txt = txt.replace("s2 d s3 d", "s2 d h3a h h3b h").replace("s4 d s5 d", "s4 d s5 d h6 h")
open(os.path.join(HERE, "stress.ctl"), "w").write(txt)
dip = ["s0", "s1", "s2", "s4", "s5"]
hap = ["h3a", "h3b", "h6"]
TWO = "YWKMSR"
THREE = "VDBH"
out = ["40", ""]  # (a blank FIRST line is an error upstream, AlignmentProcessor.c:514-524)
for locus in range(40):
    length = rng.choice([30, 60, 120, 200])
    cols = []
    anc = rng.choice("TCAG")
    for site in range(length):
        r = rng.random()
        if r < 0.06:
            cols.append(None)  # all-N column
            continue
        alt = rng.choice([b for b in "TCAG" if b != anc])
        col = {}
        for s in dip:
            u = rng.random()
            if u < 0.55:
                col[s] = anc
            elif u < 0.70:
                col[s] = alt
            elif u < 0.90:
                col[s] = rng.choice(TWO)
            elif u < 0.94:
                col[s] = rng.choice(THREE)
            else:
                col[s] = "N"
        for s in hap:
            u = rng.random()
            col[s] = anc if u < 0.6 else alt if u < 0.9 else "N"
        cols.append(col)
        if rng.random() < 0.3:
            anc = rng.choice("TCAG")
    # repeat some columns so that het patterns with count > 1 exist
    for _ in range(length // 6):
        i, j = rng.randrange(length), rng.randrange(length)
        cols[j] = cols[i]
    present = [s for s in dip + hap if rng.random() < 0.85] or ["s0"]
    extra = ["ghost1"] if rng.random() < 0.4 else []
    names = present + extra
    rng.shuffle(names)
    out.append(f"locus{locus}\t{len(names)} {length}")
    for s in names:
        if s.startswith("ghost"):
            seq = "".join(rng.choice("TCAGN") for _ in range(length))
        else:
            seq = "".join("N" if c is None else c[s] for c in cols)
        if rng.random() < 0.3:
            seq = seq.lower()
        out.append(f"{s}\t{seq}" + ("   trailing words" if rng.random() < 0.1 else ""))
    if rng.random() < 0.5:
        out.append("")
# make sure every named sample occurs at least once
open(os.path.join(HERE, "stress.seq"), "w").write("\n".join(out) + "\n")
