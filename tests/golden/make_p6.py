#!/usr/bin/env python3
"""p6: a locus with a REPEATED alignment column of 16 heterozygotes among 18 diploid samples (36 leaves).  A column that occurs
more than once gets no symmetry break (AlignmentProcessor.c:1767: `if(patternCounts[patt] > 1) continue`), so it expands into
2^16 = 65 536 phased patterns (processHetPatterns, AlignmentProcessor.c:998-1158): the phase count no longer fits 16 bits as a
plain number (the engine stores 0x8000 | exponent, include/gphocs_hip.h: GPH_NUMPHASES), and the locus has 65 554 phased
patterns -- its sequence block (1.4 MB) and conditional arrays (147 MB) live in HBM.  Writes p6.ctl / p6.seq (own generator);
make_goldens.sh then runs the real reference on them for p6.trace."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
import gen_synth  # noqa: E402

cfg = dict(pops=[6, 6, 6], bands=[(0, 1)], loci=2)
gen_synth.write_ctl(os.path.join(HERE, "p6.ctl"), cfg, "p6.seq", "p6.trace", 2, 4242, 4, 2, mig_beta=1e-5)
rng = np.random.default_rng(5)
L = 60
with open(os.path.join(HERE, "p6.seq"), "w") as f:
    f.write("2\n\n")
    for g in range(2):
        f.write(f"locus{g + 1} 18 {L}\n")
        base = rng.integers(0, 4, L)
        for d in range(18):
            s = ["TCAG"[b] for b in base]
            for i in rng.choice(np.arange(20, L), 2, replace=False):
                s[i] = "TCAG"[rng.integers(0, 4)]
            if g == 0:
                s[10] = s[11] = "Y" if d < 16 else "T"     # the repeated column with 16 hets
            f.write(f"s{d}\t{''.join(s)}\n")
        f.write("\n")
