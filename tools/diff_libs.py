#!/usr/bin/env python3
"""Debugging aid: run the same chain with two builds of the library and report the first record that differs.
   python3 tools/diff_libs.py a.so b.so [loci] [iters] [config]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
from gphocs_amd_pkg import synth
a, b = sys.argv[1], sys.argv[2]
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
config = int(sys.argv[5]) if len(sys.argv) > 5 else 4
pk = synth.make_synthetic_pack(G.Pack, config, L, mut_scale=6.5, data_seed=4711, mcmc_seed=4242, samples_per_log=8)
recs = []
for p in (a, b):
    lib = G._load_library(p)
    s = G.Sampler(pk, lib=lib)
    out = f"/tmp/diff_{os.path.basename(p)}.rec"
    s.set_record_file(out)
    s.initialize()
    try:
        for it in range(iters):
            s.iteration(it)
    except RuntimeError as ex:
        print(p, "FAILED:", ex)
    s.set_record_file(None)
    s.close()
    recs.append(open(out).read().splitlines())
n = min(len(recs[0]), len(recs[1]))
for i in range(n):
    if recs[0][i] != recs[1][i]:
        print("first difference at record", i)
        print(" ", recs[0][i][:300])
        print(" ", recs[1][i][:300])
        break
else:
    print("identical over", n, "records;", len(recs[0]), len(recs[1]))
