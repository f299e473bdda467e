#!/usr/bin/env python3
"""BASELINE configs[4] at its stated size: 200 000 loci x 1 kb, 10 diploid samples (20 leaves), 7-population tree,
4 migration bands, one fixed ancient sample (library variant `l`).  python tools/bench_config5.py [lib.so] [loci] [preroll]
GPH_BENCH_CONFIG=2 / 3 with loci 10000 / 40000: BASELINE configs[1] / configs[2] (tools/profile_config.sh).
Same protocol as bench.py: an untimed pre-roll (default 200 iterations) before 5 warm-up and 10 timed iterations."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import gphocs_amd as G, bench
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
CFG = int(os.environ.get("GPH_BENCH_CONFIG", "5"))      # another synthetic shape (g-phocs_amd/synth.py: CONFIGS), e.g. 10 = 40 leaves
pack = bench.build_workload(G, CFG, L, 6.5, 20261002 + CFG, os.path.join(REPO, "bench_cache"))
lib = G.load_library(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-" else G.load_library(dims=(pack.n, pack.K, pack.B))
s = G.Sampler(pack, lib=lib)
PRE = int(sys.argv[3]) if len(sys.argv) > 3 else 200
s.initialize()
for it in range(PRE + 5):
    s.iteration(it)
s.counters(reset=True)
for k in range(16):
    s.class_stats(k, reset=True)
t0 = time.perf_counter()
N = 10
for it in range(PRE + 5, PRE + 5 + N):
    s.iteration(it)
dt = time.perf_counter() - t0
c = s.counters()
sw, te = s.class_stats(0), s.class_stats(1)
P = np.diff(pack.pattern_offsets)
NAMES = {2: "BASELINE configs[1]: {L} loci x 1 kb, 4 diploid samples (8 leaves), 3-population tree (5 populations), no migration",
         3: "BASELINE configs[2]: {L} loci, 6 diploid samples (12 leaves), 3-population tree (5 populations), 2 migration bands, "
            "unphased-diploid het integration",
         4: "BASELINE configs[3]: {L} loci, 8 diploid samples (16 leaves), 5-population tree (9 populations), 4 migration bands",
         5: "BASELINE configs[4]: {L} loci, 20 leaves, 13 populations, 4 bands, fixed ancient sample"}
print(json.dumps({"workload": NAMES[CFG].format(L=L) if CFG in NAMES else
                              f"synthetic config {CFG}: {L} loci, {pack.n} leaves, {pack.K} populations, {pack.B} bands",
                  "loci": L, "leaves": int(pack.n), "populations": int(pack.K), "bands": int(pack.B),
                  "evals_per_locus_per_sweep": sw["evals"] / sw["launches"] / L if "evals" in sw else None,
                  "resident_wavefront_rounds": L / (256 * 32.0),
                  "mean_phased_patterns": float(P.mean()), "max_phased_patterns": int(P.max()),
                  "evals_per_s": c["evals"] / dt, "iters_per_s": N / dt, "ms_per_iteration": dt / N * 1e3,
                  "sweep_ms": sw["ms"] / sw["launches"], "sweep_algorithmic_bytes": sw["bytes"] / sw["launches"],
                  "sweep_roofline_frac": sw["bytes"] / sw["launches"] / (sw["ms"] / sw["launches"] * 1e-3) / 8e12,
                  "tau_eval_ms": te["ms"] / max(te["launches"], 1), "hbm_resident_bytes": s.hbm_bytes(),
                  "preroll_iterations": PRE, "library_build_id": lib.gph_build_id().decode()}))
s.close()
