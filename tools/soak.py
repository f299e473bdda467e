#!/usr/bin/env python3
"""Long trajectory at the benchmark size with the reference's own invariant check (checkAll, patch.c:2745: every
incremental statistic, log-likelihood and conditional array of every locus against a from-scratch recomputation) every
`period` iterations -- a failed check, a chain-walk guard or any other per-locus error code aborts the iteration with
an error status.   tools/soak.py [iterations, default 3000] [period, default 50] [synthetic config, default 4]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
period = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = {4: 100000, 5: 200000, 2: 10000, 3: 40000}.get(cfg, 100000)
pack = bench.build_workload(G, cfg, L, 6.5, 20261002 + cfg, os.path.join(REPO, "bench_cache"))
pack.samplesPerLog = period
s = G.Sampler(pack, lib=G.load_library(dims=(pack.n, pack.K, pack.B)))
s.initialize()
t0 = time.perf_counter()
for it in range(iters):
    s.iteration(it)          # raises on any error status
    if (it + 1) % 500 == 0:
        c = s.counters()
        print(f"iteration {it + 1}: {c['evals'] / (time.perf_counter() - t0) / 1e6:.1f} M evals/s, accept counts {s.accept_counts()}", flush=True)
print(f"soak OK: {iters} iterations x {L} loci, checkAll every {period}: {iters // period} passes, {time.perf_counter() - t0:.1f} s")
s.close()
