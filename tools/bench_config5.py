import os, sys, time
sys.path.insert(0, os.getcwd())
import gphocs_amd as G, bench
lib = G.load_library(sys.argv[1])
L = 50000
pack = bench.build_workload(G, 5, L, 3.0, 20261007, "bench_cache")
s = G.Sampler(pack, lib=lib); s.initialize()
for it in range(3): s.iteration(it)
s.counters(reset=True)
for k in range(16): s.class_stats(k, reset=True)
t0 = time.perf_counter()
for it in range(3, 9): s.iteration(it)
dt = time.perf_counter() - t0
c = s.counters(); sw = s.class_stats(0)
import numpy as np
print(f"{sys.argv[1]}: config5 L={L} P mean {np.diff(pack.pattern_offsets).mean():.1f}: {c['evals']/dt/1e6:.1f} M evals/s, {6/dt:.2f} it/s, sweep {sw['ms']/sw['launches']:.2f} ms")
s.close()
