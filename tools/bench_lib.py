#!/usr/bin/env python3
"""A/B helper: run the benchmark workload against an alternative build of the library."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
lib = G.load_library(sys.argv[1])
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
s = G.Sampler(pack, lib=lib)
s.initialize()
for it in range(3):
    s.iteration(it)
s.counters(reset=True)
for k in range(16):
    s.class_stats(k, reset=True)
t0 = time.perf_counter()
for it in range(3, 11):
    s.iteration(it)
dt = time.perf_counter() - t0
c = s.counters()
sw = s.class_stats(0)
te, me = s.class_stats(1), s.class_stats(2)
print(f"{sys.argv[1]}: {c['evals']/dt/1e6:.1f} M evals/s, {8/dt:.2f} it/s, sweep {sw['ms']/sw['launches']:.2f} ms, "
      f"tau_eval {te['ms']/max(te['launches'],1):.3f} ms x{te['launches']/8:.0f}, mix_eval {me['ms']/max(me['launches'],1):.3f} ms")
s.close()
