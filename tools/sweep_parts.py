#!/usr/bin/env python3
"""Diagnostic: time the fused sweep kernel with subsets of its proposal classes (flags: 1 node ages, 2 migration-event
ages, 4 SPR; 0 = stage the locus in and out only)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
s = G.Sampler(pack)
s.initialize()
for it in range(3):
    s.iteration(it)
r = G.GphSweepResult()
for flags in (0, 0, 1, 2, 4, 7, 7):
    s.lib.gph_engine_genealogy_sweep(s.engine, flags, pack.ftCoalTime, pack.ftMigTime, C.byref(r))
    print(f"flags {flags}: {s.last_kernel_ms(0):.3f} ms")
s.close()
