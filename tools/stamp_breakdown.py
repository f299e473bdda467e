#!/usr/bin/env python3
"""Diagnostic: build the engine with -DGPH_STAMPS and print where a locus's sweep spends its cycles."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
lib_path = os.path.join(REPO, "gpurun_out", "libgphocs_stamps.so")
os.makedirs(os.path.dirname(lib_path), exist_ok=True)
srcs = [os.path.join(G.CSRC, "gph_engine.hip"), os.path.join(G.CSRC, "gph_mcmc.cpp")]
subprocess.run(["hipcc"] + G.HIPCC_FLAGS + ["-DGPH_STAMPS"] + srcs + ["-o", lib_path], check=True)
lib = G.load_library(lib_path)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
s = G.Sampler(pack, lib=lib)
s.initialize()
for it in range(2):
    s.iteration(it)
r = G.GphSweepResult()
lib.gph_engine_genealogy_sweep(s.engine, 7, pack.ftCoalTime, pack.ftMigTime, C.byref(r))
# in the stamps build the sweep result fields carry cycle sums (see kb_sweep)
names = ["kernel body", "lik_compute", "consider_event_move", "trace_lineage<0>", "trace_lineage<1>",
         "internal sweep", "spr sweep", "prune_node (inside lik_compute)"]
vals = [r.accepted_internal, r.accepted_mignode, r.accepted_spr, r.dData_internal, r.dLog_internal,
        r.dLog_mignode, r.dData_spr, r.dLog_spr]
for n, v in zip(names, vals):
    print("%-34s %10.0f cycles/locus  (%.1f%%)" % (n, v / L, 100.0 * v / max(vals[0], 1)))
print("sweep kernel ms", s.last_kernel_ms(0))
s.close()
