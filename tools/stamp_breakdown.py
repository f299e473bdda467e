#!/usr/bin/env python3
"""Diagnostic: build the engine with -DGPH_STAMPS and print where a locus's sweep spends its cycles."""
import ctypes as C, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
# built where hipcc is cheap (the build container) into bench_cache/, which travels to the GPU box
MODE = 2 if "--lik" in sys.argv else 3 if "--spr" in sys.argv else 1
lib_path = os.path.join(REPO, "bench_cache", f"libgphocs_stamps{MODE}.so")
os.makedirs(os.path.dirname(lib_path), exist_ok=True)
srcs = [os.path.join(G.CSRC, f) for f in G.LIB_SOURCES]
deps = [os.path.join(G.CSRC, f) for f in os.listdir(G.CSRC)]
if not os.path.exists(lib_path) or any(os.path.getmtime(d) > os.path.getmtime(lib_path) for d in deps):
    subprocess.run(["hipcc"] + G.HIPCC_BASE + G.HIPCC_TUNING_SPILLING + [f"-DGPH_STAMPS={MODE}", "-DGPH_CAP_LEAVES=16", "-DGPH_CAP_K=9", "-DGPH_CAP_B=4",
                                                "-DGPH_SWEEP_WAVES=8"] + srcs + ["-o", lib_path], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
lib = G.load_library(lib_path)
L = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100000
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
s = G.Sampler(pack, lib=lib)
s.initialize()
PRE = int(os.environ.get('PRE', '100'))      # leave the prior-sampled start of the chain (as bench.py's pre-roll does)
for it in range(PRE):
    s.iteration(it)
r = G.GphSweepResult()
lib.gph_engine_genealogy_sweep(s.engine, 7, pack.ftCoalTime, pack.ftMigTime, C.byref(r))
# in the stamps build the sweep result fields carry cycle sums (see kb_sweep)
names = ["kernel body", "lik_compute", "consider_event_move", "trace_pair (both lineage walks)", "(unused)",
         "internal sweep", "spr sweep", "prune_node (inside lik_compute)"]
if MODE == 3:
    names[2:5] = ["SPR accept path", "SPR reject path", "migration-node sweep"]
if MODE == 2:
    names[2:5] = ["lik_compute: setup (tree regs, need fix-point, fence)", "lik_compute: write-back of masks",
                  "lik_compute: root reduction (log)"]
vals = [r.accepted_internal, r.accepted_mignode, r.accepted_spr, r.dData_internal, r.dLog_internal,
        r.dLog_mignode, r.dData_spr, r.dLog_spr]
for n, v in zip(names, vals):
    print("%-34s %10.0f cycles/locus  (%.1f%%)" % (n, v / L, 100.0 * v / max(vals[0], 1)))
print("sweep kernel ms", s.last_kernel_ms(0))
s.close()
