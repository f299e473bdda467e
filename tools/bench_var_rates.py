#!/usr/bin/env python3
"""Timing of `locus-mut-rate VAR` iterations (UpdateLocusRate = one-wavefront serial scan + parallel write-back) on the
benchmark workload: python3 tools/bench_var_rates.py [loci] [iterations]"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
from gphocs_amd_pkg import synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
synth.make_var_rates(pack, 1.0, 0.3)
s = G.Sampler(pack, lib=G.load_library(os.environ["GPH_LIB"]) if os.environ.get("GPH_LIB") else None)
s.initialize()
for it in range(2):
    s.iteration(it)
ms = C.c_double()
t0 = time.time()
scan = apply_ = 0.0
for it in range(2, 2 + iters):
    s.iteration(it)
    s.lib.gph_engine_last_kernel_ms(s.engine, 9, C.byref(ms)); scan += ms.value
    s.lib.gph_engine_last_kernel_ms(s.engine, 10, C.byref(ms)); apply_ += ms.value
dt = (time.time() - t0) / iters
acc, rv = C.c_int64(), C.c_double()
s.lib.gph_mcmc_locus_rate_state(s.mcmc, C.byref(acc), C.byref(rv))
print(f"{L} loci: {dt * 1e3:.1f} ms/iteration; locus-rate scan {scan / iters:.1f} ms ({scan / iters / (L - 1) * 1e3:.2f} us/locus), "
      f"write-back {apply_ / iters:.2f} ms; accepted {acc.value} of {(L - 1) * (iters + 2)}, rate variance {rv.value:.5f}")
s.close()
