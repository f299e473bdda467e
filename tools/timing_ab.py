#!/usr/bin/env python3
"""cost of the HIP-event brackets round every launch: step time with all kernel classes timed / only the sweep / none"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G, bench
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
s = G.Sampler(pack)
s.initialize()
it = 0
for _ in range(5):
    s.iteration(it); it += 1
for rep in range(2):
    for mask, name in ((0xffff, "all classes"), (1, "sweep only"), (0, "none")):
        s.set_timing(mask)
        for _ in range(3):
            s.iteration(it); it += 1
        t0 = time.perf_counter()
        for _ in range(30):
            s.iteration(it); it += 1
        print(f"L={L} timing {name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
s.close()
