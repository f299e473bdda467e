#!/usr/bin/env python3
"""Step 2 of tools/bbcount.sh (on the GPU): run the instrumented library on the benchmark's data set and write how often
every basic block of k_sweep was executed during ONE sweep after a pre-roll.  [LOCI=20000] [PRE=100]"""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G
import bench
import numpy as np

CAP = 1024                      # records of 8 doubles = 16384 32-bit counters


def fetch(lib, s):
    buf = (C.c_double * (8 * CAP))()
    n = C.c_int32(0)
    rc = lib.gph_engine_steplog_fetch(s.engine, 0, buf, CAP, C.byref(n), 0)
    assert rc == 0, rc
    return np.frombuffer(buf, dtype=np.uint32).copy()


def main():
    L = int(os.environ.get("LOCI", "20000"))
    pre = int(os.environ.get("PRE", "100"))
    lib = G.load_library(os.path.join(REPO, "bench_cache", "bbcount.so"))
    pack = bench.build_workload(G, 4, L, 6.5, 20261006, os.path.join(REPO, "bench_cache"))
    s = G.Sampler(pack, lib=lib)
    loci = (C.c_int64 * 1)(0)
    lib.gph_engine_steplog_enable.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32]
    lib.gph_engine_steplog_fetch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    rc = lib.gph_engine_steplog_enable(s.engine, loci, 1, CAP)
    assert rc == 0, f"steplog_enable: {rc} (library not built with -DGPH_BBCOUNT?)"
    s.initialize()
    for it in range(pre):
        s.iteration(it)
    a = fetch(lib, s)
    if os.environ.get("BB_WHAT", "sweep") == "sweep":       # one launch of k_sweep
        r = G.GphSweepResult()
        lib.gph_engine_genealogy_sweep(s.engine, 7, pack.ftCoalTime, pack.ftMigTime, C.byref(r))
    else:                                                    # one whole iteration (the evaluate kernels: BB_KERNEL names the instrumented one)
        s.iteration(pre)
    b = fetch(lib, s)
    d = (b - a).astype(np.uint32)           # modulo 2^32
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump({"loci": L, "preroll": pre, "build_id": lib.gph_build_id().decode(), "sweep_ms_instrumented": s.last_kernel_ms(0),
               "counts": {str(i): int(v) for i, v in enumerate(d) if v}}, open(os.path.join(REPO, "gpurun_out", "bbcount.json"), "w"))
    print("blocks executed:", int((d > 0).sum()), "entry count:", int(d[0]), "loci", L, "sweep ms", s.last_kernel_ms(0))
    s.close()


if __name__ == "__main__":
    main()
