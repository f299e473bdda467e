#!/usr/bin/env python3
"""Long trajectories with the reference's own invariant check (checkAll, patch.c:2745: every incremental statistic,
log-likelihood and conditional array of every locus against a from-scratch recomputation) every `period` iterations -- a
failed check, a chain-walk guard or any other per-locus error code aborts the iteration with an error status.

    tools/soak.py [iterations=3000] [period=50] [j-iterations=1000] [j-loci=2000]   ->   gpurun_out/soak.json

Leg 1: the benchmark's data set (BASELINE configs[3], 100 000 loci).  Legs 2-4 (round 6): the model shapes j1 / j2 / j3 of
tools/gen_synth.py (balanced / mixed population trees, migration bands with ancestral ends, an estimated ancient sample) at
`j-loci` loci with a high-migration prior (mig-rate-beta 4e-8), read through the library's own front end from a control +
sequence file.  Per leg: evaluations / s, checkAll passes, accept counts, migration events held at the end, rubber-band
conflicts."""
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gphocs_amd as G  # noqa: E402
import bench  # noqa: E402


def run_leg(name, pack, iters, period):
    pack.samplesPerLog = period
    s = G.Sampler(pack, lib=G.load_library(dims=(pack.n, pack.K, pack.B)))
    s.initialize()
    t0 = time.perf_counter()
    for it in range(iters):
        s.iteration(it)          # raises on any error status (a failed checkAll is one)
        if (it + 1) % 500 == 0:
            c = s.counters()
            print(f"{name}: iteration {it + 1}: {c['evals'] / (time.perf_counter() - t0) / 1e6:.1f} M evals/s, accept counts {s.accept_counts()}", flush=True)
    dt = time.perf_counter() - t0
    c = s.counters()
    hs = s.host_stats() if hasattr(s, "host_stats") else {}
    leg = dict(leg=name, loci=int(pack.L), leaves=int(pack.n), populations=int(pack.K), bands=int(pack.B), iterations=iters, checkall_period=period,
               checkall_passes=iters // period, seconds=dt, evals=float(c["evals"]), evals_per_s=c["evals"] / dt, iterations_per_s=iters / dt,
               accept_counts=[int(x) for x in s.accept_counts()], host_stats={k: (int(v) if isinstance(v, (int, bool)) else v) for k, v in hs.items()})
    s.close()
    print(f"{name}: OK, {iters} iterations x {pack.L} loci, {iters // period} checkAll passes, {dt:.1f} s", flush=True)
    return leg


def j_pack(cfg, loci, td):
    out = os.path.join(td, f"soak{cfg}")
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_synth.py"), "--config", str(cfg), "--loci", str(loci), "--seqlen", "300",
                    "--iters", "1000", "--per-log", "50", "--mig-beta", "0.00000004", "--out", out], check=True, capture_output=True)
    return G.Pack.from_control(out + ".ctl", seq_path=out + ".seq")


def main():
    a = sys.argv
    iters = int(a[1]) if len(a) > 1 else 3000
    period = int(a[2]) if len(a) > 2 else 50
    jit = int(a[3]) if len(a) > 3 else 1000
    jloci = int(a[4]) if len(a) > 4 else 2000
    G.build()
    legs = []
    if iters > 0:
        pack = bench.build_workload(G, 4, 100000, 6.5, 20261002 + 4, os.path.join(REPO, "bench_cache"))
        legs.append(run_leg("configs[3] (100 000 loci, 16 leaves, 9 populations, 4 bands)", pack, iters, period))
    with tempfile.TemporaryDirectory() as td:
        for cfg, what in ((20, "j1 (((A,B),(C,D)),E), bands AB->CD CD->AB C->AB E->ABCD"), (21, "j2 ((A,(B,C)),((D,E),F)), 8 mixed bands"),
                          (22, "j3 ((A,B),(C,D)), estimated ancient sample in C, bands AB->CD CD->AB D->C A->B")):
            if jit > 0:
                legs.append(run_leg(what, j_pack(cfg, jloci, td), jit, period))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    out = dict(what="soak: long trajectories, checkAll (patch.c:2745) every `checkall_period` iterations, any error status aborts",
               build=open(G.LIB_PATH + ".buildid").read().strip() if os.path.exists(G.LIB_PATH + ".buildid") else None, legs=legs)
    with open(os.path.join(REPO, "gpurun_out", "soak.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("soak OK")


if __name__ == "__main__":
    main()
