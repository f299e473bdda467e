#!/usr/bin/env python3
"""BASELINE configs[3] at its stated size for ONE full log period -- 100 000 loci, 100 iterations, checkAll at iteration 99 --
on the MI355X against the oracle restatement run live on ALL loci (closes the gap between "8
iterations exact" and "400 iterations of invariants").  About 10 minutes of CPU for the oracle: a script, not a -m gpu test.

   gpurun --timeout 2400 -- 'python3 tools/full_size_100it.py > gpurun_out/full_size_100it.log 2>&1'
   -> gpurun_out/full_size_100it.json   (copy to profiles/rNN_full_size_100it.json)

Per iteration: accept counters of every proposal equal (asserted), worst relative difference of the accumulators
(dataLogLikelihood, logLikelihood; bar 1e-10); per-locus state of every 50th locus after the last iteration, field by field;
wall times of both sides.  The oracle is the checker here (test infrastructure): nothing of it is on the measured path."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    loci = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    out_dir = os.path.join(REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    import gphocs_amd as G
    import bench
    from gphocs_amd_pkg import synth
    from parity_util import compare_states, _close
    G.build()
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True, capture_output=True)
    oracle = os.path.join(REPO, "oracle", "gphocs_oracle")
    pk = bench.build_workload(G, 4, loci, 6.5, 20261002 + 4, os.path.join(REPO, "bench_cache"))
    assert int(pk.samplesPerLog) == 100, "the workload's stated log period"
    tmp = os.environ.get("TMPDIR", "/tmp")
    pth, mine, theirs = (os.path.join(tmp, "fs100." + x) for x in ("gpk", "hip.rec", "oracle.rec"))
    synth.write_pack(pk, pth)
    os.environ["GPH_DUMP_STRIDE"] = "50"
    t0 = time.time()
    s = G.Sampler(pk)
    s.set_record_file(mine)
    s.initialize()
    for it in range(iters):
        s.iteration(it)
    s.dump_state(mine + ".state", False)
    s.set_record_file(None)
    cnt, acc = s.counters(), s.accept_counts()
    s.close()
    t_hip = time.time() - t0
    t0 = time.time()
    subprocess.run([oracle, "run", pth, str(iters), theirs, theirs + ".state", str(iters - 1), "0"], check=True, timeout=7200)
    t_oracle = time.time() - t0
    A, B = open(mine).read().splitlines(), open(theirs).read().splitlines()
    assert len(A) == len(B), f"record count differs: {len(A)} vs {len(B)}"
    per_it, checks = {}, 0
    for x, y in zip(A, B):
        xs, ys = x.split(), y.split()
        if xs[0] == "IT":
            assert xs[:4] == ys[:4], f"accept counters differ:\n  {x}\n  {y}"
            it = int(xs[1])
            checks += xs[2] == "CHECK"
            for u, v in zip(xs[4:], ys[4:]):
                u, v = float.fromhex(u), float.fromhex(v)
                assert _close(u, v, 1e-10), f"accumulator differs beyond 1e-10:\n  {x}\n  {y}"
                per_it[it] = max(per_it.get(it, 0.0), abs(u - v) / max(abs(v), 1e-300))
        elif xs[0] != "TRACE":
            assert x == y, f"record differs:\n  {x}\n  {y}"
    compare_states(mine + ".state", theirs + ".state")
    nstate = sum(1 for ln in open(mine + ".state") if ln.startswith("LOCUS "))
    res = {"workload": f"BASELINE configs[3]: {loci} loci x 16 leaves, 9 populations, 4 bands (bench.py's data set)",
           "iterations": iters, "records_compared": len(A), "checkall_records": int(checks),
           "accept_counters": "exact on every record", "accept_counts_total": [int(a) for a in acc],
           "evaluations": int(cnt["evals"]),
           "worst_rel_diff": max(per_it.values()), "worst_rel_diff_per_iteration": [per_it.get(i, 0.0) for i in range(-1, iters)],
           "per_locus_state_compared": nstate, "per_locus_state": "every 50th locus after the last iteration: every per-locus line byte for byte (tests/parity_util.py: STATE_TOL = 0)",
           "wall_s_hip_incl_record_io": t_hip, "wall_s_oracle_one_thread": t_oracle,
           "library_build_id": G.load_library(dims=(pk.n, pk.K, pk.B)).gph_build_id().decode()}
    json.dump(res, open(os.path.join(out_dir, "full_size_100it.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "worst_rel_diff_per_iteration"}))


if __name__ == "__main__":
    main()
