"""GPU parity at BASELINE.json's FULL size (-m gpu): the benchmark workload itself -- configs[3], 100 000 loci x
1 kb, 16 leaves, 9 populations, 4 migration bands, the exact synthetic data `bench.py` times -- through the C ABI.

 (a) directly against the oracle restatement run live on all 100 000 loci for the first iterations (the oracle
     needs seconds per iteration at this size, so the trajectory is short; every proposal type, two checkAll
     resynchronisations): accept counters of every proposal exact, accumulators within 1e-10 relative;
 (b) size-independent properties on a longer trajectory: run-to-run determinism (byte-identical records), the
     reference's own invariant check (`checkAll`, patch.c:2745: incremental statistics, log-likelihoods and
     conditionals against a from-scratch recomputation of every locus) passing at every log period, and
     shard invariance -- the same chain with the loci split over two ranks (two processes on this GPU, exchanging
     only the reduced vectors) takes identical decisions."""
import os
import subprocess
import sys

import pytest

from conftest import REPO
from parity_util import compare_records

pytestmark = pytest.mark.gpu

L_FULL = 100000
CACHE = os.path.join(REPO, "bench_cache")


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import gphocs_amd as G
    G.build()
    return G


def _workload(G, samples_per_log):
    import bench
    pk = bench.build_workload(G, 4, L_FULL, 6.5, 20261002 + 4, CACHE)   # bench.py's default data set
    pk.samplesPerLog = samples_per_log
    return pk


def _run(G, pk, iters, path):
    s = G.Sampler(pk)
    s.set_record_file(path)
    s.initialize()
    for it in range(iters):
        s.iteration(it)
    s.set_record_file(None)
    cnt, acc = s.counters(), s.accept_counts()
    s.close()
    return cnt, acc


def test_full_size_against_live_oracle(G, oracle_cli, tmp_path):
    from gphocs_amd_pkg import synth
    iters = 8
    pk = _workload(G, 4)            # checkAll (and its accumulator resynchronisation) after iterations 3 and 7
    assert pk.L == L_FULL and pk.n == 16 and pk.K == 9 and pk.B == 4
    pth = str(tmp_path / "full.gpk")
    synth.write_pack(pk, pth)
    mine, theirs = str(tmp_path / "hip.rec"), str(tmp_path / "oracle.rec")
    # per-locus state too (genealogy, event chains with ids and lineage counts, statistics, RNG slots) of every 50th
    # locus after the last iteration: 2 000 loci, field by field
    os.environ["GPH_DUMP_STRIDE"] = "50"
    try:
        s = G.Sampler(pk)
        s.set_record_file(mine)
        s.initialize()
        for it in range(iters):
            s.iteration(it)
        s.dump_state(mine + ".state", False)
        s.set_record_file(None)
        cnt = s.counters()
        s.close()
        subprocess.run([oracle_cli, "run", pth, str(iters), theirs, theirs + ".state", str(iters - 1), "0"], check=True,
                       timeout=1500)
    finally:
        os.environ.pop("GPH_DUMP_STRIDE", None)
    worst = compare_records(mine, theirs)
    from parity_util import compare_states
    compare_states(mine + ".state", theirs + ".state")
    assert sum(1 for l in open(mine + ".state") if l.startswith("LOCUS ")) == L_FULL // 50
    assert any(l.startswith(f"IT {iters - 1} CHECK") for l in open(mine).read().splitlines())
    assert cnt["evals"] > 45 * L_FULL * iters
    print(f"full size: {L_FULL} loci x {iters} iterations, {cnt['evals']} evaluations, worst accumulator rel diff "
          f"{worst:.3e}")


WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np, torch, torch.distributed as dist
import gphocs_amd as G, bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
def allreduce(sums, mins):
    if sums.size:
        t = torch.from_numpy(sums); dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if mins.size:
        t = torch.from_numpy(mins); dist.all_reduce(t, op=dist.ReduceOp.MIN)
pk = bench.build_workload(G, 4, %(L)d, 6.5, 20261006, %(cache)r)
pk.samplesPerLog = %(spl)d
s = G.Sampler(pk, device=0, rank=rank, world=world, allreduce=allreduce)
s.set_record_file(%(out)r + ".%%d" %% rank)
s.initialize()
for it in range(%(iters)d):
    s.iteration(it)
s.set_record_file(None)
s.close()
dist.destroy_process_group()
'''


def test_full_size_determinism_checkall_and_shard_invariance(G, tmp_path):
    iters, spl = 24, 6
    pk = _workload(G, spl)
    a, b = str(tmp_path / "a.rec"), str(tmp_path / "b.rec")
    cnt_a, acc_a = _run(G, pk, iters, a)
    cnt_b, acc_b = _run(G, pk, iters, b)
    ra = open(a).read()
    assert ra == open(b).read(), "two runs of the same chain differ"
    assert cnt_a == cnt_b and acc_a == acc_b
    lines = ra.splitlines()
    # checkAll ran (and passed: a failure aborts the iteration with an error status) at every log period
    assert sum(1 for l in lines if " CHECK " in l) == iters // spl
    # every proposal class was exercised and accepted somewhere along the trajectory (0 node ages, 1 migration-event
    # ages, 2 SPR, 3 theta, 4 migration rates, 5 tau, 6 mixing; 7 = migration events proposed on)
    assert all(acc_a[i] > 0 for i in (0, 1, 2, 3, 4, 5, 6, 7)), acc_a
    # the same chain over two ranks (contiguous shards of 50 000 loci, one process each, both on cuda:0)
    out = str(tmp_path / "rk")
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(repo=REPO, L=L_FULL, cache=CACHE, spl=spl, out=out, iters=iters))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    assert open(out + ".0").read() == open(out + ".1").read()
    worst = compare_records(out + ".0", a)
    print(f"full size: {iters} iterations deterministic; two ranks vs one: worst accumulator rel diff {worst:.3e}")


# ------------------------------------------------------------------------------------------------------------
# the OTHER BASELINE configs at their stated sizes (configs[3] = the benchmark workload is covered above):
#   configs[1]  10 000 loci x 1 kb, 4 diploid samples ( 8 leaves), 3-population tree, no migration bands
#   configs[2]  40 000 loci,        6 diploid samples (12 leaves), 3-population tree + 2 migration bands, het integration
#   configs[4] 200 000 loci,       10 diploid samples (20 leaves), 7-population tree + 4 bands, one FIXED ancient sample
#              (mixing off: MCMCcontrol.c:903-907); library variant `l`
FULL = {1: dict(synth=2, L=10000, n=8, K=5, B=0), 2: dict(synth=3, L=40000, n=12, K=5, B=2),
        4: dict(synth=5, L=200000, n=20, K=13, B=4)}

WORKER_SHM = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import gphocs_amd as G, bench
rank, world = int(sys.argv[1]), int(sys.argv[2])
pk = bench.build_workload(G, %(synth)d, %(L)d, 6.5, 20261002 + %(synth)d, %(cache)r)
pk.samplesPerLog = %(spl)d
lib = G.load_library(dims=(pk.n, pk.K, pk.B))
comm = lib.gph_comm_create_shm(%(name)r.encode(), rank, world)
assert comm
s = G.Sampler(pk, lib=lib, device=0, rank=rank, world=world, comm=comm)
s.set_record_file(%(out)r + ".%%d" %% rank)
s.initialize()
for it in range(%(iters)d):
    s.iteration(it)
s.set_record_file(None)
s.close()
lib.gph_comm_destroy(comm)
'''


@pytest.mark.parametrize("idx", [1, 2, 4])
def test_other_configs_at_full_size(G, oracle_cli, tmp_path, idx):
    """(a) the first 4 iterations against the oracle run live on ALL loci (two checkAll resynchronisations): accept
    counters exact, accumulators within 1e-10 relative; (b) run-to-run determinism and checkAll at every log period over
    a longer trajectory; (c) shard invariance: the same chain over two ranks (two processes on this GPU, native
    shared-memory communicator) takes identical decisions"""
    import bench
    from gphocs_amd_pkg import synth
    c = FULL[idx]
    pk = bench.build_workload(G, c["synth"], c["L"], 6.5, 20261002 + c["synth"], CACHE)
    assert (pk.L, pk.n, pk.K, pk.B) == (c["L"], c["n"], c["K"], c["B"])
    # (a)
    pk.samplesPerLog = 2
    pth = str(tmp_path / "full.gpk")
    synth.write_pack(pk, pth)
    mine, theirs = str(tmp_path / "hip.rec"), str(tmp_path / "oracle.rec")
    # per-locus state too (genealogy, event chains with ids and lineage counts, statistics, RNG slots) of every 50th
    # locus after the last iteration, field by field
    os.environ["GPH_DUMP_STRIDE"] = "50"
    try:
        s = G.Sampler(pk)
        s.set_record_file(mine)
        s.initialize()
        for it in range(4):
            s.iteration(it)
        s.dump_state(mine + ".state", False)
        s.set_record_file(None)
        cnt = s.counters()
        s.close()
        subprocess.run([oracle_cli, "run", pth, "4", theirs, theirs + ".state", "3", "0"], check=True, timeout=1800)
    finally:
        os.environ.pop("GPH_DUMP_STRIDE", None)
    os.unlink(pth)
    worst = compare_records(mine, theirs)
    from parity_util import compare_states
    compare_states(mine + ".state", theirs + ".state")
    assert sum(1 for l in open(mine + ".state") if l.startswith("LOCUS ")) == c["L"] // 50
    assert any(l.startswith("IT 3 CHECK") for l in open(mine).read().splitlines())
    # (b)
    iters, spl = 12, 4
    pk.samplesPerLog = spl
    a, b = str(tmp_path / "a.rec"), str(tmp_path / "b.rec")
    cnt_a, acc_a = _run(G, pk, iters, a)
    cnt_b, acc_b = _run(G, pk, iters, b)
    ra = open(a).read()
    assert ra == open(b).read() and cnt_a == cnt_b and acc_a == acc_b
    assert sum(1 for l in ra.splitlines() if " CHECK " in l) == iters // spl
    # (c)
    out = str(tmp_path / "rk")
    script = tmp_path / "w.py"
    script.write_text(WORKER_SHM % dict(repo=REPO, synth=c["synth"], L=c["L"], cache=CACHE, spl=spl, out=out, iters=iters,
                                        name=f"/gphocs-full-{os.getpid()}-{idx}"))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=1800) == 0
    assert open(out + ".0").read() == open(out + ".1").read()
    worst2 = compare_records(out + ".0", a)
    P = __import__("numpy").diff(pk.pattern_offsets)
    print(f"configs[{idx}]: {c['L']} loci x {c['n']} leaves, mean {P.mean():.1f} / max {P.max()} phased patterns, "
          f"{cnt['evals']} evaluations in 4 iterations; vs oracle worst rel diff {worst:.3e}; two ranks vs one {worst2:.3e}")


def test_long_trajectory_keeps_the_reference_invariants(G):
    """400 iterations of the benchmark workload (the chain well past its prior-sampled start: migration events, deeper
    trees) with `checkAll` (patch.c:2745) every 40: every incremental statistic, log-likelihood and conditional array
    of every locus against a from-scratch recomputation, ten times along the way; any per-locus error code (chain-walk
    guards, event-pool exhaustion, the reflect guard) aborts an iteration with an error status.  tools/soak.py runs the
    same for thousands of iterations."""
    pk = _workload(G, 40)
    s = G.Sampler(pk)
    s.initialize()
    for it in range(400):
        s.iteration(it)
    acc = s.accept_counts()
    hs = s.host_stats()
    s.close()
    assert all(a > 0 for a in acc[:8]), acc
    assert hs["resident"] and hs["syncs"] <= 400 + 64     # one per iteration, + initialisation and the checkAll passes


def test_end_to_end_through_the_unchanged_files_at_100k_loci(G):
    """the bench data set as a 100 000-locus sequence file + control file through `G-PhoCS-hip`
    (gph_loci_read -> engine -> trace writer); the trace file against the REAL binary's on the same files
    (tests/golden/e2e100k.trace, generated in the build container by tools/e2e_files.py golden): parameter columns
    character-identical, log-likelihood columns within 1e-10 relative.  Also times gph_loci_read at 20k / 50k / 100k loci
    (gpurun_out/e2e_100k.json; the reference's start-up seconds sit in tests/golden/e2e100k.ref.json)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import e2e_files
    out = e2e_files.run(100000, 24)
    assert out["trace_rows_compared"] == 24 and out["worst_relative_difference_of_a_log_likelihood_column"] <= 1e-10
