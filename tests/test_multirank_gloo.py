"""world_size-2 test of the sharded path on CPU (gloo): two processes, each with an engine over its
contiguous shard of loci, exchanging only the small reduced vectors through the all-reduce hook
(gph_engine_set_allreduce) -- here torch.distributed/gloo, on the GPU box RCCL.  Uses the host
build of the engine sources (tests/hostemu); the result must equal the single-rank run: accept
counters exact, accumulators within 1e-10, and the conflict early-out must pick the same locus."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO
from parity_util import compare_records


def _free_port():
    """a port nobody listens on right now (tests of this file may run side by side under pytest -n)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests", "hostemu"))
import numpy as np, torch, torch.distributed as dist
import gphocs_amd as G, run_hostemu as R
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
def allreduce(sums, mins):
    if sums.size:
        t = torch.from_numpy(sums); dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if mins.size:
        t = torch.from_numpy(mins); dist.all_reduce(t, op=dist.ReduceOp.MIN)
lib = G.load_library(R.build_hostemu())
pk = G.Pack.load(%(pack)r)
if %(presharded)d:
    # bench.py's weak-scaling path: the rank's pack holds ONLY its own loci (global_L / global_begin)
    b, e = pk.shard(rank, world)
    o0, o1 = int(pk.pattern_offsets[b]), int(pk.pattern_offsets[e])
    full_L = pk.L
    pk.pattern_offsets = pk.pattern_offsets[b:e + 1] - o0
    pk.leafcodes, pk.numPhases, pk.counts = pk.leafcodes[o0:o1], pk.numPhases[o0:o1], pk.counts[o0:o1]
    pk.mutRates = pk.mutRates[b:e]
    pk.L = pk.numLoci = e - b
    pk.global_L, pk.global_begin = full_L, b
s = G.Sampler(pk, lib=lib, rank=rank, world=world, allreduce=allreduce)
s.set_record_file(%(out)r + ".%%d" %% rank)
s.initialize()
for it in range(%(iters)d):
    s.iteration(it)
s.set_record_file(None)
print("rank", rank, "loci", s.begin, s.end, "conflicts", s.accept_counts()[8])
s.close()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("name,iters,presharded,world", [("m3", 60, 0, 2), ("g1", 12, 0, 2), ("m3", 40, 1, 2),
                                                         # locus-mut-rate VAR: the serial scan of UpdateLocusRate is chained
                                                         # through the ranks, the reference locus travels from rank 0
                                                         ("v8", 60, 0, 2), ("v9", 40, 1, 2), ("v8", 30, 0, 3)])
def test_two_ranks_equal_one_rank(name, iters, presharded, world, tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import run_hostemu as R
    R.build_hostemu()
    out = str(tmp_path / "rec")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(repo=REPO, pack=os.path.join(GOLDEN, name + ".gpk"), out=out, iters=iters,
                                      presharded=presharded))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    # every rank runs the same host driver and must have written identical records
    for r in range(1, world):
        assert open(out + ".0").read() == open(out + ".%d" % r).read()
    # and they must equal the single-rank run (= the reference golden) up to reduction order
    golden = open(os.path.join(GOLDEN, name + ".rtrace")).read().splitlines()
    mine = open(out + ".0").read().splitlines()
    # golden has more iterations: compare the common prefix line by line
    cut = len(mine)
    (tmp_path / "g").write_text("\n".join(golden[:cut]) + "\n")
    compare_records(out + ".0", str(tmp_path / "g"))


@pytest.mark.parametrize("name", ["m3", "v8"])
def test_program_over_two_ranks(tmp_path, name):
    """the whole program (control file -> trace file) over two ranks: tools/run_multi_gpu.py with gloo and the
    host-emulation build; rank 0's trace file against the real binary's (reduction order differs from the
    single-process run, so values agree to ~1e-13 relative and the printed digits may differ in the last place)"""
    import shutil
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import run_hostemu as R
    lib = R.build_hostemu()
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, name + ext), tmp_path)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    cmd = [sys.executable, os.path.join(REPO, "tools", "run_multi_gpu.py"), name + ".ctl", "--backend", "gloo", "--lib", lib]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=tmp_path) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    want = open(os.path.join(GOLDEN, name + ".trace")).read().splitlines()
    got = open(os.path.join(tmp_path, name + ".trace")).read().splitlines()
    assert want[0] == got[0] and len(want) == len(got)
    for w, g in zip(want[1:], got[1:]):
        if w == g:
            continue
        wf, gf = [float(x) for x in w.split()], [float(x) for x in g.split()]
        assert len(wf) == len(gf) and wf[0] == gf[0]
        assert all(abs(x - y) <= 1.5e-5 * max(1.0, abs(x)) for x, y in zip(wf, gf)), (w, g)
