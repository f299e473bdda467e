"""`bench.py --gpus N` starts its N ranks itself (the driver's command shape with no launcher around it).  On the CPU
the ranks run the host-emulation build of the engine sources with the shared-memory exchange and gloo for the bench's
own bookkeeping (--host-emulation, tests only): what is checked is the script's N-rank path -- process start-up,
sharding, the communicator, the ONE relayed JSON line, the failure path -- not a measurement.  The reference's knob
for the same thing is `-n threads` over a fixed numLoci (GPhoCS.c:95, 116-145; MultiCoreUtils.h:8)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")
ARGS = ["--host-emulation", "--loci", "400", "--preroll", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]


def _bench(gpus, extra=(), timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus)] + ARGS + list(extra), capture_output=True, text=True,
                       timeout=timeout, env=env)
    return r


@pytest.fixture(scope="module")
def one_rank():
    r = _bench(1)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_starts_its_own_ranks(one_rank, gpus):
    r = _bench(gpus)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout                       # ONE JSON line, rank 0's
    line = json.loads(lines[0])
    c, c1 = line["config"], one_rank["config"]
    assert line["n_gpus"] == gpus and line["scaling"] == "strong"
    per = -(-c["loci_total"] // gpus)
    assert c["loci_total"] == 400 and c["loci_this_rank"] == per
    assert c["communicator"] == "shm" and c["communicator_world"] == gpus
    assert c["collectives_per_iteration"] >= 5             # sweep, A ancestral populations, mixing, ...
    # one chain whatever the rank count: the same evaluations and the same accept counters as the one-rank run
    assert c["evals_timed"] == c1["evals_timed"]
    assert c["accept_counts_timed"] == c1["accept_counts_timed"]
    assert one_rank["n_gpus"] == 1 and c1["loci_this_rank"] == 400 and c1["collectives_per_iteration"] == 0


def test_bench_fails_as_a_whole_when_a_rank_fails():
    """more ranks than blocks of loci: the rank left without loci exits non-zero, the launcher stops the others (which
    would wait for it in the first exchange) and reports failure -- no line, no hang"""
    r = _bench(3, extra=["--loci", "2"], timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_bench_refuses_more_gpus_than_devices():
    """without --host-emulation the parent counts devices (without initialising one) before it starts anything"""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(max(n, 2)), "--loci", "400", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "HIP device" in r.stderr and r.stdout.strip() == ""


def test_bench_as_ranks_of_torch_distributed_run(one_rank):
    """the driver's other launch form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- the
    script finds RANK / WORLD_SIZE in its environment and is ONE rank (it must not start ranks of its own)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29677", BENCH, "--gpus", "2"] + ARGS
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["loci_this_rank"] == 200
    assert line["config"]["accept_counts_timed"] == one_rank["config"]["accept_counts_timed"]
