#!/usr/bin/env python3
"""How dependable is the in-kernel exchange between thread ranks (GPH_PEER_EXCHANGE=1)?  N fresh processes per world size; every
process first creates and destroys a few dozen streams (what a long test session leaves behind: the round-robin position of
HIP's hardware-queue pool), then runs `iters` iterations of golden m3 as `world` thread ranks with the exchange on and compares
the records with the single-rank run.  A run whose bounded wait gives up (Fatal Error 9997: two ranks' streams behind each other
on one hardware queue) counts as a failure.  Since round 6 rank r's stream has priority level r mod 3 -- HIP keeps a queue pool
per level -- so worlds up to 3 cannot share a queue.

    python3 tools/peer_exchange_stress.py [N=10] [iters=40]   ->   gpurun_out/peer_stress.json"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys, threading
sys.path.insert(0, %(repo)r)
import torch
for _ in range(%(churn)d):
    ss = [torch.cuda.Stream() for _ in range(3)]
    del ss
import gphocs_amd as G
pk = G.Pack.load(%(pack)r)
lib = G.load_library(dims=(pk.n, pk.K, pk.B))
iters, out, world = %(iters)d, %(out)r, %(world)d
s = G.Sampler(pk, lib=lib)
s.set_record_file(out + ".one"); s.initialize()
for it in range(iters): s.iteration(it)
s.set_record_file(None); s.close()
os.environ["GPH_PEER_EXCHANGE"] = "1"
group = lib.gph_comm_local_group(world, 0)
comms = [lib.gph_comm_create_local(group, r) for r in range(world)]
errs = []
def work(r):
    try:
        s = G.Sampler(pk, lib=lib, rank=r, world=world, comm=comms[r])
        s.set_record_file(out + ".x.%%d" %% r); s.initialize()
        for it in range(iters): s.iteration(it)
        s.set_record_file(None); s.close()
    except Exception as ex:
        errs.append((r, str(ex))); lib.gph_comm_destroy(comms[r]); comms[r] = None
th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join(timeout=300) for t in th]
[lib.gph_comm_destroy(c) for c in comms if c]
assert not errs, errs
sys.path.insert(0, os.path.join(%(repo)r, "tests"))
from parity_util import compare_records
for r in range(world):
    compare_records(out + ".x.%%d" %% r, out + ".one")      # accept counters exact, sums over loci (rank-order combine) <= 1e-10
    assert open(out + ".x.%%d" %% r).read() == open(out + ".x.0").read(), "ranks disagree"
print("OK")
"""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    pack = os.path.join(REPO, "tests", "golden", "m3.gpk")
    res = {}
    for world in (2, 3, 4):
        ok = fail9997 = other = 0
        for k in range(n):
            with tempfile.TemporaryDirectory() as td:
                sc = os.path.join(td, "w.py")
                open(sc, "w").write(WORKER % dict(repo=REPO, pack=pack, iters=iters, out=os.path.join(td, "rec"), world=world, churn=7 * k))
                r = subprocess.run([sys.executable, sc], capture_output=True, text=True, timeout=900)
                if r.returncode == 0 and "OK" in r.stdout:
                    ok += 1
                elif "9997" in r.stderr:
                    fail9997 += 1
                else:
                    other += 1
                    print(r.stderr[-1500:], file=sys.stderr)
        res[f"world{world}"] = dict(runs=n, ok=ok, wait_gave_up_9997=fail9997, other_failures=other)
        print(world, res[f"world{world}"], flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(dict(what="in-kernel exchange between thread ranks, fresh process per run, stream churn 0..7(N-1) before the run; "
                        "rank r's stream at priority level r mod 3", iters=iters, results=res),
              open(os.path.join(REPO, "gpurun_out", "peer_stress.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
