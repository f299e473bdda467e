"""Decision-level parity: the per-proposal transcript of the real reference compiled with -DLOG_STEPS
(tests/golden/*.logsteps, tests/golden/make_logsteps.py) against the engine's step log -- for two loci of m3 and a7, their
first 200 proposals of UpdateGB_InternalNode / UpdateGB_MigrationNode / UpdateGB_MigSPR: the proposed value, the event
ids considerEventMove works on, lnacceptance to the printed six digits, and the decision of EVERY proposal.  The engine
side is compiled into the test builds only (GPH_LOGSTEPS: the host build here, libgphocs_hip_plain.so on the MI355X)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))

CASES = {"m3": (3, 11), "a7": (2, 9), "j1": (2, 7)}
KEEP = 200
NUM = re.compile(r"-?\d+\.?\d*(?:e[-+]?\d+)?")


def _format(gen, recs):
    """the engine's records of one locus as upstream prints them (GPhoCS.c:2364, 2380, 2387/2400, 2541, 2655; patch.c:1452)"""
    out, cur = [], None
    for r in recs:
        kind = int(r[0])
        if kind == 1:
            cur = "  gen %d, internal node %d, proposing age shift: %g-->%g, " % (gen, int(r[1]), r[2], r[3])
        elif kind == 4:
            cur = "  gen %d, migration node %d, proposing age shift: %g-->%g, " % (gen, int(r[1]), r[2], r[3])
        elif kind == 5:
            cur = "  gen %d, node %d, detaching father %d, pop %d, " % (gen, int(r[1]), int(r[2]), int(r[3]))
        elif kind == 2:
            cur += "considerEventMove: gen %d, event %d, pops %d--> %d, ages %g --> %g. New event %d.\n" % (
                gen, int(r[1]), int(r[2]), int(r[3]), r[4], r[5], int(r[6]))
        elif kind == 3:
            cur += "lnacceptance = %g, %s" % (r[2], "accepting." if r[1] else "rejecting.")
            out.append(cur)
            cur = None
        else:
            raise AssertionError(f"unknown record kind {kind}")
    return out


def _same(a, b):
    """equal text; a number may differ by one unit of its last printed (sixth significant) digit"""
    if a == b:
        return True
    ta, tb = NUM.split(a), NUM.split(b)
    na, nb = NUM.findall(a), NUM.findall(b)
    if ta != tb or len(na) != len(nb):
        return False
    for x, y in zip(na, nb):
        if x != y and abs(float(x) - float(y)) > 2e-6 * max(abs(float(x)), abs(float(y))):
            return False
    return True


def _check(G, lib, name):
    loci = CASES[name]
    want = {}
    cur = None
    for ln in open(os.path.join(GOLDEN, name + ".logsteps")):
        ln = ln.rstrip("\n")
        if ln.startswith("# locus "):
            cur = int(ln.split()[2].rstrip(":"))
            want[cur] = []
        elif ln.startswith("  gen "):
            want[cur].append(ln)
        else:
            want[cur][-1] += "\n" + ln
    pk = G.Pack.load(os.path.join(GOLDEN, name + ".gpk"))
    s = G.Sampler(pk, lib=lib)
    sel = (C.c_int64 * len(loci))(*loci)
    cap = 4096
    assert lib.gph_engine_steplog_enable(s.engine, sel, len(loci), cap) == 0, "this build of the library has no step log"
    s.initialize()
    for it in range(10):
        s.iteration(it)
    ndiff = 0
    for idx, g in enumerate(loci):
        buf = np.zeros((cap, 8))
        n = C.c_int32()
        assert lib.gph_engine_steplog_fetch(s.engine, idx, buf.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n), 0) == 0
        assert 0 < n.value <= cap
        got = _format(g, buf[:n.value])
        assert len(got) >= KEEP, (g, len(got))
        for k in range(KEEP):
            assert _same(got[k], want[g][k]), f"locus {g}, proposal {k}:\n  engine    {got[k]!r}\n  reference {want[g][k]!r}"
            ndiff += got[k] != want[g][k]
    s.close()
    return ndiff


@pytest.mark.parametrize("name", sorted(CASES))
def test_step_log_matches_the_reference_transcript(name):
    import gphocs_amd as G
    import run_hostemu as R
    lib = G.load_library(R.build_hostemu())
    ndiff = _check(G, lib, name)
    assert ndiff <= 8          # printed digits that straddle a rounding boundary


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_step_log_matches_the_reference_transcript_on_the_gpu(name):
    import torch
    assert torch.cuda.is_available()
    import gphocs_amd as G
    G.build()
    lib = G.load_library(os.path.join(REPO, "g-phocs_amd", G.PLAIN_LIB))
    assert _check(G, lib, name) <= 8
