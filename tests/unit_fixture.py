"""Kernel-level fixtures: tests/golden/<case>.unit holds the outputs of SINGLE calls of
the real reference's per-locus functions (oracle/ref_harness.c `unit`) after a fixed number of iterations; the engine
replays the same calls through gph_engine_unit.  Bit-exact: the device arithmetic is operation-for-operation the
reference's, and every call is undone, so one differing line names the function, the locus and the node."""
import os

import numpy as np

UNIT_CASES = {"m4": 60, "a7": 40, "g2": 20}     # case -> iterations before the calls (tests/golden/make_goldens.sh)


def load_unit(path):
    A, B, C = {}, {}, {}
    for l in open(path):
        t = l.split()
        if t[0] == "A":
            A[(int(t[1]), int(t[2]))] = tuple(float.fromhex(x) for x in t[3:6])
        elif t[0] == "B":
            B[int(t[1])] = float.fromhex(t[2])
        elif t[0] == "C":
            C[(int(t[1]), int(t[2]))] = (float.fromhex(t[3]), int(t[4]), int(t[5]), float.fromhex(t[6]))
    return A, B, C


def check_unit(G, lib, golden_dir, name, exact=True):
    pk = G.Pack.load(os.path.join(golden_dir, name + ".gpk"))
    s = G.Sampler(pk, lib=lib)
    s.initialize()
    for it in range(UNIT_CASES[name]):
        s.iteration(it)
    A, B, C = load_unit(os.path.join(golden_dir, name + ".unit"))
    n = pk.n

    def same(x, y, what):
        if exact:
            assert x == y, (what, x.hex() if isinstance(x, float) else x, y.hex() if isinstance(y, float) else y)
        else:
            assert x == y or abs(x - y) <= 1e-9 * max(abs(x), abs(y)), (what, x, y)
    a = s.unit(0)
    for (g, inode), (tnew, lnld, dprior) in A.items():
        row = a[g, 3 * (inode - n):3 * (inode - n) + 3]
        same(float(row[0]), tnew, ("A tnew", g, inode))
        same(float(row[1]), lnld, ("A lnLd", g, inode))
        same(float(row[2]), dprior, ("A dprior", g, inode))
    for ap in range(pk.Kc, pk.K):
        c = s.unit(2, ap)
        for g in range(pk.L):
            d, n0, n1, lik = C[(g, ap)]
            same(float(c[g, 0]), d, ("C delta", g, ap))
            assert (int(c[g, 1]), int(c[g, 2])) == (n0, n1), ("C counts", g, ap)
            same(float(c[g, 3]), lik, ("C lik", g, ap))
    b = s.unit(1)
    for g, v in B.items():
        same(float(b[g, 0]), v, ("B", g))
    # the calls were undone: the chain continues as if nothing had happened
    s.iteration(UNIT_CASES[name])
    s.close()
    return len(A) + len(B) + len(C)
