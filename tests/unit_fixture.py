"""Kernel-level fixtures: tests/golden/<case>.unit holds the outputs of SINGLE calls of
the real reference's per-locus functions (oracle/ref_harness.c `unit`) after a fixed number of iterations; the engine
replays the same calls through gph_engine_unit.  Bit-exact: the device arithmetic is operation-for-operation the
reference's, and every call is undone, so one differing line names the function, the locus and the node."""
import os

import numpy as np

UNIT_CASES = {"m4": 60, "a7": 40, "g2": 20, "j1": 70, "j2": 50}     # case -> iterations before the calls (tests/golden/make_goldens.sh)


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


def load_unit2(path):
    D, E, F, Gt, H = {}, {}, {}, {}, {}
    for l in open(path):
        t = l.split()
        if t[0] == "D":
            D.setdefault((int(t[1]), int(t[2])), []).append((int(t[3]), float.fromhex(t[4]), int(t[5]), float.fromhex(t[6]), int(t[7])))
        elif t[0] == "E":
            E[(int(t[1]), int(t[2]))] = (float.fromhex(t[3]), float.fromhex(t[4]))
        elif t[0] == "F":
            F[int(t[1])] = (int(t[2]), float.fromhex(t[3]), float.fromhex(t[4]))
        elif t[0] == "H":
            H[int(t[1])] = (int(t[2]), float.fromhex(t[3]), float.fromhex(t[4]))
        elif t[0] == "G":
            Gt[(int(t[1]), int(t[2]))] = (int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), float.fromhex(t[8]),
                                          float.fromhex(t[9]), float.fromhex(t[10]), float.fromhex(t[11]), int(t[12]), int(t[13]), int(t[14]))
    return D, E, F, Gt, H


def check_unit2(G, lib, golden_dir, name):
    """second set (tests/golden/<case>.unit2, oracle/ref_harness.c `unit2`): executeGenSPR with every return code,
    scaleAllNodeAges + revert, rubberBandRipple do / undo, traceLineage outcomes -- every value bit for bit"""
    pk = G.Pack.load(os.path.join(golden_dir, name + ".gpk"))
    s = G.Sampler(pk, lib=lib)
    s.initialize()
    for it in range(UNIT_CASES[name]):
        s.iteration(it)
    D, E, F, Gt, H = load_unit2(os.path.join(golden_dir, name + ".unit2"))
    N = 2 * pk.n - 1
    codes = set()
    for node in range(N):
        d = s.unit(3, node)
        for g in range(pk.L):
            want = D.get((g, node), [])
            assert int(d[g, 0]) == len(want), ("D calls", g, node, int(d[g, 0]), len(want))
            for k, (target, age, ret, lnl, root) in enumerate(want):
                r = d[g, 1 + 5 * k:6 + 5 * k]
                assert (int(r[0]), int(r[2]), int(r[4])) == (target, ret, root), ("D target / code / root", g, node, k, r, want[k])
                assert float(r[1]) == age and float(r[3]) == lnl, ("D age / value", g, node, target, float(r[1]).hex(), age.hex(), float(r[3]).hex(), lnl.hex())
                codes.add(ret)
    assert codes == {r for v in D.values() for (_, _, r, _, _) in v} and codes >= {0, 1}, codes      # (m4 and a7 hold code 2: a subtree pruned from below the root)
    for arg in sorted({a for (_, a) in E}):
        e = s.unit(4, arg)
        for g in range(pk.L):
            assert (float(e[g, 0]), float(e[g, 1])) == E[(g, arg)], ("E", g, arg, float(e[g, 0]).hex(), float(e[g, 1]).hex(), E[(g, arg)])
    f = s.unit(5)
    for g in range(pk.L):
        assert (int(f[g, 0]), float(f[g, 1]), float(f[g, 2])) == F[g], ("F", g, f[g, :3], F[g])
    if H:      # rubberBandRipple over band START / END events (start_or_end == 1): fixtures made from round 6 on
        h = s.unit(7)
        for g in range(pk.L):
            assert (int(h[g, 0]), float(h[g, 1]), float(h[g, 2])) == H[g], ("H", g, h[g, :3], H[g])
    for node in range(N):
        t = s.unit(6, node)
        for g in range(pk.L):
            if (g, node) not in Gt:
                assert int(t[g, 0]) == 0, ("G root", g, node)
                continue
            w = Gt[(g, node)]
            got = (int(t[g, 1]), int(t[g, 2]), int(t[g, 3]), int(t[g, 4]), int(t[g, 5]), float(t[g, 6]), float(t[g, 7]), float(t[g, 8]),
                   float(t[g, 9]), int(t[g, 10]), int(t[g, 11]), int(t[g, 12]))
            assert int(t[g, 0]) == 1 and got == w, ("G", g, node, got, w)
    # nothing of it was written back: the chain continues as if nothing had happened
    s.iteration(UNIT_CASES[name])
    s.close()
    return sum(len(v) for v in D.values()) + len(E) + len(F) + len(Gt) + len(H)
