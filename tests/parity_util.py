"""Comparison helpers for the parity tests.

Bar (BASELINE.json north_star): accept/reject counters bit-exact, fp64 log-likelihoods within
1e-10 relative.  The HIP path keeps the reference's operand order per locus (and its libm's exp/log
bit for bit), so everything PER LOCUS is compared byte for byte; only sums over loci -- reduced with a
fixed-shape tree on the device, serially upstream -- get REL_TOL.  Every integer (counters, topology,
event ids, lineage counts, RNG state) must be identical."""
import math

REL_TOL = 1e-10      # cross-locus sums: log-likelihoods / accumulators (north_star tolerance; the summation tree differs)
STATE_TOL = 0.0      # per-locus doubles (ages, elapsed times, statistics, conditionals, per-locus lnL): byte-equal


def _close(a, b, tol):
    if a == b:
        return True
    return abs(a - b) <= tol * max(abs(a), abs(b), 1e-300) or abs(a - b) < 1e-300


def compare_records(path_a, path_b, tol=REL_TOL):
    """IT / CONFLICTS / TRACE lines: proposal name + accept count exact, accumulators within tol"""
    A = open(path_a).read().splitlines()
    B = open(path_b).read().splitlines()
    assert len(A) == len(B), f"record count differs: {len(A)} vs {len(B)}"
    worst = 0.0
    for x, y in zip(A, B):
        xs, ys = x.split(), y.split()
        if xs[0] == "IT":
            assert xs[:4] == ys[:4], f"accept counters differ:\n  {x}\n  {y}"
            for u, v in zip(xs[4:], ys[4:]):
                u, v = float.fromhex(u), float.fromhex(v)
                assert _close(u, v, tol), f"accumulator differs beyond {tol}:\n  {x}\n  {y}"
                worst = max(worst, abs(u - v) / max(abs(v), 1e-300))
        elif xs[0] == "TRACE":
            assert len(xs) == len(ys)
            for u, v in zip(xs[1:], ys[1:]):
                assert abs(float(u) - float(v)) <= 2e-5 * max(1.0, abs(float(v)) * 1e-3), f"trace row differs:\n  {x}\n  {y}"
        else:
            assert x == y, f"record differs:\n  {x}\n  {y}"
    return worst


def compare_trace_files(want_path, got_path, lnl_tol=REL_TOL):
    """the trace file of the real binary (GPhoCS.c:1763-1769) against the program's: header and row count equal; every
    PARAMETER column (thetas, taus, migration rates, sample ages, rate variance: printParamVals, %8.5f of per-chain values that
    follow from exact accept decisions) character-identical; only the two log-likelihood columns at the end of a row (sums
    over loci, %.6f) within `lnl_tol` relative, plus one unit of their last printed digit.  Returns the number of rows whose
    log-likelihood text differs."""
    want = open(want_path).read().splitlines()
    got = open(got_path).read().splitlines()
    assert want[0] == got[0], "trace header differs"
    assert len(want) == len(got), f"trace rows: {len(want)} vs {len(got)}"
    ndiff = 0
    for w, g in zip(want[1:], got[1:]):
        if w == g:
            continue
        ndiff += 1
        wt, gt = w.split("\t"), g.split("\t")
        assert len(wt) == len(gt), (w, g)
        assert wt[:-2] == gt[:-2], f"a parameter column differs:\n  {w}\n  {g}"
        for a, b in zip(wt[-2:], gt[-2:]):
            x, y = float(a), float(b)
            assert abs(x - y) <= lnl_tol * abs(x) + 1.000001e-6, f"log-likelihood column differs beyond {lnl_tol}:\n  {w}\n  {g}"
    return ndiff


def _tok_close(u, v, tol):
    """compare one whitespace token that may be int, hexfloat or colon-joined mixture"""
    if u == v:
        return True
    if ":" in u or ":" in v:
        us, vs = u.split(":"), v.split(":")
        return len(us) == len(vs) and all(_tok_close(a, b, tol) for a, b in zip(us, vs))
    if "x" in u or "x" in v or "." in u or "." in v:
        try:
            return _close(float.fromhex(u), float.fromhex(v), tol)
        except ValueError:
            return False
    return False  # integers must match exactly


def _ulps(u, v):
    """distance of two hex-float tokens in units of the last place of the larger one (None: not both floats)"""
    try:
        a, b = float.fromhex(u), float.fromhex(v)
    except ValueError:
        return None
    if a == b:
        return 0.0
    m = max(abs(a), abs(b))
    return abs(a - b) / math.ulp(m) if m > 0 else 0.0


# Lines of a canonical state dump that sum over loci (fixed-shape tree on the device, serial order upstream): the only
# ones compared with a tolerance.  Everything else -- MODEL, and per locus LOCUS / N / C / K / M / S: ages, elapsed times,
# statistics, log-likelihoods, conditionals, RNG slots, event ids, lineage counts -- must be BYTE-EQUAL (DESIGN section 5:
# the reference's operand order per locus, LocusDataLikelihood.c:471-479, patch.c:1730, 1671).
CROSS_LOCUS_LINES = ("GLOBAL", "TOTALS")


def compare_states(path_a, path_b, tol=REL_TOL, skip_global=False, per_locus_tol=0.0):
    """canonical state dumps: per-locus lines byte for byte (per_locus_tol = 0), the cross-locus sums of the GLOBAL / TOTALS
    lines within `tol` (their integers exact).  A failure lists EVERY differing field with its distance in ulps."""
    A = open(path_a).read().splitlines()
    B = open(path_b).read().splitlines()
    assert len(A) == len(B), f"state line count differs: {len(A)} vs {len(B)}"
    bad, locus = [], "-"
    for x, y in zip(A, B):
        if x.startswith("LOCUS "):
            locus = x.split()[1]
        if x == y:
            continue
        xs, ys = x.split(), y.split()
        assert len(xs) == len(ys), f"state line differs:\n  {x[:200]}\n  {y[:200]}"
        if xs[0] in CROSS_LOCUS_LINES:
            if skip_global:
                continue
            for u, v in zip(xs, ys):
                assert _tok_close(u, v, tol), f"state differs ({u} vs {v}):\n  {x[:300]}\n  {y[:300]}"
            continue
        for col, (u, v) in enumerate(zip(xs, ys)):
            if u == v:
                continue
            if per_locus_tol > 0 and _tok_close(u, v, per_locus_tol):
                continue
            for a, b in zip(u.split(":"), v.split(":")):
                if a != b:
                    d = _ulps(a, b)
                    bad.append(f"locus {locus} line {xs[0]} {xs[1] if len(xs) > 1 else ''} field {col}: {a} vs {b}"
                               + (f" ({d:.3g} ulp)" if d is not None else " (integer)"))
    assert not bad, (f"{len(bad)} per-locus field(s) not byte-equal ({path_a} vs {path_b}):\n  " + "\n  ".join(bad[:40]))
    return True


def reinit_after_accepted_mixing(G, lib, pack_path, tmp_path, max_iters=200, more=6):
    """an accepted mixing proposal leaves its commit for the head of the NEXT sweep kernel (the evaluated
    state waits in the shadow page).  Re-initialising the genealogies in that window must drop the owed commit: the next
    sweep would otherwise stage the OLD chain's shadow page over the fresh genealogies.  Engine A runs until a mixing
    proposal has just been accepted and re-initialises; engine B is fresh, takes A's chain state above the loci, and
    initialises once: per-locus state and records of both must be identical after `more` iterations."""
    import ctypes as C

    class Chain(C.Structure):
        _fields_ = [("theta", C.c_double * 40), ("popAge", C.c_double * 40), ("sampleAge", C.c_double * 40),
                    ("migRate", C.c_double * 100), ("bandStart", C.c_double * 100), ("bandEnd", C.c_double * 100),
                    ("rng", C.c_uint32 * 3), ("logLikelihood", C.c_double), ("dataLogLikelihood", C.c_double),
                    ("rateVar", C.c_double), ("coal_stats", C.c_double * 40), ("num_coals", C.c_double * 40),
                    ("mig_stats", C.c_double * 100), ("num_migs", C.c_double * 100), ("rubberband_mig_conflicts", C.c_int64)]
    pk = G.Pack.load(pack_path)
    a = G.Sampler(pk, lib=lib)
    a.initialize()
    it, got = 0, False
    while it < max_iters:
        before = a.accept_counts()[6]
        a.iteration(it)
        it += 1
        if a.accept_counts()[6] > before and (it + 1) % int(pk.samplesPerLog) != 0 and it != int(pk.startMig):
            got = True
            break
    assert got, "no accepted mixing proposal within the iteration budget"
    ch = Chain()
    assert lib.gph_mcmc_get_chain(a.mcmc, C.byref(ch)) == 0
    b = G.Sampler(pk, lib=lib)
    b.initialize()
    assert lib.gph_mcmc_set_chain(b.mcmc, C.byref(ch)) == 0
    assert lib.gph_mcmc_set_chain(a.mcmc, C.byref(ch)) == 0
    ta, tb = str(tmp_path / "a.trace"), str(tmp_path / "b.trace")
    a.set_record_file(ta)
    b.set_record_file(tb)
    assert lib.gph_mcmc_initialize_genealogies(a.mcmc) == 0
    assert lib.gph_mcmc_initialize_genealogies(b.mcmc) == 0
    for k in range(more):
        a.iteration(it + k)
        b.iteration(it + k)
    sa, sb = str(tmp_path / "a.state"), str(tmp_path / "b.state")
    a.dump_state(sa, True)
    b.dump_state(sb, True)
    a.set_record_file(None)
    b.set_record_file(None)
    a.close()
    b.close()
    assert open(sa, "rb").read() == open(sb, "rb").read(), "per-locus state differs after re-initialisation"
    ra, rb = open(ta).read(), open(tb).read()
    assert ra == rb and len(ra.splitlines()) > more, "records differ after re-initialisation"
