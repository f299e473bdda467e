"""The native cross-rank exchange (csrc/gph_comm.cpp) and the `G-PhoCS-hip -g N` launcher, on the CPU: the
host-emulation build of the engine sources + the host shared-memory transport (what ranks that share a GPU use;
RCCL needs one GPU per rank and is covered by the -m gpu tests with a one-rank communicator and by the driver's
multi-GPU runs).  One chain over 2 / 3 ranks must write the trace file the REAL reference binary wrote
(tests/golden/*.trace) -- rubber-band conflicts (m3), estimated sample ages (a7) and UpdateLocusRate's scan
chained through the ranks (v8) included."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO

sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))


@pytest.fixture(scope="module")
def hostemu():
    import run_hostemu
    import gphocs_amd as G
    G.build()                       # the launcher executable (g++) -- the HIP libraries are not loaded here
    return run_hostemu.build_hostemu()


def _same_trace(want_path, got_path):
    want = open(want_path).read().splitlines()
    got = open(got_path).read().splitlines()
    assert want[0] == got[0] and len(want) == len(got)
    ndiff = 0
    for w, g in zip(want[1:], got[1:]):
        if w == g:
            continue
        ndiff += 1
        wf, gf = [float(x) for x in w.split()], [float(x) for x in g.split()]
        assert len(wf) == len(gf) and wf[0] == gf[0]
        assert all(abs(x - y) <= 1.5e-5 * max(1.0, abs(x)) for x, y in zip(wf, gf)), (w, g)
    assert ndiff <= len(want) // 10     # cross-rank sums differ from the serial order in the last printed digit at most


@pytest.mark.parametrize("name,ranks", [("m3", 2), ("m3", 3), ("a7", 2), ("v8", 2), ("v8", 3), ("g1", 4)])
def test_launcher_ranks_write_the_reference_trace(hostemu, tmp_path, name, ranks):
    exe = os.path.join(REPO, "g-phocs_amd", "G-PhoCS-hip")
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, name + ext), tmp_path)
    r = subprocess.run([exe, "-g", str(ranks), name + ".ctl"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GPHOCS_HIP_LIB=hostemu))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _same_trace(os.path.join(GOLDEN, name + ".trace"), os.path.join(tmp_path, name + ".trace"))


def test_launcher_fails_as_a_whole(hostemu, tmp_path):
    """a rank that fails (more ranks than loci here) takes the job down with a non-zero status instead of leaving
    the others waiting in the next exchange"""
    exe = os.path.join(REPO, "g-phocs_amd", "G-PhoCS-hip")
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, "z0" + ext), tmp_path)
    nloci = int(open(os.path.join(GOLDEN, "z0.seq")).read().split()[0])
    ranks = min(nloci * 2 + 1, 40)
    r = subprocess.run([exe, "-g", str(ranks), "z0.ctl"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GPHOCS_HIP_LIB=hostemu))
    if ranks > nloci:
        assert r.returncode != 0


WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import gphocs_amd as G
rank, world = int(sys.argv[1]), int(sys.argv[2])
lib = G.load_library(%(lib)r)
comm = lib.gph_comm_create_shm(%(name)r.encode(), rank, world)
assert comm
s = G.Sampler(G.Pack.load(%(pack)r), lib=lib, rank=rank, world=world, comm=comm)
s.set_record_file(%(out)r + ".%%d" %% rank)
s.initialize()
for it in range(%(iters)d):
    s.iteration(it)
s.set_record_file(None)
hs = s.host_stats()
assert hs["collectives"] > 0 and not hs["resident"]
s.close()
lib.gph_comm_destroy(comm)
'''


@pytest.mark.parametrize("name,iters", [("m3", 60), ("a6", 40)])
def test_sampler_over_native_shm_comm(hostemu, tmp_path, name, iters):
    """two processes, each with the engine over its shard and a gph_comm_create_shm() communicator: every rank
    writes the single-rank golden's records"""
    from parity_util import compare_records
    out = str(tmp_path / "rec")
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(repo=REPO, lib=hostemu, name=f"/gphocs-test-{os.getpid()}-{name}",
                                    pack=os.path.join(GOLDEN, name + ".gpk"), out=out, iters=iters))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    assert open(out + ".0").read() == open(out + ".1").read()
    mine = open(out + ".0").read().splitlines()
    golden = open(os.path.join(GOLDEN, name + ".rtrace")).read().splitlines()[:len(mine)]
    (tmp_path / "g").write_text("\n".join(golden) + "\n")
    compare_records(out + ".0", str(tmp_path / "g"))


@pytest.mark.parametrize("name,ranks", [("m3", 1), ("a7", 2), ("v8", 1), ("x8", 1), ("y9@mid", 1), ("m3@mid", 1), ("j1", 1), ("j2", 2), ("j3@mid", 1),
                                        ("m3@w64", 1), ("j1@w64", 1), ("x8@w64", 1)])     # @w64: the DEVICE forms on the 64-lane micro-wave (csrc/gph_emu64.h)
def test_engine_sources_under_asan_ubsan(hostemu, tmp_path, name, ranks):
    """the engine sources (host build) under AddressSanitizer + UndefinedBehaviorSanitizer, whole program through the
    launcher: no report (either aborts the run) and the reference's trace file.  The GPU pool offers no sanitizer, so
    this is where out-of-bounds indices into the locus image, the chain state and the reduced rows would show.  Round 6: the
    sanitizer builds also carry the index checks of the checked build (-DGPH_BOUNDS: an index that leaves its ARRAY but stays inside
    the image is invisible to AddressSanitizer); the program fails when one fired."""
    import run_hostemu
    mid = name.endswith("@mid")        # the variant-h configuration (64 / 39 / 16: two-word node sets, fused walk): ADVICE round 4
    w64 = name.endswith("@w64")        # round 6: the device forms of lik_compute & co. on 64 emulated lanes, under the sanitizers and the index checks
    name = name.split("@")[0]
    san = run_hostemu.build_hostemu(sanitize=True, mid=mid, wave64=w64)
    rt = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) and os.path.exists(p) for p in rt):
        pytest.skip("sanitizer runtimes are not installed")
    exe = os.path.join(REPO, "g-phocs_amd", "G-PhoCS-hip")
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, name + ext), tmp_path)
    want = os.path.join(GOLDEN, name + ".trace")
    if w64:
        # the micro-wave under the sanitizers runs a tenth of the speed of the plain host build: the first 12 iterations of the chain
        # (the trace file's first 12 rows) keep the CPU suite within minutes
        import re
        ctl = os.path.join(tmp_path, name + ".ctl")
        txt = open(ctl).read()
        txt2 = re.sub(r"(mcmc-iterations\s+)\d+", lambda m: m.group(1) + "12", txt)
        assert re.search(r"mcmc-iterations\s+\d+", txt)
        open(ctl, "w").write(txt2)
        want = os.path.join(tmp_path, "want.trace")
        open(want, "w").write("".join(l + "\n" for l in open(os.path.join(GOLDEN, name + ".trace")).read().splitlines()[:13]))
    env = dict(os.environ, GPHOCS_HIP_LIB=san, LD_PRELOAD=":".join(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    args = [exe] + (["-g", str(ranks)] if ranks > 1 else []) + [name + ".ctl"]
    r = subprocess.run(args, cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    _same_trace(want, os.path.join(tmp_path, name + ".trace"))


def test_shm_exchange_ignores_a_leftover_segment(hostemu, tmp_path):
    """a segment of the same name left behind by a crashed run (right size, right magic, stale arrival counters) must not
    be mistaken for this run's: a rank only settles on a segment whose live rank 0 answers its hello; rank 1 starts
    first here and finds the leftover, rank 0 comes later and replaces it"""
    import time
    import gphocs_amd as G
    from parity_util import compare_records
    lib = G.load_library(hostemu)
    name = f"/gphocs-test-stale-{os.getpid()}"
    stale = lib.gph_comm_create_shm(name.encode(), 0, 1)      # world 1: returns at once; never destroyed = "crashed"
    assert stale and os.path.exists("/dev/shm" + name)
    # poison the leftover's arrival counters the way a run that died mid-exchange would leave them
    with open("/dev/shm" + name, "r+b") as f:
        f.seek(64 + 4 * 128)          # past magic/failed + pad and the hello/ack words: the per-rank sequence slots
        f.write(b"\x07" * 64 * 8)
    out = str(tmp_path / "rec")
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(repo=REPO, lib=hostemu, name=name, pack=os.path.join(GOLDEN, "m3.gpk"), out=out, iters=20))
    p1 = subprocess.Popen([sys.executable, str(script), "1", "2"])
    time.sleep(1.5)
    p0 = subprocess.Popen([sys.executable, str(script), "0", "2"])
    assert p0.wait(timeout=300) == 0 and p1.wait(timeout=300) == 0
    assert open(out + ".0").read() == open(out + ".1").read()
    golden = open(os.path.join(GOLDEN, "m3.rtrace")).read().splitlines()[:len(open(out + ".0").read().splitlines())]
    (tmp_path / "g").write_text("\n".join(golden) + "\n")
    compare_records(out + ".0", str(tmp_path / "g"))


BROKEN = r"""
import sys
sys.path.insert(0, %(repo)r)
import gphocs_amd as G
lib = G.load_library(%(lib)r)
out = []
for pop in range(5):
    s = G.Sampler(G.Pack.load(%(pack)r), lib=lib)
    s.initialize()
    for it in range(5):
        s.iteration(it)
    assert lib.gph_engine_debug_break_chain(s.engine, 11, pop) == 0
    try:
        s.iteration(5)
        out.append((pop, -1, 0))
    except RuntimeError:
        out.append((pop,) + s.last_error())
    s.close()
print("RESULT", out)
"""


def test_broken_chains_fail_cleanly_and_alike_in_both_walk_forms(hostemu, tmp_path):
    """the fused pruning + sampling walk (trace_pair) finishes the step that found an inconsistency and
    reports it behind its loop; the two-walk form (GPH_TWO_WALKS) returns at once.  A broken event chain in each of five
    populations of one locus must give the SAME first failing locus and reference-style code from both forms (8: the age
    check at a population boundary, 92: a walk off the end of a chain), the run must end with GPH_EKERNEL -- and the
    failure path must not touch memory it does not own: the fused form runs under AddressSanitizer + UBSan here (a failed
    rubberBandRipple used to be undone through event id -1)."""
    import run_hostemu
    rt = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    have_san = all(os.path.isabs(p) and os.path.exists(p) for p in rt)
    res = {}
    for tag, lib, san in (("fused", run_hostemu.build_hostemu(sanitize=have_san), have_san), ("two", run_hostemu.build_hostemu(two_walks=True), False)):
        script = tmp_path / f"b_{tag}.py"
        script.write_text(BROKEN % dict(repo=REPO, lib=lib, pack=os.path.join(GOLDEN, "m3.gpk")))
        env = dict(os.environ)
        if san:
            env.update(LD_PRELOAD=":".join(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        res[tag] = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    assert res["fused"] == res["two"], res
    got = eval(res["fused"].split(" ", 1)[1])
    assert all(locus == 11 and code in (6, 8, 91, 92, 96) for _, locus, code in got), got


ERR_WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import gphocs_amd as G
rank, world = int(sys.argv[1]), int(sys.argv[2])
lib = G.load_library(%(lib)r)
comm = lib.gph_comm_create_shm(%(name)r.encode(), rank, world)
assert comm
s = G.Sampler(G.Pack.load(%(pack)r), lib=lib, rank=rank, world=world, comm=comm)
s.initialize()
for it in range(4):
    s.iteration(it)
victim = %(victim)d
rc = lib.gph_engine_debug_break_chain(s.engine, victim, 1)      # every rank calls it (it completes a deferred collective pass first)
assert (rc == 0) == (s.begin <= victim < s.end), rc     # ... and the rank that holds the locus breaks one of its chains
try:
    s.iteration(4)
    print("RESULT", rank, "no error")
except RuntimeError:
    print("RESULT", rank, s.last_error())
'''


def test_fatal_error_of_one_shard_names_the_locus_on_every_rank(hostemu, tmp_path):
    """the first failing locus and its code ride in the reduced row as ONE column combined by maximum over the ranks
    ((2^30 - locus) 2^14 + code): a broken chain in rank 1's shard ends the iteration on BOTH ranks with GPH_EKERNEL, and
    gph_engine_last_error names the same global locus and code on both"""
    script = tmp_path / "w.py"
    script.write_text(ERR_WORKER % dict(repo=REPO, lib=hostemu, name=f"/gphocs-test-err-{os.getpid()}",
                                        pack=os.path.join(GOLDEN, "m3.gpk"), victim=13))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    res = sorted(ln for o, _ in outs for ln in o.splitlines() if ln.startswith("RESULT"))
    assert len(res) == 2, outs
    r0, r1 = (eval(ln.split(" ", 2)[2]) for ln in res)
    assert r0 == r1 and r0[0] == 13 and 0 < r0[1] < 10000, res
    # only the rank that holds the locus can print its genealogy
    assert sum("genealogy and event chains of locus 13" in e for _, e in outs) == 1
