"""Input front end (g-phocs_amd/csrc/gph_input.cpp, gph_program.cpp): the reference's control-file,
sequence-file and trace-file formats, unchanged.

  * gph_control_read + gph_loci_read against the processed-locus packs the REAL reference produced
    from the same files (tests/golden/*.gpk, written by oracle/_ref/gphocs_ref pack): model,
    priors, finetunes, print factors and -- bit for bit -- every locus's phased pattern table
    (pattern order, phase order, counts).  stress.* exercises 2-/3-way IUPAC codes, haploids,
    missing samples, unknown samples, all-N columns, lower case (tests/golden/make_stress.py).
  * gph_run_control_file (the G-PhoCS main() equivalent) in the host-emulation build against the
    trace FILES the real reference binary wrote for the same control files (tests/golden/*.trace).
  * the reference's rejection behaviour for malformed sequence files.
These run on the CPU: the front end is host code; the chain itself runs in the hostemu build here
and on the MI355X in test_gpu_parity.py::test_program_trace_file.
"""
import os
import shutil
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
import gphocs_amd as G  # noqa: E402
import run_hostemu as R  # noqa: E402

CASES = ["g1", "g2", "m3", "m4", "c5", "s3", "a6", "a7", "z0", "stress", "v8", "v9", "y9", "r5", "b2", "n7", "j1", "j2", "j3"]


@pytest.fixture(scope="module")
def lib():
    return G.load_library(R.build_hostemu())


def _in_dir(path):
    class _Cd:
        def __enter__(self):
            self.old = os.getcwd()
            os.chdir(path)

        def __exit__(self, *a):
            os.chdir(self.old)
    return _Cd()


# name -> secondary control file (the optional second command-line argument of G-PhoCS, GPhoCS.c:35-43, 154-164;
# readSecondaryControlFile, MCMCcontrol.c:178-210: its GENERAL-INFO keys override, its MIG-BANDS module replaces)
SECONDARY = {"w2": "w2b.ctl"}


@pytest.mark.parametrize("name", CASES + ["w2"])
def test_control_and_sequences_match_reference_pack(lib, name):
    ref = G.Pack.load(os.path.join(GOLDEN, name + ".gpk"))
    with _in_dir(GOLDEN):   # seq-file names in the control files are relative
        got = G.Pack.from_control(name + ".ctl", lib=lib, threads=3, secondary=SECONDARY.get(name))
    for f in ("n", "Kc", "K", "B", "rootPop", "L", "seed", "startMig", "doMixing", "samplesPerLog", "mutRateMode",
              "numParameters", "burnin", "sampleSkip", "ftCoalTime", "ftMigTime", "ftTheta", "ftMigRate",
              "ftMixing", "popName") + (("varRatesAlpha", "ftLocusRate") if ref.mutRateMode == 1 else ()):
        assert getattr(ref, f) == getattr(got, f), f
    for f in ("samplesPerPop", "popFather", "popSon0", "popSon1", "sampleAge", "updateSampleAge", "thetaAlpha",
              "thetaBeta", "thetaStart", "ageAlpha", "ageBeta", "ageStart", "ftTaus", "printFactors", "mutRates"):
        assert np.array_equal(np.asarray(getattr(ref, f)), np.asarray(getattr(got, f))), f   # bit-exact
    for f in ("bandSrc", "bandTgt", "mrAlpha", "mrBeta"):
        assert np.array_equal(np.asarray(getattr(ref, f))[:ref.B], np.asarray(getattr(got, f))[:ref.B]), f
    # the phased pattern tables: integer/byte data, bit-exact, order included
    for f in ("pattern_offsets", "leafcodes", "numPhases", "counts"):
        a, b = np.asarray(getattr(ref, f)), np.asarray(getattr(got, f))
        assert a.shape == b.shape and np.array_equal(a, b), f


def test_thread_count_does_not_change_the_result(lib):
    with _in_dir(GOLDEN):
        a = G.Pack.from_control("stress.ctl", lib=lib, threads=1)
        b = G.Pack.from_control("stress.ctl", lib=lib, threads=8)
    for f in ("pattern_offsets", "leafcodes", "numPhases", "counts"):
        assert np.array_equal(getattr(a, f), getattr(b, f))


@pytest.mark.parametrize("name", ["g1", "m3", "a7", "f3", "v8", "w2", "r5", "j1", "j2", "j3"])
def test_program_writes_the_reference_trace_file(lib, name, tmp_path):
    """same control file + sequence file -> the trace file of the real G-PhoCS binary, byte for byte
    (f3: find-finetunes TRUE -- the step-size search of performMCMC, GPhoCS.c:1896-2180, incl. its acceptance
    bookkeeping quirks, must take the same decisions for the chain to stay on the reference's trajectory)"""
    for ext in (".ctl", ".seq", ".rates"):
        if os.path.exists(os.path.join(GOLDEN, name + ext)):
            shutil.copy(os.path.join(GOLDEN, name + ext), tmp_path)
    ctl2 = SECONDARY.get(name)
    if ctl2:
        shutil.copy(os.path.join(GOLDEN, ctl2), tmp_path)
    with _in_dir(tmp_path):
        assert lib.gph_run_control_file((name + ".ctl").encode(), ctl2.encode() if ctl2 else None, 0, 0) == 0
    # parameter columns character-identical, the log-likelihood columns within 1e-10 relative (host build: serial sums, in
    # practice the same text)
    from parity_util import compare_trace_files
    compare_trace_files(os.path.join(GOLDEN, name + ".trace"), os.path.join(tmp_path, name + ".trace"))


def test_phase_count_beyond_16_bits(tmp_path):
    """round 6: golden p6 -- a REPEATED column of 16 heterozygotes among 18 diploids (no symmetry break
    at count > 1, AlignmentProcessor.c:1767) = 2^16 phases of one pattern, 65 554 phased patterns in one locus.  Refused until
    round 5 ("the engine stores phase counts in 16 bits"); now the word carries 0x8000 | exponent (GPH_NUMPHASES).  The front end's
    table (phase counts decoded), and the program's trace file against the real binary's, character for character -- host build of
    the 64-leaf configuration (the block and the conditional arrays of that locus are far beyond any LDS)"""
    from parity_util import compare_trace_files
    lib64 = G.load_library(R.build_hostemu(mid=True))
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, "p6" + ext), tmp_path)
    pk = G.Pack.from_control(os.path.join(tmp_path, "p6.ctl"), lib=lib64, seq_path=os.path.join(tmp_path, "p6.seq"))
    P = np.diff(pk.pattern_offsets)
    assert list(P) == [65554, 17]
    ph = G.decode_phases(pk.numPhases)
    assert int(ph.max()) == 65536 and int(pk.numPhases.max()) == (0x8000 | 16) and int(ph.sum()) == int(P.sum())
    assert np.array_equal(G.encode_phases(ph), pk.numPhases)
    with _in_dir(tmp_path):
        assert lib64.gph_run_control_file(b"p6.ctl", None, 0, 0) == 0
    assert compare_trace_files(os.path.join(GOLDEN, "p6.trace"), os.path.join(tmp_path, "p6.trace")) == 0


def _seq_error(lib, tmp_path, mutate):
    for ext in (".ctl", ".seq"):
        shutil.copy(os.path.join(GOLDEN, "g1" + ext), tmp_path)
    p = os.path.join(tmp_path, "g1.seq")
    text = mutate(open(p).read())
    open(p, "w").write(text)
    with _in_dir(tmp_path):
        with pytest.raises(ValueError) as e:
            G.Pack.from_control("g1.ctl", lib=lib)
    return str(e.value)


def test_rejects_what_the_reference_rejects(lib, tmp_path):
    # blank first line: "Unexpected End of File when trying to read number of loci" (AlignmentProcessor.c:514-524)
    assert "number of loci" in _seq_error(lib, tmp_path, lambda s: "\n" + s)
    # '-' and '?' are not in the accepted alphabet TCAGYWKMSRVDBHN (AlignmentProcessor.c:61, 1467)
    def bad_base(s):
        lines = s.split("\n")
        name, seq = lines[3].split()
        lines[3] = name + "\t" + seq[:5] + "-" + seq[6:]
        return "\n".join(lines)
    assert "Illegal base type '-'" in _seq_error(lib, tmp_path, bad_base)
    # short sequence
    def short(s):
        lines = s.split("\n")
        name, seq = lines[3].split()
        lines[3] = name + "\t" + seq[:-3]
        return "\n".join(lines)
    assert "contained only" in _seq_error(lib, tmp_path, short)
    # fewer loci than announced
    assert "only contains" in _seq_error(lib, tmp_path, lambda s: s[:s.index("locus8 ")])
    # a sample of the control file that never occurs
    assert "no samples for this name" in _seq_error(lib, tmp_path, lambda s: s.replace("s0\t", "zz\t").replace("s0 ", "zz "))


def test_control_file_errors(lib, tmp_path):
    txt = open(os.path.join(GOLDEN, "m3.ctl")).read()
    p = os.path.join(tmp_path, "bad.ctl")
    open(p, "w").write(txt.replace("finetune-theta", "finetune-thetaX"))
    with pytest.raises(ValueError):
        G.Pack.from_control(p, lib=lib)
    # a band whose source is an ancestor of its target is refused (MCMCcontrol.c:1229-1237)
    open(p, "w").write(txt.replace("source  A", "source  AB", 1))
    with pytest.raises(ValueError):
        G.Pack.from_control(p, lib=lib)


# ---------------------------------------------------------------------------- readTrace
READTRACE_CASES = [("g1", [], "all"), ("g1", ["-b", "3"], "b3"), ("m3", ["-b", "40"], "b40"),
                   ("a7", ["-d", "20"], "d20"), ("f3", ["-b", "10", "-d", "20"], "b10_d20")]


def _read_trace(lib, path, block=-1, discard=0):
    import ctypes as C
    n = C.c_size_t(0)
    err = C.create_string_buffer(512)
    lib.gph_read_trace.restype = C.c_int
    lib.gph_read_trace.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.c_char_p, C.c_size_t]
    rc = lib.gph_read_trace(path.encode(), block, discard, None, 0, C.byref(n), err, 512)
    if rc != 0:
        return rc, "", err.value.decode()
    buf = C.create_string_buffer(n.value + 1)
    rc = lib.gph_read_trace(path.encode(), block, discard, buf, n.value + 1, C.byref(n), err, 512)
    return rc, buf.value.decode(), err.value.decode()


@pytest.mark.parametrize("name,args,tag", READTRACE_CASES)
def test_read_trace_prints_what_the_reference_tool_prints(lib, name, args, tag):
    """gph_read_trace / the readTrace executable against the text the reference's own readTrace (src/readTrace.c,
    compiled unmodified into oracle/_ref/readTrace_ref) printed for its own trace files."""
    import subprocess
    want = open(os.path.join(GOLDEN, f"{name}.readtrace_{tag}.txt")).read()
    block = int(args[args.index("-b") + 1]) if "-b" in args else -1
    discard = int(args[args.index("-d") + 1]) if "-d" in args else 0
    rc, got, err = _read_trace(lib, os.path.join(GOLDEN, name + ".trace"), block, discard)
    assert rc == 0 and err == ""
    assert got == want
    # the executable: same text on stdout
    exe = os.path.join(REPO, "tests", "hostemu", "readTrace_test")
    src = os.path.join(G.CSRC, "gph_readtrace.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-DGPH_READTRACE_MAIN", src, "-o", exe], check=True)
    out = subprocess.run([exe, os.path.join(GOLDEN, name + ".trace")] + args, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == want


def test_read_trace_partial_tail_and_errors(lib, tmp_path):
    """A trailing partial block is averaged over its own length (readTrace.c:253-264; upstream then prints
    uninitialised storage for it unless the column widens, here the means are printed); error paths."""
    path = os.path.join(GOLDEN, "g1.trace")
    rows = [l.split() for l in open(path).read().strip().split("\n")]
    data = np.array([[np.float32(x) for x in r[1:]] for r in rows[1:]], dtype=np.float64)   # %f into a float
    rc, got, _ = _read_trace(lib, path, block=4, discard=1)
    assert rc == 0
    lines = got.split("\n")
    assert lines[0].split() == rows[0][1:]
    body = data[1:]
    nblk = (len(body) + 3) // 4
    assert len(lines) == 1 + nblk + 2          # title, blocks, the closing empty line
    for b in range(nblk):
        blk = body[4 * b:4 * b + 4]
        want = ["%.6f" % v for v in blk.sum(axis=0) / len(blk)]
        assert lines[1 + b].split() == want
    rc, _, err = _read_trace(lib, path, discard=100)
    assert rc == 1 and err == "100 lines specified to discard, but trace file contains only 30 lines.\n"
    rc, _, err = _read_trace(lib, os.path.join(tmp_path, "nope.trace"))
    assert rc == 1 and "Could not find trace file" in err
    # a last line without its newline is not a sample (readTrace.c:217-218), but it was counted (:121-131)
    p = os.path.join(tmp_path, "t.trace")
    open(p, "w").write("Sample\ta\tb\n0\t1.0\t2.0\n1\t3.0\t5.0\n2\t100.0\t100.0")
    rc, got, _ = _read_trace(lib, p)
    assert rc == 0 and got.split("\n")[1].split() == ["2.000000", "3.500000"]


@pytest.mark.parametrize("name", ["g1", "f3", "v8", "a7", "w2", "r5"])
def test_program_prints_the_reference_log(name, tmp_path):
    """stdout of gph_run_control_file against the real binary's stdout for the same control file (tests/golden/*.stdout):
    from "Reading control settings" on (i.e. everything but the version banner and the thread-count line) -- title, one `\\r`-refreshed line per log period with the acceptance percentages exactly as
    upstream computes them (GPhoCS.c:1821-1895, quirks included), the finetune-search lines, the closing line -- equal
    but for the elapsed-time column.  a7: upstream's first TAU entry of an estimated sample age is read before it is
    ever written (GPhoCS.c:1620-1628), that one number is skipped."""
    import re
    import subprocess
    for ext in (".ctl", ".seq", ".rates"):
        if os.path.exists(os.path.join(GOLDEN, name + ext)):
            shutil.copy(os.path.join(GOLDEN, name + ext), tmp_path)
    ctl2 = SECONDARY.get(name)
    if ctl2:
        shutil.copy(os.path.join(GOLDEN, ctl2), tmp_path)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import gphocs_amd as G, run_hostemu as R\n"
            "lib = G.load_library(R.build_hostemu())\n"
            "lib.gph_run_control_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]\n"
            "sys.exit(lib.gph_run_control_file(%r, %r, 0, 0))\n") % (REPO, os.path.join(REPO, "tests", "hostemu"),
                                                                   (name + ".ctl").encode(), ctl2.encode() if ctl2 else None)
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

    def norm(t):
        t = t.replace("\r", "\n")
        t = re.sub(r"\|\s*\d+:\d\d(:\d\d)?", "| T", t)
        t = re.sub(r"Time used:\s+\d+:\d\d(:\d\d)?", "Time used: T", t)
        return t[t.index("Reading control settings"):].split("\n")
    want, got = norm(open(os.path.join(GOLDEN, name + ".stdout")).read()), norm(r.stdout)
    assert len(want) == len(got)
    for i, (w, g) in enumerate(zip(want, got)):
        if name == "a7" and w != g and w.split()[:6] == g.split()[:6] and w.split()[7:] == g.split()[7:]:
            continue      # the uninitialised TAU entry of the first period
        assert w == g, (i, w, g)


def test_reference_sample_control_file_end_to_end(lib, ref_cli, tmp_path):
    """BASELINE configs[0]: the reference's OWN sample-control-file.ctl (read where it lies under /root/reference; it is
    not copied into the repo, so this runs in the build container only).  Its sequence file is absent upstream
    (.MISSING_LARGE_BLOBS), so a small synthetic one with the file's sample names is written here; a secondary control
    file gives the seed and a short run.  The model, the priors, every phased pattern table and the trace file must equal
    the real reference's for the same three files."""
    import random
    import subprocess
    src = "/root/reference/sample-control-file.ctl"
    if not os.path.exists(src) or ref_cli is None:
        pytest.skip("the reference tree is absent on this box")
    shutil.copy(src, tmp_path / "sample.ctl")
    rnd = random.Random(20261002)
    names = ["one", "two", "three", "five"]           # the `samples ... d` entries of the file's four current populations
    with open(tmp_path / "seqs-sample.txt", "w") as f:
        f.write("9\n\n")
        for g in range(9):
            L = 220
            base = [rnd.choice("TCAG") for _ in range(L)]
            f.write(f"locus{g + 1} {len(names)} {L}\n")
            for s in names:
                row = []
                for b in base:
                    u = rnd.random()
                    row.append(b if u < 0.93 else rnd.choice("TCAG") if u < 0.96 else rnd.choice("YRMKSW") if u < 0.995 else "N")
                f.write(f"{s}\t{''.join(row)}\n")
            f.write("\n")
    (tmp_path / "short.ctl").write_text("GENERAL-INFO-START\n\trandom-seed 4711\n\tmcmc-iterations 40\n\titerations-per-log 10\n"
                                        "\tlogs-per-line 2\nGENERAL-INFO-END\n")
    env = dict(os.environ, GPH_REF_CTL2="short.ctl")
    subprocess.run([ref_cli, "pack", "sample.ctl", "ref.gpk"], cwd=tmp_path, check=True, capture_output=True, timeout=300, env=env)
    ref = G.Pack.load(str(tmp_path / "ref.gpk"))
    with _in_dir(tmp_path):
        got = G.Pack.from_control("sample.ctl", lib=lib, secondary="short.ctl")
    assert (got.n, got.Kc, got.K, got.B) == (8, 4, 7, 1) == (ref.n, ref.Kc, ref.K, ref.B)
    for f in ("rootPop", "L", "seed", "samplesPerLog", "numParameters", "ftCoalTime", "ftMigTime", "ftTheta", "ftMigRate", "ftMixing", "popName"):
        assert getattr(ref, f) == getattr(got, f), f
    for f in ("samplesPerPop", "popFather", "popSon0", "popSon1", "thetaAlpha", "thetaBeta", "thetaStart", "ageAlpha", "ageBeta",
              "ageStart", "ftTaus", "printFactors", "pattern_offsets", "leafcodes", "numPhases", "counts"):
        assert np.array_equal(np.asarray(getattr(ref, f)), np.asarray(getattr(got, f))), f
    # the whole program: the real binary's trace file (mcmc.log) against gph_run_control_file's
    subprocess.run([ref_cli, "main", "-n", "1", "sample.ctl", "short.ctl"], cwd=tmp_path, check=True, capture_output=True, timeout=600)
    want = open(tmp_path / "mcmc.log").read()
    os.remove(tmp_path / "mcmc.log")
    with _in_dir(tmp_path):
        assert lib.gph_run_control_file(b"sample.ctl", b"short.ctl", 0, 0) == 0
    got_t = open(tmp_path / "mcmc.log").read()
    w, g = want.splitlines(), got_t.splitlines()
    assert w[0] == g[0] and len(w) == len(g) == 41
    for a, b in zip(w[1:], g[1:]):
        if a != b:
            af, bf = [float(x) for x in a.split()], [float(x) for x in b.split()]
            assert all(abs(x - y) <= 1.5e-5 * max(1.0, abs(x)) for x, y in zip(af, bf)), (a, b)


@pytest.mark.parametrize("kind", ["few", "many", "neg", "missing"])
def test_rate_file_errors_carry_the_reference_text(lib, kind, tmp_path):
    """locus-mut-rate FIXED <file> with too few / too many / a non-positive entry / no file: the message bodies of
    readRateFile (GPhoCS.c:491-579) and of its caller (:1149-1154) -- tests/golden/r5_<kind>.stderr is the stderr of the
    real binary run on the same control file."""
    import ctypes as C
    import subprocess
    for f in ("r5_%s.ctl" % kind, "r5_%s.rates" % kind, "r5.seq"):
        if os.path.exists(os.path.join(GOLDEN, f)):
            shutil.copy(os.path.join(GOLDEN, f), tmp_path)
    want = [ln for ln in open(os.path.join(GOLDEN, "r5_%s.stderr" % kind)).read().splitlines() if ln.strip()]
    # through the C ABI: the body of the first line
    err = C.create_string_buffer(512)
    ctl, loci = C.c_void_p(), C.c_void_p()
    with _in_dir(tmp_path):
        assert lib.gph_control_read(("r5_%s.ctl" % kind).encode(), None, C.byref(ctl)) == 0
        rc = lib.gph_loci_read(ctl, None, 1, C.byref(loci), err, 512)
    assert rc != 0 and "Error: " + err.value.decode() == want[0]
    lib.gph_control_free(ctl)
    # through the program: the same lines on stderr (next to the program's own status line)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import gphocs_amd as G, run_hostemu as R\n"
            "lib = G.load_library(R.build_hostemu())\n"
            "lib.gph_run_control_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]\n"
            "sys.exit(1 if lib.gph_run_control_file(%r, None, 0, 0) else 0)\n") % (REPO, os.path.join(REPO, "tests", "hostemu"),
                                                                                 ("r5_%s.ctl" % kind).encode())
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1
    got = [ln for ln in r.stderr.splitlines() if ln.startswith("Error: ")]
    assert got == want, (got, want)
