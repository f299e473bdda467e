"""The drop-in boundary, EXECUTED: the reference's own program -- its main(), control-file and sequence readers,
performMCMC with the trace writer, the finetune search and the log lines, samplePopParameters / sampleMigRates, all
compiled unmodified from /root/reference/src -- with the functions performMCMC calls (upstream src/GPhoCS.h:84-100:
initializeMCMC, UpdateGB_InternalNode / _MigrationNode / _MigSPR, UpdateLocusRate, UpdateTheta, UpdateMigRates, UpdateTau,
UpdateSampleAge, mixing; patch.h: synchronizeEvents, gtreeLnLikelihood, checkAll) replaced at link time by
oracle/integration_binding.c, whose bodies are calls of include/gphocs_hip.h's per-function entry points.  Nothing of
the reference's per-locus path runs.  The program must write the trace file the unmodified reference binary wrote
(tests/golden/*.trace) -- migration bands and rubber-band conflicts (m3), estimated sample ages (a7), the reference's
own find-finetunes search steering the engine's step sizes (f3), `locus-mut-rate VAR` (v8), a secondary control file
(w2), the engine's hard caps (x8).

CPU: the binding over the host build of the engine sources (oracle/_ref/gphocs_boundary_emu).  -m gpu: the same
binding over the gfx950 library (gphocs_boundary_hip): the reference's own driver on the MI355X.
The binaries are built where /root/reference exists (`make -C oracle boundary`) and travel prebuilt to the GPU box."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import GOLDEN, ORACLE_DIR, REPO

CASES = ["g1", "m3", "a7", "f3", "v8", "w2", "x8", "r5", "j1", "j2", "j3"]   # (y9 needs the big-tree build of the engine sources: tests/test_host_logic.py, -m gpu)


def _build():
    if os.path.isdir("/root/reference/src"):
        sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
        import run_hostemu
        run_hostemu.build_hostemu()
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True, timeout=900)
        subprocess.run(["make", "-C", ORACLE_DIR, "boundary"], check=True, capture_output=True, timeout=900)


def _run(exe, name, tmp_path):
    for f in os.listdir(GOLDEN):
        if f.startswith(name + ".") and f.endswith((".ctl", ".seq", ".rates")) or f == name + "b.ctl":
            shutil.copy(os.path.join(GOLDEN, f), tmp_path)
    args = [exe, name + ".ctl"] + ([name + "b.ctl"] if os.path.exists(os.path.join(GOLDEN, name + "b.ctl")) else [])
    r = subprocess.run(args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # the reference's own performMCMC ran: its version banner, its log header
    assert "G-Phocs version" in r.stdout and "CoalTimes" in r.stdout and "MCMC finished" in r.stdout
    want = open(os.path.join(GOLDEN, name + ".trace")).read()
    got = open(os.path.join(tmp_path, name + ".trace")).read()
    return want, got


@pytest.mark.parametrize("name", CASES)
def test_reference_performMCMC_over_the_engine(tmp_path, name):
    _build()
    exe = os.path.join(ORACLE_DIR, "_ref", "gphocs_boundary_emu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gphocs_boundary_emu is built where /root/reference exists")
    want, got = _run(exe, name, tmp_path)
    assert got == want          # byte for byte: same decisions, same printed digits


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_reference_performMCMC_over_the_engine_on_the_gpu(tmp_path, name):
    exe = os.path.join(ORACLE_DIR, "_ref", "gphocs_boundary_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gphocs_boundary_hip was not built (needs /root/reference at build time)")
    want, got = _run(exe, name, tmp_path)
    # cross-locus sums are a fixed-shape tree on the device: only the two log-likelihood columns may differ (1e-10 relative),
    # every parameter column is character-identical
    from parity_util import compare_trace_files
    (tmp_path / "want.trace").write_text(want)
    assert compare_trace_files(tmp_path / "want.trace", os.path.join(tmp_path, name + ".trace")) <= len(want.splitlines()) // 10
