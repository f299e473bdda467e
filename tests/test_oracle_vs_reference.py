"""Pins the CPU restatement (oracle/) against golden vectors generated from the REAL reference
(tests/golden/make_goldens.sh -> oracle/_ref/gphocs_ref).  Everything is compared EXACTLY (hex
floats, integer accept counters, full per-locus state incl. conditional-likelihood arrays):
the restatement is serial and keeps the reference's floating-point operation order.
CPU-only; no GPU needed."""
import filecmp
import os
import subprocess

import pytest

from conftest import GOLDEN

CASES = {  # name: iterations (must match make_goldens.sh)
    "g1": 30, "g2": 30, "m3": 120, "m4": 60, "c5": 30, "s3": 40, "a6": 80, "a7": 100, "z0": 12, "v8": 60, "v9": 60,
    "w2": 50,   # model read from a primary + a secondary control file
    "r5": 60,   # locus-mut-rate FIXED r5.rates: per-locus rates spread over 0.2 .. 5 (readRateFile, GPhoCS.c:491-579)
    "x8": 24,   # 32 leaves, 31 populations, 16 bands
    "y9": 16,   # 40 leaves, 39 populations (the reference's NSPECIES cap), 16 bands
    "n7": 12,   # 72 leaves (beyond the 64 of 128-bit node sets; the reference allows 200)
    "b2": 24,   # 20 migration bands (beyond the 16 of the nibble list; the reference allows 100)
    "j1": 150, "j2": 100, "j3": 120,   # balanced / mixed population trees, bands with ancestral endpoints, estimated ancient sample below a band target
    "q6": 8,    # 72 leaves, two 20-kb loci with 145 / 698 phased patterns, up to 512 phases per pattern (state dumps without conditionals)
}
NOCOND = {"q6"}


@pytest.mark.parametrize("seed", [12345, 777])
def test_rng_stream(oracle_cli, seed, tmp_path):
    out = subprocess.run([oracle_cli, "rng", str(seed), "300"], check=True, capture_output=True,
                         text=True, timeout=60).stdout
    assert out == open(os.path.join(GOLDEN, f"rng_{seed}.txt")).read()


def test_reflect_table(oracle_cli):
    out = subprocess.run([oracle_cli, "reflect"], check=True, capture_output=True, text=True,
                         timeout=60).stdout
    assert out == open(os.path.join(GOLDEN, "reflect.txt")).read()


@pytest.mark.parametrize("name", sorted(CASES))
def test_initial_state(oracle_cli, name, tmp_path):
    """initializeMCMC: prior-sampled genealogies, event chains, statistics, full pruning."""
    st = tmp_path / "init.state"
    subprocess.run([oracle_cli, "run", os.path.join(GOLDEN, name + ".gpk"), "0", str(tmp_path / "t"),
                    str(st), "-1", "0" if name in NOCOND else "1"], check=True, timeout=300)
    assert filecmp.cmp(st, os.path.join(GOLDEN, name + ".init.state"), shallow=False)


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_run(oracle_cli, name, tmp_path):
    """every proposal's accept count + accumulators for all iterations, then the full final state"""
    tr, st = tmp_path / "trace", tmp_path / "state"
    it = CASES[name]
    subprocess.run([oracle_cli, "run", os.path.join(GOLDEN, name + ".gpk"), str(it), str(tr), str(st),
                    str(it - 1), "0" if name in NOCOND else "1"], check=True, timeout=900)
    assert open(tr).read() == open(os.path.join(GOLDEN, name + ".rtrace")).read()
    assert filecmp.cmp(st, os.path.join(GOLDEN, name + ".state"), shallow=False)


def test_goldens_exercise_migration_paths():
    """the fixtures must cover migration events and rubber-band conflicts, not just the easy path"""
    st = open(os.path.join(GOLDEN, "m4.state")).read()
    assert sum(1 for l in st.splitlines() if l.startswith("M ") and not l.startswith("M 0")) > 3
    tr = open(os.path.join(GOLDEN, "m3.rtrace")).read().splitlines()
    conflicts = [int(l.split()[1]) for l in tr if l.startswith("CONFLICTS")]
    assert conflicts[-1] > 0
    assert sum(int(l.split()[3]) for l in tr if " MIGN " in l) > 100


def test_live_reference_if_present(oracle_cli, ref_cli, tmp_path):
    """where the prebuilt real reference is present, run it live and compare a fresh case"""
    if ref_cli is None:
        pytest.skip("oracle/_ref/gphocs_ref not built")
    ctl = os.path.join(GOLDEN, "m3.ctl")
    rt, ot = tmp_path / "r.trace", tmp_path / "o.trace"
    subprocess.run([ref_cli, "run", ctl, "25", str(rt)], check=True, cwd=GOLDEN, timeout=600,
                   capture_output=True)
    subprocess.run([oracle_cli, "run", os.path.join(GOLDEN, "m3.gpk"), "25", str(ot)], check=True,
                   timeout=600)
    assert open(rt).read() == open(ot).read()
