"""Model shapes beyond the caterpillar: random binary population trees, random legal migration bands
incl. ANCESTRAL endpoints, random sample counts, optional fixed / estimated ancient sample -- tools/random_models.py.

Fixtures (tests/golden/rnd/rKK.{gpk,rtrace,json}) come from the REAL reference: its pack and its per-proposal records.  Three of
them are models on which the reference itself ABORTS (a band a few 1e-9 long: "Fatal Error 0025" in recalcStats, patch.c:2452-2466,
or a failed checkAll): the engine must stop in the same iteration, after the same records.

 * not gpu: the oracle reproduces every fixture's records byte for byte; the host build of the engine sources reproduces them
   (counters exact) and the oracle's final per-locus state byte for byte; where oracle/_ref is present, 20 FRESH models are run
   through the real reference and the oracle (records + state byte for byte).
 * gpu: the HIP library through the C ABI -- records against the reference's, final per-locus state against the live oracle's,
   byte for byte; the abort cases fail in the reference's iteration with GPH_EKERNEL.
The band-start branches of UpdateTau (GPhoCS.c:3353-3431), tau bounds from two ancestral sons (:3266-3267) and rubberBandRipple
with start_or_end == 1 (patch.c:815-869) are reached by these models and by goldens j1-j3, and by nothing else in tests/."""
import glob
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO
from parity_util import compare_records, compare_states

RND = os.path.join(GOLDEN, "rnd")
MODELS = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(RND, "r*.json")))
META = {m: json.load(open(os.path.join(RND, m + ".json"))) for m in MODELS}
RUNS = [m for m in MODELS if not META[m]["reference_aborts"]]
ABORTS = [m for m in MODELS if META[m]["reference_aborts"]]


def test_fixture_set_is_what_the_docstring_says():
    assert len(RUNS) >= 12 and len(ABORTS) == 3
    anc = [m for m in RUNS if any(len(s) > 1 or len(t) > 1 for s, t in META[m]["model"]["bands"])]
    assert len(anc) >= 10, "bands with an ancestral endpoint"
    conflicts = 0
    for m in RUNS:
        c = [int(l.split()[1]) for l in open(os.path.join(RND, m + ".rtrace")) if l.startswith("CONFLICTS")]
        conflicts += c[-1] if c else 0
    assert conflicts > 100, "rubber-band conflicts of UpdateTau / UpdateSampleAge"


def _records_before(path, it):
    out = []
    for l in open(path).read().splitlines():
        if l.startswith("IT ") and int(l.split()[1]) >= it:
            break
        out.append(l)
    return out


@pytest.mark.parametrize("name", MODELS)
def test_oracle_reproduces_the_reference_records(oracle_cli, name, tmp_path):
    it = META[name]["run"]["iters"]
    tr = tmp_path / "o.rtrace"
    r = subprocess.run([oracle_cli, "run", os.path.join(RND, name + ".gpk"), str(it), str(tr), str(tmp_path / "o.state"), str(it - 1), "0"],
                       capture_output=True, timeout=600)
    assert (r.returncode != 0) == META[name]["reference_aborts"], r.stderr[-300:]
    assert open(tr).read() == open(os.path.join(RND, name + ".rtrace")).read()


def _engine_run(G, lib, name, tmp_path, tag):
    """-> (records file, final state file or None, iteration in which the engine failed or None)"""
    it = META[name]["run"]["iters"]
    tr, st = str(tmp_path / f"{tag}.rtrace"), str(tmp_path / f"{tag}.state")
    s = G.Sampler(G.Pack.load(os.path.join(RND, name + ".gpk")), lib=lib) if lib is not None else G.Sampler(G.Pack.load(os.path.join(RND, name + ".gpk")))
    s.set_record_file(tr)
    failed = None
    try:
        try:
            s.initialize()
        except RuntimeError:
            failed = -1
        if failed is None:
            for k in range(it):
                try:
                    s.iteration(k)
                except RuntimeError:
                    failed = k
                    break
        if failed is None:
            s.dump_state(st, True)
    finally:
        s.set_record_file(None)
        s.close()
    return tr, (st if failed is None else None), failed


def _check_engine(G, lib, oracle_cli, name, tmp_path):
    it = META[name]["run"]["iters"]
    tr, st, failed = _engine_run(G, lib, name, tmp_path, "e")
    ref = os.path.join(RND, name + ".rtrace")
    if META[name]["reference_aborts"]:
        last = META[name]["reference_last_iteration"]
        # the reference dies inside iteration `last` (or right after its records, in the checkAll that follows): the engine in the same one
        assert failed is not None and failed in (last, last + 1), f"engine failed in {failed}, the reference in {last}"
        a, b = _records_before(tr, failed), _records_before(ref, failed)
        pa, pb = tmp_path / "a.part", tmp_path / "b.part"
        pa.write_text("".join(l + "\n" for l in a))
        pb.write_text("".join(l + "\n" for l in b))
        compare_records(pa, pb)
        return
    assert failed is None, f"engine failed in iteration {failed}"
    compare_records(tr, ref)
    ot, os_ = tmp_path / "o.rtrace", tmp_path / "o.state"
    subprocess.run([oracle_cli, "run", os.path.join(RND, name + ".gpk"), str(it), str(ot), str(os_), str(it - 1), "1"], check=True, timeout=600)
    compare_states(st, os_)


@pytest.mark.parametrize("name", MODELS)
def test_host_build_on_random_models(oracle_cli, name, tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
    import run_hostemu as R
    import gphocs_amd as G
    # 20 leaves / 13 populations / 8 bands at most: the small host build (32 / 32 / 16)
    _check_engine(G, G.load_library(R.build_hostemu()), oracle_cli, name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
def test_hip_on_random_models(oracle_cli, name, tmp_path):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import gphocs_amd as G
    G.build()
    _check_engine(G, None, oracle_cli, name, tmp_path)


def test_fresh_random_models_reference_vs_oracle(oracle_cli, ref_cli, tmp_path):
    """20 models that are NOT committed fixtures (ids 200-219), through the real reference and the oracle"""
    if ref_cli is None or not os.path.isdir("/root/reference/src"):
        pytest.skip("oracle/_ref/gphocs_ref not built (the GPU box): the committed fixtures cover this there")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import random_models as RM
    bad = 0
    for k in range(200, 220):
        name, cfg, run = RM.generate(k, str(tmp_path))
        rc = RM.reference_run(name, str(tmp_path), run["iters"])
        o = subprocess.run([oracle_cli, "run", name + ".gpk", str(run["iters"]), name + ".o.rtrace", name + ".o.state", str(run["iters"] - 1), "1"],
                           cwd=tmp_path, capture_output=True, timeout=600)
        assert rc[0] == 0

        def rd(ext):
            p = tmp_path / (name + ext)
            return p.read_bytes() if p.exists() else b""
        aborted = rc[1] != 0
        ok = (o.returncode != 0) == aborted and rd(".rtrace") == rd(".o.rtrace") and (aborted or rd(".state") == rd(".o.state"))
        bad += not ok
    assert bad == 0
