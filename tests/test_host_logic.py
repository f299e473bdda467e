"""CPU-only checks of the product's host logic and of its per-locus code compiled for the host.

The HIP library has no CPU path; `tests/hostemu/` compiles the SAME sources with -DGPH_HOSTEMU (a
1-lane "wave", plain memory instead of LDS/HBM) purely as a test/debug build (sanitizers, gdb, and
the world_size-2 gloo test) -- it is not linked into libgphocs_hip.so.  These tests pin the engine's
logic to the real-reference goldens in the GPU-less container; the -m gpu tests do the same on the
device through the C ABI."""
import ctypes
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO
from parity_util import compare_records, compare_states

sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))


@pytest.fixture(scope="module")
def hostemu():
    import run_hostemu as R
    import gphocs_amd as G
    return R, G.load_library(R.build_hostemu())


@pytest.mark.parametrize("name,iters", [("g1", 30), ("g2", 30), ("m3", 120), ("m4", 60), ("c5", 30), ("s3", 40), ("a6", 80), ("a7", 100), ("z0", 12), ("v8", 60), ("v9", 60), ("w2", 50), ("x8", 24), ("r5", 60), ("j1", 150), ("j2", 100), ("j3", 120)])
def test_hostemu_matches_reference_goldens(hostemu, name, iters, tmp_path):
    R, lib = hostemu
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(os.path.join(GOLDEN, name + ".gpk"), iters, str(tr), str(st), iters - 1, lib=lib)
    worst = compare_records(tr, os.path.join(GOLDEN, name + ".rtrace"))
    compare_states(st, os.path.join(GOLDEN, name + ".state"))
    assert worst < 1e-12


@pytest.mark.parametrize("name,iters", [("m3", 120), ("v8", 60)])
def test_sequence_block_with_32bit_counts(hostemu, name, iters, tmp_path, monkeypatch):
    """the sequence block stores pattern counts as 16-bit words when every count of the data set allows it; the 32-bit
    form (a locus longer than 65 535 sites can have such a count) must read the same values"""
    R, lib = hostemu
    monkeypatch.setenv("GPH_CNT16", "0")
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(os.path.join(GOLDEN, name + ".gpk"), iters, str(tr), str(st), iters - 1, lib=lib)
    worst = compare_records(tr, os.path.join(GOLDEN, name + ".rtrace"))
    compare_states(st, os.path.join(GOLDEN, name + ".state"))
    assert worst < 1e-12


@pytest.mark.parametrize("name,iters", [("y9", 16), ("x8", 24), ("m3", 120), ("a7", 100), ("v8", 60), ("b2", 24), ("n7", 12), ("q6", 8), ("j1", 150), ("j2", 100), ("j3", 120)])
def test_big_tree_build_matches_reference_goldens(name, iters, tmp_path):
    """the 64-leaf / 39-population capacities (library variant `h`: 128-bit node sets, 64-bit population sets, 16-bit
    event ids, list-driven forms instead of the lane-per-node programs): golden y9 -- 40 leaves, 20 current populations
    = the reference's NSPECIES cap (patch.h:19), 16 bands, from the real reference -- and goldens of the smaller builds"""
    import run_hostemu as R
    import gphocs_amd as G
    lib = G.load_library(R.build_hostemu(big=True))
    tr, st = tmp_path / "t", tmp_path / "s"
    # q6: 72 leaves, two 20-kb loci with 145 / 698 phased patterns: the second one's sequence block stays in HBM (no
    # conditional arrays in its golden state: 4 MB of hex floats)
    R.run(os.path.join(GOLDEN, name + ".gpk"), iters, str(tr), str(st), iters - 1, with_cond=name != "q6", lib=lib)
    worst = compare_records(tr, os.path.join(GOLDEN, name + ".rtrace"))
    compare_states(st, os.path.join(GOLDEN, name + ".state"))
    assert worst < 1e-12


@pytest.mark.parametrize("name,iters", [("y9", 16), ("m3", 120), ("a7", 100), ("m4", 60), ("x8", 24), ("j1", 150), ("j2", 100), ("j3", 120)])
def test_variant_h_configuration_matches_reference_goldens(name, iters, tmp_path):
    """the configuration of library variants g / h (64 leaves / 39 populations / 16 bands: GPH_BIG_TREE with
    two-word node sets and WITHOUT the many-band forms -- the fused trace_pair walk with its parked state, lik_spr over
    multi-word node sets) has a CPU build of its own: y9 and the SPR- / migration-heavy goldens"""
    import run_hostemu as R
    import gphocs_amd as G
    lib = G.load_library(R.build_hostemu(mid=True))
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(os.path.join(GOLDEN, name + ".gpk"), iters, str(tr), str(st), iters - 1, lib=lib)
    worst = compare_records(tr, os.path.join(GOLDEN, name + ".rtrace"))
    compare_states(st, os.path.join(GOLDEN, name + ".state"))
    assert worst < 1e-12


@pytest.mark.parametrize("name,iters", [("m3", 120), ("a7", 40), ("x8", 24), ("j1", 50), ("j2", 40), ("v8", 30), ("bigp", 20), ("stress", 6),
                                        ("y9@mid", 16), ("j1@mid", 30), ("b2@big", 24)])
def test_wave64_device_forms_match_reference_goldens(oracle_cli, name, iters, tmp_path):
    """round 6: the DEVICE forms of the lane-parallel functions -- lik_compute (ballot fix-point, a lane
    per node and per pattern, conditionals forwarded in registers), prune_node_q / child_factor4, add_phases (DPP shifts),
    ordered_sum64 (lane reads), the generic (pattern, base) pruning and root sum of pattern-rich loci, edges_for_time_pop
    (ballot-ordered candidate list) -- compiled for the HOST and run on a 64-lane micro-wave of fibers (csrc/gph_emu64.h) inside
    the host build: every cross-lane operation and every accessor store is a rendezvous of all 64 lanes, lanes that wait at
    different sites abort.  Records and per-locus state of the real reference's goldens (bigp / stress: the live oracle), and the
    same bytes as the one-lane list forms of the same library (GPH_EMU64=0 semantics: the plain host build)."""
    import ctypes as C
    import run_hostemu as R
    import gphocs_amd as G
    # @mid / @big: the capacities of library variants g / h (64 leaves, 39 populations) and b / n (200 leaves, 100 bands): there the
    # micro-wave runs the LIST-DRIVEN device forms of the big-tree builds (edge probabilities on a lane per child node through LDS,
    # a lane per pattern with register forwarding, multi-word node sets, the 64-nodes-a-round candidate test of the SPR)
    kind = dict(mid=name.endswith("@mid"), big=name.endswith("@big"))
    name = name.split("@")[0]
    lib = G.load_library(R.build_hostemu(wave64=True, **kind))
    w0, r0 = C.c_longlong(), C.c_longlong()
    assert lib.gph_debug_emu64_stats(C.byref(w0), C.byref(r0)) == 1
    pack = os.path.join(GOLDEN, name + ".gpk")
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pack, iters, str(tr), str(st), iters - 1, lib=lib)
    w1, r1 = C.c_longlong(), C.c_longlong()
    lib.gph_debug_emu64_stats(C.byref(w1), C.byref(r1))
    assert w1.value - w0.value > 50 * iters and r1.value - r0.value > 20 * (w1.value - w0.value), (w1.value, r1.value)
    gold = os.path.join(GOLDEN, name + ".rtrace")
    if os.path.exists(gold) and iters == {"m3": 120, "x8": 24, "y9": 16, "b2": 24}.get(name):          # the golden's own length: records and final state
        assert compare_records(tr, gold) < 1e-12
        compare_states(st, os.path.join(GOLDEN, name + ".state"))
    elif os.path.exists(gold):                                                        # a prefix of the golden's records
        mine = open(tr).read().splitlines()
        (tmp_path / "g").write_text("".join(l + "\n" for l in open(gold).read().splitlines()[:len(mine)]))
        assert compare_records(tr, tmp_path / "g") < 1e-12
    else:
        ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
        subprocess.run([oracle_cli, "run", pack, str(iters), str(ot), str(os_), str(iters - 1), "1"], check=True, timeout=600)
        assert compare_records(tr, ot) < 1e-12
        compare_states(st, os_)
    # the one-lane host build: byte for byte the same records and state
    tr1, st1 = tmp_path / "t1", tmp_path / "s1"
    R.run(pack, iters, str(tr1), str(st1), iters - 1, lib=G.load_library(R.build_hostemu(**kind)))
    assert open(tr).read() == open(tr1).read() and open(st).read() == open(st1).read()


def test_c_abi_library_exports_every_declared_symbol():
    """build the real HIP library (hipcc cross-compiles without a GPU) and check that every function
    declared in include/gphocs_hip.h is exported (no compute calls: there is no GPU here)"""
    import re
    import gphocs_amd as G
    G.build()
    lib = ctypes.CDLL(G.LIB_PATH)
    hdr = open(os.path.join(REPO, "include", "gphocs_hip.h")).read()
    names = set(re.findall(r"\b(gph_[a-z_0-9]+)\s*\(", hdr)) - {"gph_allreduce_fn"}
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/gphocs_hip.h but not exported"
    assert names == set(G.EXPORTS), names ^ set(G.EXPORTS)


def test_engine_creation_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gphocs_amd as G
    pk = G.Pack.load(os.path.join(GOLDEN, "g1.gpk"))
    with pytest.raises(RuntimeError):
        G.Sampler(pk)      # libgphocs_hip.so: no device -> GPH_EHIP, no silent CPU fallback


@pytest.mark.parametrize("loci,zero_ref", [(1, False), (2, False), (3, False), (40, False), (40, True)])
def test_locus_rate_edge_cases_against_live_oracle(hostemu, oracle_cli, tmp_path, loci, zero_ref):
    """`locus-mut-rate VAR` (UpdateLocusRate, GPhoCS.c:4598): a chain with only the reference locus, with one and two
    proposing loci, and with a reference locus that has no informative column (P = 0: its likelihood is 0 at any rate);
    large steps (reflections at both bounds) and alpha != 1.  Host build of the engine sources against the oracle's
    serial loop on the same pack: records and final per-locus state (rates included)."""
    import subprocess
    import numpy as np
    import gphocs_amd as G
    from gphocs_amd_pkg import synth
    R, lib = hostemu
    pk = synth.make_synthetic_pack(G.Pack, 3, loci, mut_scale=1.0, data_seed=5, mcmc_seed=99, samples_per_log=4)
    synth.make_var_rates(pk, 1.3, 0.8)
    if zero_ref:      # drop every pattern of locus 0
        p0 = int(pk.pattern_offsets[1])
        pk.leafcodes, pk.numPhases, pk.counts = pk.leafcodes[p0:], pk.numPhases[p0:], pk.counts[p0:]
        pk.pattern_offsets = np.concatenate([[0], pk.pattern_offsets[1:] - p0]).astype(np.int64)
    pth = str(tmp_path / "ve.gpk")
    synth.write_pack(pk, pth)
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pth, 12, str(tr), str(st), 11, lib=lib)
    ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
    subprocess.run([oracle_cli, "run", pth, "12", str(ot), str(os_), "11", "1"], check=True, timeout=300)
    assert compare_records(tr, ot) < 1e-12
    compare_states(st, os_)


def test_odd_leaf_count_and_pattern_rich_loci_against_live_oracle(hostemu, oracle_cli, tmp_path):
    """13 leaves (an odd count: the last byte of a pattern's 4-bit leaf codes is half used) and loci with up to 485 phased
    patterns (30 of the 40 with more than 64): the front end's `stress` pack through the engine sources, against the
    oracle's serial loop on the same pack"""
    R, lib = hostemu
    pack = os.path.join(GOLDEN, "stress.gpk")
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pack, 10, str(tr), str(st), 9, lib=lib)
    ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
    subprocess.run([oracle_cli, "run", pack, "10", str(ot), str(os_), "9", "1"], check=True, timeout=300)
    assert compare_records(tr, ot) < 1e-12
    compare_states(st, os_)


@pytest.mark.parametrize("budget", [8000, 10000, 11000])
def test_loci_whose_sequence_block_outgrows_lds_read_it_from_hbm(hostemu, oracle_cli, tmp_path, monkeypatch, budget):
    """a locus whose sequence block does not fit the LDS budget of a launch group is no longer
    refused -- the block stays in HBM and the generic (pattern, base) paths read it there (the reference mallocs any P,
    LocusDataLikelihood.c:251).  GPH_HUGE_LDS shrinks the budget so that the `stress` pack's pattern-rich loci (up to 485
    phased patterns) split into the three launch groups -- HBM block / LDS block with the generic mapping / a pattern per
    lane: records and per-locus state against the oracle's serial loop, as with everything in LDS"""
    R, lib = hostemu
    monkeypatch.setenv("GPH_HUGE_LDS", str(budget))
    pack = os.path.join(GOLDEN, "stress.gpk")
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pack, 10, str(tr), str(st), 9, lib=lib)
    ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
    subprocess.run([oracle_cli, "run", pack, "10", str(ot), str(os_), "9", "1"], check=True, timeout=300)
    assert compare_records(tr, ot) < 1e-12
    compare_states(st, os_)


@pytest.mark.parametrize("budget", [8000, 11000])
def test_variable_locus_rates_with_sequence_blocks_in_hbm(hostemu, oracle_cli, tmp_path, monkeypatch, budget):
    """round 6 (ADVICE): `locus-mut-rate VAR` together with loci whose sequence block stays in HBM was
    refused at the first UpdateLocusRate; the serial scan stages every block it evaluates from HBM into its own LDS, so the only
    limit left is that LDS (checked collectively).  The `stress` pack as a VAR chain with a large step, blocks forced into HBM:
    records, rates and per-locus state against the oracle's serial loop (GPhoCS.c:4598-4680)"""
    import gphocs_amd as G
    from gphocs_amd_pkg import synth
    R, lib = hostemu
    monkeypatch.setenv("GPH_HUGE_LDS", str(budget))
    pk = synth.make_var_rates(G.Pack.load(os.path.join(GOLDEN, "stress.gpk")), 1.4, 0.9)
    pk.popName = [f"p{k}" for k in range(pk.K)] if not getattr(pk, "popName", None) else pk.popName
    pth = str(tmp_path / "stress_var.gpk")
    synth.write_pack(pk, pth)
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pth, 12, str(tr), str(st), 11, lib=lib)
    ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
    subprocess.run([oracle_cli, "run", pth, "12", str(ot), str(os_), "11", "1"], check=True, timeout=300)
    assert compare_records(tr, ot) < 1e-12
    compare_states(st, os_)
    assert sum(int(l.split()[3]) for l in open(tr) if " LRATE " in l) > 20       # accepted rate proposals


def test_pattern_counts_beyond_16_bits_against_live_oracle(hostemu, oracle_cli, tmp_path):
    """a data set with pattern counts above 65 535 (long loci): the sequence block falls back to 32-bit counts by itself;
    host build of the engine sources against the oracle's serial loop on the same pack"""
    import subprocess
    import numpy as np
    import gphocs_amd as G
    from gphocs_amd_pkg import synth
    R, lib = hostemu
    pk = synth.make_synthetic_pack(G.Pack, 3, 24, mut_scale=1.0, data_seed=11, mcmc_seed=7, samples_per_log=4)
    pk.counts = (np.asarray(pk.counts, dtype=np.int64) * 400).astype(np.asarray(pk.counts).dtype)
    assert int(np.max(pk.counts)) > 65535
    pth = str(tmp_path / "big.gpk")
    synth.write_pack(pk, pth)
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pth, 12, str(tr), str(st), 11, lib=lib)
    ot, os_ = tmp_path / "o.t", tmp_path / "o.s"
    subprocess.run([oracle_cli, "run", pth, "12", str(ot), str(os_), "11", "1"], check=True, timeout=300)
    assert compare_records(tr, ot) < 1e-12
    compare_states(st, os_)


@pytest.mark.parametrize("name", ["m4", "a7", "g2", "j1", "j2"])     # j1 / j2: bands with ancestral ends (band START events inside a chain)
def test_kernel_level_fixtures_hostemu(hostemu, name):
    """single calls of the per-locus functions against the real reference's (tests/golden/*.unit), host-emulation build"""
    import gphocs_amd as G
    import unit_fixture
    _, lib = hostemu
    assert unit_fixture.check_unit(G, lib, GOLDEN, name) > 100
    # executeGenSPR with every return code, scaleAllNodeAges + revert, rubberBandRipple do / undo, traceLineage outcomes
    assert unit_fixture.check_unit2(G, lib, GOLDEN, name) > 1000


def test_reinitialise_drops_an_owed_mixing_commit(hostemu, tmp_path):
    """ADVICE round 3 (gph_engine_init_genealogies left mix_owed / fin_owed / sync_pending set)"""
    from parity_util import reinit_after_accepted_mixing
    import gphocs_amd as G
    R, lib = hostemu
    reinit_after_accepted_mixing(G, lib, os.path.join(GOLDEN, "m3.gpk"), tmp_path)


@pytest.mark.parametrize("config,loci,iters,mut", [(12, 40, 12, 3.0), (13, 12, 8, 2.0), (14, 6, 6, 0.5)])
def test_big_build_against_live_oracle(oracle_cli, tmp_path, config, loci, iters, mut):
    """the reference's own capacities on the host build of the engine sources (200 leaves / 39 populations / 100 bands) against
    the oracle run live: 20 migration bands (live-band list in LDS), 72 leaves (node sets of several words), 136 leaves with
    132 lineages in one population (lineage counts beyond a signed byte)"""
    import run_hostemu as R
    import gphocs_amd as G
    from gphocs_amd_pkg import synth
    lib = G.load_library(R.build_hostemu(big=True))
    pk = synth.make_synthetic_pack(G.Pack, config, loci, seqlen=300, mut_scale=mut, data_seed=31 + config, mcmc_seed=99, samples_per_log=4,
                                   mig_beta=4e-8)
    pth = str(tmp_path / "big.gpk")
    synth.write_pack(pk, pth)
    tr, st = tmp_path / "t", tmp_path / "s"
    R.run(pth, iters, str(tr), str(st), iters - 1, with_cond=False, lib=lib)
    ot, os_ = tmp_path / "ot", tmp_path / "os"
    subprocess.run([oracle_cli, "run", pth, str(iters), str(ot), str(os_), str(iters - 1), "0"], check=True, timeout=1200)
    assert compare_records(tr, ot) < 1e-10
    compare_states(st, os_)
    if config == 14:   # event records "id:type:node:lineages:time" of the state dump
        import re
        assert max(int(m.group(1)) for ln in open(st) for m in re.finditer(r" \d+:\d+:-?\d+:(-?\d+):0x", ln)) > 127


def fatal_error_names_the_locus(G, lib, tmp_path, capfd=None, victim=11, second=14):
    """a broken event chain in ONE locus: the failing call returns GPH_EKERNEL, gph_engine_last_error names that locus (the
    first failing one, global index) and a reference-style code, and stderr carries the locus's genealogy and event chains
    -- what printGenealogyAndExit prints upstream (GPhoCS.c:660-676)"""
    pk = G.Pack.load(os.path.join(GOLDEN, "m3.gpk"))
    s = G.Sampler(pk, lib=lib)
    s.initialize()
    for it in range(5):
        s.iteration(it)
    assert s.last_error() == (-1, 0)
    assert lib.gph_engine_debug_break_chain(s.engine, victim, 0 if victim == 11 else 1) == 0
    assert lib.gph_engine_debug_break_chain(s.engine, second, 1) == 0        # a second broken locus: the FIRST one is reported
    with pytest.raises(RuntimeError):
        s.iteration(5)
    locus, code = s.last_error()
    assert locus == victim and 0 < code < 10000 and code not in (75, 9999), (locus, code)
    s.close()
    return locus, code


@pytest.mark.parametrize("victim,second", [(11, 14), (13, 15)])
def test_fatal_error_names_the_locus_and_prints_its_genealogy(hostemu, tmp_path, capfd, victim, second):
    import gphocs_amd as G
    R, lib = hostemu
    locus, code = fatal_error_names_the_locus(G, lib, tmp_path, victim=victim, second=second)
    err = capfd.readouterr().err
    assert f"Fatal Error {code:04d}" in err and f"first in locus {locus}" in err
    assert f"LOCUS {locus} root" in err and "\nC 0" in err and "\nN 0 " in err
