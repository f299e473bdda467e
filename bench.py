#!/usr/bin/env python3
"""Benchmark of the hot path: full MCMC iterations of the per-locus likelihood engine.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  ... bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or bare -- then this
  process never touches a GPU: it starts the N rank processes itself (one per GPU, the engine's RCCL communicator
  between them), waits, relays rank 0's ONE JSON line and exits non-zero if there are fewer than N devices or any
  rank fails.  The reference's knob for the same thing is `-n threads` (GPhoCS.c:95, 116-145, MultiCoreUtils.h:8).

step      = one MCMC iteration of performMCMC's proposal sequence (GPhoCS.c:1476-1821) over every
            locus: fused genealogy sweep (node ages, migration ages, SPR), theta, migration rates,
            tau rubber bands, mixing, event synchronisation -- nothing skipped.
workload  = BASELINE.json configs[3], the one the metric is quoted on: 100k loci x 1 kb, 8 diploid
            samples (16 leaves), 5 current populations + 4 migration bands; synthetic data
            (g-phocs_amd/synth.py), resident in HBM before the timed region.
scaling   = strong (default): ONE --loci-locus data set (100k) sharded over the ranks in contiguous blocks, the
            reference's `-n threads` over a fixed numLoci (GPhoCS.c:116-145, MultiCoreUtils.h:8); --weak: --loci
            loci on every rank.  whole-job value = all ranks' units / max-over-ranks time.  Ranks share nothing but
            the reduced row (RCCL all-gather on the engine's stream) of each reduction point.
value     = locus-likelihood evaluations per second (computeLocusDataLikelihood(useOld=1)
            equivalents, counted by the kernels); MCMC iterations/s is reported next to it.
roofline  = dominant kernel (fused genealogy sweep): algorithmic bytes (96*R*P + 20*N + 8*U + 8 per
            evaluation, counted per evaluation by the kernel) / HIP-event duration, vs 8 TB/s HBM.
cpu_baseline = the REAL reference (oracle/_ref, OpenMP build).  VALUE: timed on ALL loci of the workload (--cpu-full-loci, default
            100 000 = the size the metric is quoted on), one start-up, 1 warm-up + 5 iterations (median) at 1 thread and at the
            best thread count of the thread sweep.  The thread sweep (`thread_sweep`): the first --cpu-loci loci (20 000: a
            0.7-GB working set, beyond any L3) at 1 / 8 / 16 / 32 / all host threads.  Secondary, labelled: the serial build and
            the oracle restatement on 5 000 loci.  Rank 0, N = 1 only.  The whole leg stays under five minutes.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402


def _gen_chunk(args):
    import gphocs_amd as G
    config, n, seed, mut_scale = args
    return G.make_synthetic_pack(G.Pack, config, n, mut_scale=mut_scale, data_seed=seed)


def build_workload(G, config, L, mut_scale, seed0, cache_dir, begin=0, L_total=None):
    """loci [begin, begin + L) of an L_total-locus synthetic data set of the config's shape, generated in
    parallel chunks (chunk c always has seed seed0 + 1000 + c, so a rank's shard is the same data whether it
    is generated alone or as part of the whole set) and cached as .npz"""
    os.makedirs(cache_dir, exist_ok=True)
    L_total = L if L_total is None else L_total
    key = os.path.join(cache_dir, f"synth_c{config}_L{L}_b{begin}_m{mut_scale}_s{seed0}.npz")
    base = G.Pack()
    from gphocs_amd_pkg import synth
    synth.make_model(base, config)
    chunk = 4000
    assert begin % chunk == 0 or L_total == L
    if os.path.exists(key):
        z = np.load(key)
        base.pattern_offsets, base.leafcodes = z["offs"], z["leaf"]
        base.numPhases, base.counts = z["phases"], z["counts"]
    else:
        jobs = [(config, min(chunk, L - i), seed0 + 1000 + (begin + i) // chunk, mut_scale) for i in range(0, L, chunk)]
        nproc = min(len(jobs), max(1, ((os.cpu_count() or 2) - 1) // max(1, L_total // L)), 16)
        if nproc > 1:
            with mp.get_context("fork").Pool(nproc) as pool:
                parts = pool.map(_gen_chunk, jobs)
        else:
            parts = [_gen_chunk(j) for j in jobs]
        offs = [0]
        for p in parts:
            offs.extend((p.pattern_offsets[1:] + offs[-1]).tolist())
        base.pattern_offsets = np.array(offs, np.int64)
        base.leafcodes = np.concatenate([p.leafcodes for p in parts])
        base.numPhases = np.concatenate([p.numPhases for p in parts])
        base.counts = np.concatenate([p.counts for p in parts])
        try:
            tmp = key + f".tmp{os.getpid()}.npz"
            np.savez(tmp, offs=base.pattern_offsets, leaf=base.leafcodes, phases=base.numPhases, counts=base.counts)
            os.replace(tmp, key)
        except OSError:
            pass
    base.L = base.numLoci = L
    base.mutRates = np.ones(L)
    if L_total != L:
        base.global_L, base.global_begin = L_total, begin
    return base


BASE_CODE = "TCAG"
IUPAC = {frozenset("CT"): "Y", frozenset("AG"): "R", frozenset("AC"): "M", frozenset("GT"): "K",
         frozenset("CG"): "S", frozenset("AT"): "W"}


def write_seq_sample(pack, nloci, path):
    """sequence file (the reference's unchanged input format) for the first `nloci` loci of a synthetic
    pack: every unphased pattern becomes `count` alignment columns of diploid genotypes (IUPAC het codes)"""
    nd = pack.n // 2
    with open(path, "w") as f:
        f.write(f"{nloci}\n\n")
        for g in range(nloci):
            o0, o1 = int(pack.pattern_offsets[g]), int(pack.pattern_offsets[g + 1])
            cols = [[] for _ in range(nd)]
            for r in range(o0, o1):
                if pack.numPhases[r] == 0:
                    continue
                row, cnt = pack.leafcodes[r], int(pack.counts[r])
                for d in range(nd):
                    a, b = int(row[2 * d]), int(row[2 * d + 1])
                    if a == 4 or b == 4:
                        ch = "N"
                    elif a == b:
                        ch = BASE_CODE[a]
                    else:
                        ch = IUPAC[frozenset((BASE_CODE[a], BASE_CODE[b]))]
                    cols[d].append(ch * cnt)
            seqs = ["".join(c) for c in cols]
            f.write(f"locus{g + 1} {nd} {len(seqs[0])}\n")
            for d in range(nd):
                f.write(f"s{d}\t{seqs[d]}\n")
            f.write("\n")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_reference(config, pack, nloci, iters, small_loci, small_iters, full_loci=0):
    """the REAL reference (oracle/_ref, compiled from its own sources in the build container) timed on the host cores of this
    box, on its OpenMP build (`-n threads`, GPhoCS.c:116-145).
      1. thread sweep on the first `nloci` loci (20 000: a 0.7-GB working set, far beyond L3): ONE start-up, then 1 warm-up +
         `iters` iterations timed one by one at each of {1, 8, 16, 32, all cores}; the median iteration counts.
      2. THE VALUE (round 6: like for like): the same on ALL `full_loci` loci the metric is quoted on, at 1 thread and at the best
         thread count of step 1 -- one start-up, 1 warm-up + `iters` iterations each, median.
      3. secondary: the serial build on `small_loci` loci (the earlier rounds' figure)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import gen_synth
    ref = os.path.join(REPO, "oracle", "_ref", "gphocs_ref")
    ref_omp = os.path.join(REPO, "oracle", "_ref", "gphocs_ref_omp")
    if not os.path.exists(ref):
        return None
    ncores = os.cpu_count() or 1
    out = {"unit": "evals/s", "kind": "reference", "cpu_model": cpu_model(), "host_cores": ncores}

    def sweep(td, tag, L, spec):
        t0 = time.perf_counter()
        write_seq_sample(pack, L, os.path.join(td, tag + ".seq"))
        gen_synth.write_ctl(os.path.join(td, tag + ".ctl"), gen_synth.CONFIGS[config], tag + ".seq", tag + ".trace", L, 12345, 1000, 100000)
        t1 = time.perf_counter()
        r = subprocess.run([ref_omp, "timesweep", tag + ".ctl", str(iters), "1", spec], cwd=td, check=True, capture_output=True,
                           text=True, timeout=1500)
        os.unlink(os.path.join(td, tag + ".seq"))
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        by = {int(x["threads"]): x for x in rows if "threads" in x}
        return by, rows[0].get("startup_seconds", 0.0), t1 - t0, time.perf_counter() - t0

    def table(by):
        return {str(t): {"value": x["evals_per_s"], "seconds": x["seconds"], "iterations": x["iters"],
                         "median_iteration_s": x.get("median_iteration_seconds"), "min_iteration_s": x.get("min_iteration_seconds"),
                         "max_iteration_s": x.get("max_iteration_seconds")} for t, x in sorted(by.items())}
    with tempfile.TemporaryDirectory() as td:
        if os.path.exists(ref_omp) and nloci > 0:
            counts = sorted({t for t in (1, 8, 16, 32, ncores) if t <= ncores})
            # >= 5 timed iterations per thread count, each timed on its own, the MEDIAN is the figure; the all-cores count
            # (never the best: the `omp atomic` accumulations and a static split over few loci per thread) gets 3
            spec = ",".join(f"{t}:3" if t == ncores and t > 64 else str(t) for t in counts)
            by, startup, wr, leg = sweep(td, "b", nloci, spec)
            best = max(by.values(), key=lambda x: x["evals_per_s"])
            part = {"value": best["evals_per_s"], "cores": int(best["threads"]), "loci": nloci,
                    "sample": f"first {nloci} loci of the workload (sequence file, {nloci * 0.035:.0f} MB of per-locus state) on the "
                              f"reference's OpenMP build: one start-up ({startup:.0f} s, untimed), then 1 warm-up + {iters} iterations "
                              f"timed one by one at each of {counts} threads, value = from the MEDIAN iteration; best = "
                              f"{int(best['threads'])} threads ({best['seconds']:.1f} s); whole leg {leg:.0f} s",
                    "iters_per_s_at_sample": best["iters_per_s"], "statistic": "median of the timed iterations", "by_threads": table(by)}
            out.update(part)
            if full_loci > nloci:
                # the metric's own size: 1 thread and the best count of the sweep above (at most 16 + 7 iterations of ~5 M evaluations)
                tb = int(best["threads"])
                byf, startup_f, wr_f, leg_f = sweep(td, "f", full_loci, "1," + str(tb) if tb != 1 else "1")
                bestf = max(byf.values(), key=lambda x: x["evals_per_s"])
                out.update({"value": bestf["evals_per_s"], "cores": int(bestf["threads"]), "loci": full_loci,
                            "sample": f"ALL {full_loci} loci of the workload ({full_loci * 0.035 / 1000:.1f} GB of per-locus state) on the "
                                      f"reference's OpenMP build: sequence file written in {wr_f:.0f} s, one start-up ({startup_f:.0f} s, "
                                      f"untimed), then 1 warm-up + {iters} iterations timed one by one at 1 thread and at {tb} threads "
                                      f"(the best count of the {nloci}-locus thread sweep, `thread_sweep`); value = from the MEDIAN "
                                      f"iteration of the better one ({int(bestf['threads'])} threads, {bestf['seconds']:.1f} s); whole leg "
                                      f"{leg_f:.0f} s",
                            "iters_per_s_at_sample": bestf["iters_per_s"], "by_threads": table(byf), "thread_sweep": part})
        write_seq_sample(pack, small_loci, os.path.join(td, "s.seq"))
        gen_synth.write_ctl(os.path.join(td, "s.ctl"), gen_synth.CONFIGS[config], "s.seq", "s.trace", small_loci, 12345,
                            1000, 100000)
        r = json.loads(subprocess.run([ref, "time", "s.ctl", str(small_iters), "2"], cwd=td, check=True,
                                      capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
        small = {"value": r["evals_per_s"], "cores": 1, "kind": "reference",
                 "sample": f"SECONDARY (not like for like: {small_loci * 0.035:.0f}-MB working set): first {small_loci} loci, "
                           f"{small_iters} iterations after 2 warm-up ({r['seconds']:.1f} s), serial reference build"}
        if "value" not in out:
            out.update({"value": small["value"], "cores": 1, "sample": small["sample"]})
        out["serial_small_sample"] = small
    return out


def cpu_baseline(G, pack, nloci, iters):
    """oracle restatement, single thread, bounded sample of the same workload"""
    from gphocs_amd_pkg import synth
    exe = os.path.join(REPO, "oracle", "gphocs_oracle")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True, capture_output=True)
    sub = G.Pack()
    sub.__dict__.update(pack.__dict__)
    sub.L = sub.numLoci = nloci
    o1 = int(pack.pattern_offsets[nloci])
    sub.pattern_offsets = pack.pattern_offsets[:nloci + 1]
    sub.leafcodes, sub.numPhases, sub.counts = pack.leafcodes[:o1], pack.numPhases[:o1], pack.counts[:o1]
    sub.mutRates = np.ones(nloci)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "sample.gpk")
        synth.write_pack(sub, pth)
        out = subprocess.run([exe, "time", pth, str(iters), "2"], check=True, capture_output=True, text=True,
                             timeout=600).stdout
    r = json.loads(out)
    return {"value": r["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"first {nloci} loci of the workload, {iters} iterations after 2 warm-up "
                      f"({r['seconds']:.1f} s); {r['iters_per_s'] * nloci:.0f} locus-iterations/s",
            "iters_per_s_at_sample": r["iters_per_s"]}


def shard_of(L_total, rank, world):
    """contiguous block of ceil(L/world) loci (OpenMP static scheduling of the reference, MultiCoreUtils.h:8)"""
    per = (L_total + world - 1) // world
    return min(rank * per, L_total), min((rank + 1) * per, L_total)


def build_shard(G, config, L_total, begin, end, mut_scale, seed0, cache_dir):
    """loci [begin, end) of the L_total-locus synthetic data set: the 4000-locus generator chunks that overlap the
    block are generated (or read from the cache) and sliced -- every rank sees the same data set whatever the rank count"""
    chunk = 4000
    c0, c1 = begin // chunk, (end - 1) // chunk
    lo = c0 * chunk
    hi = min((c1 + 1) * chunk, L_total)
    pk = build_workload(G, config, hi - lo, mut_scale, seed0, cache_dir, begin=lo, L_total=L_total)
    a, b = begin - lo, end - lo
    o0, o1 = int(pk.pattern_offsets[a]), int(pk.pattern_offsets[b])
    pk.pattern_offsets = (pk.pattern_offsets[a:b + 1] - o0).astype(np.int64)
    pk.leafcodes, pk.numPhases, pk.counts = pk.leafcodes[o0:o1], pk.numPhases[o0:o1], pk.counts[o0:o1]
    pk.L = pk.numLoci = end - begin
    pk.mutRates = np.ones(pk.L)
    pk.global_L, pk.global_begin = L_total, begin
    return pk


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


# Issue cost of one wave-instruction on its SIMD, by class, MEASURED on the MI355X with 8 wavefronts per SIMD
# (tools/probe/class_probe.cpp -> profiles/r04_class_probe.txt; cycles at 2.4 GHz).  A CU has four SIMD-32s: a pure
# vector-register 32-bit instruction takes 2.4 cycles, but everything fp64, and EVERY vector instruction that reads or
# writes a scalar register (lane reads / writes, v_mov from an SGPR, compares into a lane mask, v_readfirstlane) holds the
# SIMD 4.2 cycles; the scalar unit is one per CU (1 instruction per cycle = 4.35 cycles of each SIMD's share), the LDS pipe
# takes one DS instruction per 4.1 cycles per CU (16.4 per SIMD share).
CLASS_CYCLES = {"f64": 4.27, "trans_f64": 16.2, "int32_low": 2.5, "int32_high": 4.27, "int64": 4.27, "cvt": 4.22,
                "other_low": 2.4, "other_high": 4.25, "salu": 4.35, "lds": 16.4}


def issue_roofline(tj, sweep_ms, loci, source):
    """Pipe occupancy of the sweep kernel from committed counter passes (profiles/traffic_k_sweep.json): instructions per
    wavefront by class x measured issue cost / the SIMD-cycles one locus has (4 SIMDs per CU x the CU's cycles per locus at
    the nominal 2.4 GHz, which is what the kernel runs at: 2.34-2.41 GHz read inside k_sweep by a probe build).  The
    counters cannot tell a register-to-register move (2.4 cycles) from a lane read (4.25), so the vector pipe gets a
    range: `other` priced low and high."""
    cyc_cu = sweep_ms * 1e-3 * 2.4e9 / (loci / 256.0)
    budget = 4.0 * cyc_cu
    valu, salu, lds = tj["valu_per_wave"], tj["salu_per_wave"], tj.get("lds_per_wave") or 0.0
    C = CLASS_CYCLES
    mix = tj.get("valu_mix_per_wave")
    if mix:
        f64 = mix["add_f64"] + mix["mul_f64"] + mix["fma_f64"]
        known = f64 + mix["trans_f64"] + mix["int32"] + mix["int64"] + mix["cvt"]
        other = max(valu - known, 0.0)
        fixed = f64 * C["f64"] + mix["trans_f64"] * C["trans_f64"] + mix["int64"] * C["int64"] + mix["cvt"] * C["cvt"]
        lo = fixed + mix["int32"] * C["int32_low"] + other * C["other_low"]
        hi = fixed + mix["int32"] * C["int32_high"] + other * C["other_high"]
    else:
        f64, other, lo, hi = None, None, valu * C["other_low"], valu * C["f64"]
    pipes = {"valu_low": lo / budget, "valu_high": hi / budget, "salu": salu * C["salu"] / budget, "lds": lds * C["lds"] / budget}
    busiest = max(("valu", pipes["valu_high"]), ("salu", pipes["salu"]), ("lds", pipes["lds"]), key=lambda kv: kv[1])
    return {"valu_per_wave": valu, "salu_per_wave": salu, "lds_per_wave": lds, "valu_f64_per_wave": f64,
            "valu_moves_lane_ops_compares_selects_per_wave": other,
            "class_cycles": C, "class_cycles_source": "profiles/r04_class_probe.txt (tools/probe/class_probe.cpp, 8 wavefronts per SIMD)",
            "clock_ghz_assumed": 2.4, "clock_ghz_measured_in_kernel": "2.34-2.41 (round-3 probe build, not this run)",
            "cycles_per_locus_per_cu": cyc_cu, "simd_cycles_per_locus": budget,
            "valu_busy_frac": [pipes["valu_low"], pipes["valu_high"]], "salu_busy_frac": pipes["salu"], "lds_busy_frac": pipes["lds"],
            "floor_ms": sweep_ms * max(pipes["valu_low"], pipes["salu"], pipes["lds"]),
            "binds": f"no pipe is saturated; the busiest is {busiest[0]} at {busiest[1]:.2f}: eight in-order wavefronts per SIMD alternate "
                     "between the vector pipe, the CU's one scalar unit and the CU's LDS pipe (a closed queue: time follows the "
                     "total instruction count of a wavefront)",
            # kept for comparison with the round-3 lines (every vector instruction priced at 4 cycles, SALU at 1 per cycle per CU)
            "valu_issue_frac": valu / cyc_cu, "salu_issue_frac": salu / cyc_cu,
            "source": source}


def secondary_roofline(tj, kernels, source):
    """the evaluate kernels of the global proposals against the HBM peak: counter bytes per launch (committed passes,
    profiles/traffic_k_sweep.json: secondary) / this run's HIP-event time per launch / 8 TB/s"""
    out = {}
    for kk, v in (tj.get("secondary") or {}).items():
        ms = (kernels.get(kk[2:]) or {}).get("avg_ms")
        if ms:
            byts = v["fetch_bytes"] + v["write_bytes"]
            out[kk] = {"bound": "hbm", "traffic": byts, "fetch_bytes": v["fetch_bytes"], "write_bytes": v["write_bytes"],
                       "avg_launch_ms": ms, "achieved": byts / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": byts / (ms * 1e-3) / 8e12, "source": source}
    return out or None


def launch_ranks(a):
    """`bench.py --gpus N` with no launcher around it: this process starts the N ranks (one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment -- what torch.distributed.run would set), waits for them and
    relays rank 0's JSON line.  It only counts devices (torch.cuda.device_count(): hipGetDeviceCount on ROCm) and starts
    child processes -- it never execs another program and never launches GPU work itself."""
    n = a.gpus
    if not a.host_emulation:
        import torch
        nd = torch.cuda.device_count()
        if nd < n:
            print(f"bench: --gpus {n} but this node shows {nd} HIP device(s)", file=sys.stderr)
            return 2
    import gphocs_amd as G
    if a.host_emulation:
        sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
        import run_hostemu
        run_hostemu.build_hostemu()             # once, before the ranks ask for it
    else:
        G.build()                               # once, before the ranks race for the build lock
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    import tempfile
    rank0_out = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout goes to a temporary FILE, read at the end: a pipe read only after every rank has exited would
        # block rank 0 as soon as it (or a library under it) wrote more than the pipe holds
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=rank0_out if r == 0 else sys.stderr))
    deadline = time.time() + a.launch_timeout
    bad = None
    while bad is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc is not None and rc != 0:
                bad = (r, rc)
        if time.time() > deadline:
            bad = (-1, "timeout")
        time.sleep(0.05)
    if bad is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                bad = (r, p.returncode)
    if bad is not None:
        # a rank that failed leaves the others waiting in the next exchange: stop exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench: rank {bad[0]} failed ({bad[1]}); the job was stopped", file=sys.stderr)
        return 1
    rank0_out.seek(0)
    out = rank0_out.read()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if not lines:
        print("bench: rank 0 printed no result line", file=sys.stderr)
        return 1
    line = json.loads(lines[-1])
    if line.get("n_gpus") != n:
        print(f"bench: rank 0 reports n_gpus = {line.get('n_gpus')}, expected {n}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loci", type=int, default=100000, help="loci of the data set (strong scaling: sharded over the "
                    "ranks; with --weak: per GPU)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --loci loci on every rank")
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--mut-scale", type=float, default=6.5, help="mutation scale of the synthetic data (P ~ 18)")
    ap.add_argument("--preroll", type=int, default=200, help="untimed iterations before the warm-up (SURVEY 8d: measure "
                    "after 200 iterations from the prior-sampled start)")
    ap.add_argument("--samples-per-log", type=int, default=0, help="checkAll period (0 = the pack's: 100)")
    ap.add_argument("--cpu-loci", type=int, default=20000, help="loci of the CPU baseline's data set (the first ones of the workload)")
    ap.add_argument("--cpu-full-loci", type=int, default=100000,
                    help="the CPU baseline's VALUE is timed on this many loci (all of the workload's by default; 0 = thread sweep only)")
    ap.add_argument("--cpu-iters", type=int, default=5, help="timed iterations per thread count of the CPU baseline (median reported)")
    ap.add_argument("--cpu-small-loci", type=int, default=5000, help="the secondary, cache-friendlier CPU sample")
    ap.add_argument("--cpu-small-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="alternative build of the library (A/B measurements)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "hook", "shm"], help="cross-rank exchange: native RCCL all-gather on "
                    "the engine's stream (default), the torch.distributed hook (a host round trip per reduction), or the "
                    "host shared-memory exchange (ranks that share a device)")
    ap.add_argument("--host-emulation", action="store_true", help="TESTS ONLY: the host build of the engine sources "
                    "(tests/hostemu), gloo, shared-memory exchange -- exercises this script's N-rank path without a GPU; "
                    "the line is not a measurement")
    ap.add_argument("--launch-timeout", type=float, default=3000.0)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))

    # stdout carries exactly ONE line (the JSON): anything native libraries print there (RCCL's version banner,
    # the HIP runtime) is sent to stderr instead; the saved descriptor is used for the result line only
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        print(f"bench: --gpus {a.gpus} but WORLD_SIZE = {world}: the launcher's world size is what runs", file=sys.stderr)
    import torch
    import gphocs_amd as G
    emu = a.host_emulation
    if emu:
        sys.path.insert(0, os.path.join(REPO, "tests", "hostemu"))
        import run_hostemu
        a.lib = run_hostemu.build_hostemu()
        if a.comm == "rccl":
            a.comm = "shm"
    else:
        G.build()
        torch.cuda.set_device(local_rank)
    dev_sync = (lambda: None) if emu else torch.cuda.synchronize
    dist = None
    allreduce = None
    comm = None
    comm_fallback = False
    ncoll = [0]
    force_dist = os.environ.get("GPH_BENCH_FORCE_DIST") == "1"   # exercise the collective path on one GPU
    L_total = a.loci * world if a.weak else a.loci
    begin, end = (rank * a.loci, (rank + 1) * a.loci) if a.weak else shard_of(L_total, rank, world)
    if end <= begin:
        print(f"bench: rank {rank} of {world} gets no loci out of {L_total}", file=sys.stderr)
        sys.exit(2)
    pack = build_shard(G, a.config, L_total, begin, end, a.mut_scale, 20261002 + a.config, os.path.join(REPO, "bench_cache"))
    if a.samples_per_log > 0:
        pack.samplesPerLog = a.samples_per_log
    lib = G.load_library(a.lib) if a.lib else G.load_library(dims=(pack.n, pack.K, pack.B))
    if world > 1 or force_dist:
        import torch.distributed as dist
        # bench bookkeeping only (barrier, max-over-ranks time, the 128-byte RCCL id); nccl == RCCL on ROCm
        dist.init_process_group("gloo" if emu else "nccl", rank=rank, world_size=world)
        dev = torch.device("cpu") if emu else torch.device("cuda", local_rank)
        if a.comm == "shm":
            comm = lib.gph_comm_create_shm(f"/gphocs-bench-{os.environ.get('MASTER_PORT', '0')}".encode(), rank, world)
            if not comm:
                print("bench: the shared-memory communicator could not be created", file=sys.stderr)
                sys.exit(2)
        if a.comm == "rccl":
            # the engine's own communicator: rank 0 makes the id, torch.distributed carries the 128 bytes
            idbuf = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                import ctypes
                raw = (ctypes.c_uint8 * 128)()
                if lib.gph_comm_unique_id(raw) == 0:
                    idbuf.copy_(torch.tensor(list(raw), dtype=torch.uint8))
            dist.broadcast(idbuf, 0)
            idbytes = bytes(idbuf.cpu().tolist())
            if any(idbytes):
                comm = lib.gph_comm_create_rccl(idbytes, rank, world, local_rank)
            # every rank must end up on the same exchange: if ANY rank could not join, all fall back to the hook
            ok = torch.tensor([1 if comm else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm:
                    lib.gph_comm_destroy(comm)
                comm = None
                comm_fallback = True
                if rank == 0:
                    print("bench: the engine's RCCL communicator could not be created on every rank -- "
                          "falling back to the torch.distributed hook", file=sys.stderr)
        if comm is None:
            # <= 1.5-KB payloads through torch.distributed: every rank all-gathers the (sums | mins) vector and
            # reduces the `world` rows itself in rank order
            SLOTS = 192
            hbuf = torch.zeros(SLOTS, dtype=torch.float64)
            hout = torch.zeros(world * SLOTS, dtype=torch.float64)
            if not emu:
                hbuf, hout = hbuf.pin_memory(), hout.pin_memory()
            dbuf = torch.zeros(SLOTS, dtype=torch.float64, device=dev)
            dout = torch.zeros(world * SLOTS, dtype=torch.float64, device=dev)

            def allreduce(sums, mins):
                ns, nm = sums.size, mins.size
                assert ns + nm <= SLOTS
                hb = hbuf.numpy()
                hb[:ns] = sums
                hb[ns:ns + nm] = mins
                dbuf.copy_(hbuf, non_blocking=True)
                dist.all_gather_into_tensor(dout, dbuf)
                hout.copy_(dout, non_blocking=True)
                if not emu:
                    torch.cuda.current_stream().synchronize()
                rows = hout.numpy().reshape(world, SLOTS)
                if ns:
                    acc = rows[0, :ns].copy()
                    for r in range(1, world):
                        acc += rows[r, :ns]
                    sums[:] = acc
                if nm:
                    mins[:] = rows[:, ns:ns + nm].min(axis=0)
                ncoll[0] += 1

    P = np.diff(pack.pattern_offsets)
    s = G.Sampler(pack, lib=lib, device=local_rank, rank=rank, world=world, allreduce=allreduce, comm=comm)
    s.initialize()
    it0 = 0
    # untimed pre-roll: the chain leaves the prior-sampled start; then W warm-up steps; then K timed
    for it in range(a.preroll):
        s.iteration(it)
    it0 = a.preroll
    for it in range(it0, it0 + a.warmup):
        s.iteration(it)
    it0 += a.warmup
    # per-class kernel times come from the untimed pre-roll + warm-up (every launch bracketed by HIP events); in the timed
    # region only the dominant kernel keeps its bracket: 44 event records per iteration cost 0.07 ms of a step
    names = {0: "sweep", 1: "tau_eval", 2: "mix_eval", 4: "check", 5: "tau_finish", 7: "mix_finish", 8: "sync"}
    kern_pre = {}
    for k, nm in names.items():
        st = s.class_stats(k)
        if st["launches"]:
            kern_pre[nm] = {"launches": int(st["launches"]), "avg_ms": st["ms"] / st["launches"], "from": "pre-roll + warm-up"}
    s.counters(reset=True)
    for k in range(16):
        s.class_stats(k, reset=True)
    s.set_timing(1)
    hs0 = s.host_stats()
    acc0 = s.accept_counts()
    if dist:
        dist.barrier()
    dev_sync()
    t0 = time.perf_counter()
    for it in range(it0, it0 + a.steps):
        s.iteration(it)
    dev_sync()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    hs1 = s.host_stats()
    acc1 = s.accept_counts()
    cnt = s.counters()            # summed over all ranks by the engine (the counters ride in the reduced rows)
    sweep = s.class_stats(0)
    evals, tmax = float(cnt["evals"]), dt
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if emu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
    # amortised cost of checkAll (patch.c:2745, every iterations-per-log = 100 iterations): one untimed-region call
    check_ms = None
    if rank == 0 or dist:
        spl = int(pack.samplesPerLog)
        in_window = sum(1 for it in range(it0, it0 + a.steps) if (it + 1) % spl == 0)
        if "check" in kern_pre:
            check_ms = kern_pre["check"]["avg_ms"]
    if rank == 0:
        kern = dict(kern_pre)
        if sweep["launches"]:
            kern["sweep"] = {"launches": int(sweep["launches"]), "avg_ms": sweep["ms"] / sweep["launches"], "from": "timed region"}
        nl = max(sweep["launches"], 1)
        sweep_ms = sweep["ms"] / nl
        # sweep["bytes"] is summed over the ranks (the counters ride in the reduced rows); the kernel time is this rank's
        alg_bytes = sweep["bytes"] / nl / world
        ach = alg_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        L_local = pack.L
        # load-once / store-once bound of a sweep: 2 * (32 N P + 32 E + 20 N) bytes per locus
        N_, E_ = 2 * pack.n - 1, 2 * pack.n + 40 + 3 * pack.B + pack.K + 10
        per_sweep_bytes = float(np.sum(2.0 * (32.0 * N_ * P + 32.0 * E_ + 20.0 * N_)))
        # HBM traffic and instruction counts of the dominant kernel: separate rocprofv3 --pmc passes (FETCH_SIZE x2
        # gfx950 correction + WRITE_SIZE, per launch; SQ_INSTS_* per wavefront) of an EARLIER run of this command,
        # committed under profiles/ -- not measured here.  They are the measurement of ONE build: used only when the
        # file's build id is the loaded library's and the shard has the file's size.
        build_id = lib.gph_build_id().decode()
        traffic, traffic_src, issue, secondary = None, None, None, None
        tf = os.path.join(REPO, "profiles", "traffic_k_sweep.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                compiler_now = lib.gph_build_compiler().decode()
                # (the build id no longer hashes the compiler, so the counters are accepted only for the same
                # sources + flags AND the same compiler as the library they were measured with)
                same_compiler = tj.get("compiler") == compiler_now
                if tj.get("loci") == L_local and tj.get("build_id") == build_id and tj.get("compiler") is not None and not same_compiler:
                    traffic_src = (f"none: profiles/traffic_k_sweep.json was measured with a library built by '{tj.get('compiler')}', "
                                   f"the loaded library was built by '{compiler_now}'")
                elif tj.get("loci") == L_local and tj.get("build_id") == build_id:
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_src = "profiles/traffic_k_sweep.json (%s)" % tj.get("build", "committed rocprofv3 --pmc passes, not this run")
                    if tj.get("valu_per_wave") and sweep_ms > 0:
                        issue = issue_roofline(tj, sweep_ms, L_local, traffic_src)
                    # the second kernel class: the evaluate kernels of the global proposals are the ones that DO sit on the
                    # memory system (counter bytes per launch / this run's HIP-event time / 8 TB/s)
                    secondary = secondary_roofline(tj, kern_pre, traffic_src)
                elif tj.get("loci") == L_local:
                    traffic_src = (f"none: profiles/traffic_k_sweep.json was measured with build {tj.get('build_id')}, "
                                   f"the loaded library is {build_id}")
                else:
                    traffic_src = (f"none: profiles/traffic_k_sweep.json was measured with {tj.get('loci')} loci per launch, "
                                   f"this rank holds {L_local}")
            except Exception:
                pass
        nsync = (hs1["syncs"] - hs0["syncs"]) / a.steps
        ncol = (hs1["collectives"] - hs0["collectives"]) / a.steps if comm else ncoll[0] / max(a.preroll + a.warmup + a.steps, 1)
        comm_kind = lib.gph_comm_kind(comm).decode() if comm else ("torch.distributed hook" if dist else "none")
        comm_world = int(lib.gph_comm_world(comm)) if comm else (world if dist else 1)
        line = {
            "metric": "locus-likelihood evals/sec (+ MCMC iters/sec), 100k loci, 1/2/4/8 MI355X",
            "value": evals / tmax, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": tmax / a.steps * 1e3, "higher_is_better": True, "scaling": "weak" if a.weak else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "mcmc_iters_per_sec": a.steps / tmax,
            "config": {"workload": f"BASELINE configs[3]: {L_total} loci x 1 kb, 8 diploid samples (16 leaves), "
                                   f"5 current pops + 4 migration bands, full MCMC iteration; "
                                   + (f"{a.loci} loci on every rank (weak)" if a.weak else
                                      f"ONE data set sharded over {world} rank(s) (strong)"),
                       "loci_total": L_total, "loci_this_rank": int(L_local), "leaves": int(pack.n), "pops": int(pack.K),
                       "bands": int(pack.B),
                       "mean_phased_patterns": float(P.mean()), "max_phased_patterns": int(P.max()),
                       "data_diversity": f"mutation scale {a.mut_scale} x the prior mean theta (tunes P to ~18 phased patterns per 1-kb locus)",
                       "evals_timed": evals,
                       "evals_per_locus_iter": evals / (L_total * a.steps),
                       "recomputed_nodes_per_eval": cnt["eval_nodes"] / max(cnt["evals"], 1),
                       "algorithmic_bytes_per_eval": cnt["eval_bytes"] / max(cnt["evals"], 1),
                       "accept_counts_timed": [int(y - x) for x, y in zip(acc0, acc1)],
                       "preroll_iterations": a.preroll,
                       "checkall_period": int(pack.samplesPerLog), "checkall_in_timed_window": in_window,
                       "checkall_ms": check_ms,
                       "checkall_amortised_ms_per_iteration": (check_ms / int(pack.samplesPerLog)) if check_ms else None,
                       "host_syncs_per_iteration": nsync,
                       "collectives_per_iteration": ncol if (dist is not None) else 0,
                       "communicator": comm_kind, "communicator_world": comm_world,
                       "decisions": "device-resident (k_global)" if hs1["resident"] else "host",
                       "kernel_launches_per_iteration": (hs1["launches"] - hs0["launches"]) / a.steps,
                       "library_build_id": build_id,
                       "compiler": lib.gph_build_compiler().decode(), "runtime": lib.gph_runtime_version().decode(),
                       "parallelism": f"loci sharded over {world} rank(s), one process per GPU"
                                      + (f", native {comm_kind} exchange of the reduced row" + (" on the engine's stream" if hs1["resident"] else "") if comm else
                                         (", torch.distributed hook" + (" (fallback: the RCCL communicator could not be created)" if comm_fallback else "") if dist else ""))},
            # bound: what BINDS the dominant kernel as measured ("issue": the length of a locus's dependent instruction
            # stream, with the vector pipe, the CU's scalar unit and the LDS pipe each half to three-quarters busy -- the `issue`
            # object; no memory pipe binds); achieved / peak / frac stay SURVEY 8d's accounting against the HBM peak, which is
            # the roofline the path is HBM-bound by construction under (0.33 flop/B)
            "roofline": {"bound": "issue", "bound_by_construction": "hbm",
                         "kernel": "k_sweep (fused UpdateGB_InternalNode+MigrationNode+MigSPR)",
                         "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                         "accounting": "ALGORITHMIC bytes per evaluation (96 R P + 20 N + 8 U + 8, SURVEY 8d) / HIP-event "
                                       "kernel time; mostly L2 / Infinity-Cache hits, see hbm_counter_frac for DRAM",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_counter_frac": (traffic / (sweep_ms * 1e-3) / 8e12) if traffic and sweep_ms > 0 else None,
                         "issue": issue,
                         "secondary": secondary,
                         "per_sweep": {"bytes": per_sweep_bytes, "bound_ms_at_peak": per_sweep_bytes / 8e12 * 1e3,
                                       "frac": per_sweep_bytes / 8e12 * 1e3 / sweep_ms if sweep_ms > 0 else None,
                                       "accounting": "load-once/store-once: 2 (32 N P + 32 E + 20 N) bytes per locus per sweep"},
                         "avg_launch_ms": sweep_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "evals_per_launch": sweep["evals"] / nl / world},
            "kernels": kern,
            "hbm_resident_bytes": s.hbm_bytes(),
        }
        if emu:
            line["data"] = "synthetic; HOST EMULATION of the engine sources (tests only): not a measurement"
        if world == 1 and not a.no_cpu_baseline and not emu:
            try:
                cb = cpu_baseline_reference(a.config, pack, min(a.cpu_loci, L_total), a.cpu_iters,
                                            min(a.cpu_small_loci, L_total), a.cpu_small_iters,
                                            full_loci=min(a.cpu_full_loci, L_total))
                port = cpu_baseline(G, pack, min(a.cpu_small_loci, L_total), a.cpu_small_iters)
                if cb is None:
                    cb = port
                    cb["note"] = "oracle/_ref (the real reference binary) is absent on this box: the restatement was timed"
                else:
                    cb["port_single_thread_small_sample"] = {"value": port["value"], "kind": "port"}
                line["cpu_baseline"] = cb
            except Exception as ex:  # pragma: no cover
                line["cpu_baseline"] = {"error": str(ex)}
    s.close()
    if comm:
        lib.gph_comm_destroy(comm)
    if dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
