#!/usr/bin/env python3
"""Benchmark of the hot path: full MCMC iterations of the per-locus likelihood engine.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

step      = one MCMC iteration of performMCMC's proposal sequence (GPhoCS.c:1476-1821) over every
            locus: fused genealogy sweep (node ages, migration ages, SPR), theta, migration rates,
            tau rubber bands, mixing, event synchronisation -- nothing skipped.
workload  = BASELINE.json configs[3], the one the metric is quoted on: 100k loci x 1 kb, 8 diploid
            samples (16 leaves), 5 current populations + 4 migration bands; synthetic data
            (g-phocs_amd/synth.py), resident in HBM before the timed region.
scaling   = weak: every rank holds --loci loci (default 100k); whole-job value = all ranks' units
            / max-over-ranks time.  Ranks share nothing but the <= 240-byte all-reduce (RCCL) of
            each global proposal.
value     = locus-likelihood evaluations per second (computeLocusDataLikelihood(useOld=1)
            equivalents, counted by the kernels); MCMC iterations/s is reported next to it.
roofline  = dominant kernel (fused genealogy sweep): algorithmic bytes (96*R*P + 20*N + 8*U + 8 per
            evaluation, counted per evaluation by the kernel) / HIP-event duration, vs 8 TB/s HBM.
cpu_baseline = the oracle restatement (bit-identical to the reference on the golden vectors) timed
            single-threaded on a bounded sample (first --cpu-loci loci, a few iterations) on this
            box's host cores -- rank 0, N = 1 only.
"""
import argparse
import json
import multiprocessing as mp
import os
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


def cpu_baseline_reference(config, pack, nloci, iters):
    """the REAL reference (oracle/_ref, compiled from its own sources in the build container) timed on a
    bounded sample of the same workload: serial build on 1 core and its OpenMP build on all cores"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import gen_synth
    ref = os.path.join(REPO, "oracle", "_ref", "gphocs_ref")
    ref_omp = os.path.join(REPO, "oracle", "_ref", "gphocs_ref_omp")
    if not os.path.exists(ref):
        return None
    out = {}
    with tempfile.TemporaryDirectory() as td:
        write_seq_sample(pack, nloci, os.path.join(td, "s.seq"))
        gen_synth.write_ctl(os.path.join(td, "s.ctl"), gen_synth.CONFIGS[config], "s.seq", "s.trace", nloci, 12345,
                            1000, 100000)
        r = json.loads(subprocess.run([ref, "time", "s.ctl", str(iters), "2"], cwd=td, check=True,
                                      capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
        out = {"value": r["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "reference",
               "sample": f"first {nloci} loci of the workload written as a sequence file, {iters} iterations after "
                         f"2 warm-up ({r['seconds']:.1f} s), serial reference build; "
                         f"{r['iters_per_s'] * nloci:.0f} locus-iterations/s",
               "iters_per_s_at_sample": r["iters_per_s"]}
        if os.path.exists(ref_omp):
            nc = min(os.cpu_count() or 1, 8)   # the reference scales ~3x on 8 threads and collapses beyond (atomics)
            env = dict(os.environ, OMP_NUM_THREADS=str(nc))
            try:
                r2 = json.loads(subprocess.run([ref_omp, "time", "s.ctl", str(iters), "2"], cwd=td, check=True, env=env,
                                               capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1])
                out["openmp_all_cores"] = {"value": r2["evals_per_s"], "cores": nc, "seconds": r2["seconds"]}
            except Exception as ex:  # pragma: no cover
                out["openmp_all_cores"] = {"error": str(ex)}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loci", type=int, default=100000, help="loci per GPU")
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--mut-scale", type=float, default=6.5, help="mutation scale of the synthetic data (P ~ 18)")
    ap.add_argument("--cpu-loci", type=int, default=5000)
    ap.add_argument("--cpu-iters", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): anything native libraries print there (RCCL's version banner,
    # the HIP runtime) is sent to stderr instead; the saved descriptor is used for the result line only
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import gphocs_amd as G
    G.build()
    torch.cuda.set_device(local_rank)
    dist = None
    allreduce = None
    force_dist = os.environ.get("GPH_BENCH_FORCE_DIST") == "1"   # exercise the collective path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)   # nccl == RCCL on ROCm
        dev = torch.device("cuda", local_rank)

        # <= 240-byte payloads.  One collective per reduction point: every rank all-gathers the (sums | mins)
        # vector over RCCL and reduces the `world` rows itself in rank order -- one launch instead of an
        # all-reduce(SUM) plus an all-reduce(MIN), and every rank adds in the same order (identical bits)
        SLOTS = 64
        hbuf = torch.zeros(SLOTS, dtype=torch.float64).pin_memory()
        dbuf = torch.zeros(SLOTS, dtype=torch.float64, device=dev)
        dout = torch.zeros(world * SLOTS, dtype=torch.float64, device=dev)
        hout = torch.zeros(world * SLOTS, dtype=torch.float64).pin_memory()
        ncoll = [0]

        def allreduce(sums, mins):
            ns, nm = sums.size, mins.size
            assert ns + nm <= SLOTS
            hb = hbuf.numpy()
            hb[:ns] = sums
            hb[ns:ns + nm] = mins
            dbuf.copy_(hbuf, non_blocking=True)
            dist.all_gather_into_tensor(dout, dbuf)
            hout.copy_(dout, non_blocking=True)
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

    L_total = a.loci * world
    # weak scaling: every rank generates (and holds) only its own a.loci loci of the L_total-locus data set
    pack = build_workload(G, a.config, a.loci, a.mut_scale, 20261002 + a.config,
                          os.path.join(REPO, "bench_cache"), begin=rank * a.loci, L_total=L_total)
    P = np.diff(pack.pattern_offsets)
    s = G.Sampler(pack, device=local_rank, rank=rank, world=world, allreduce=allreduce)
    s.initialize()
    for it in range(a.warmup):
        s.iteration(it)
    s.counters(reset=True)
    for k in range(16):
        s.class_stats(k, reset=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.warmup, a.warmup + a.steps):
        s.iteration(it)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    cnt = s.counters()
    sweep = s.class_stats(0)
    evals, tmax = float(cnt["evals"]), dt
    if dist:
        t = torch.tensor([evals], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        evals = float(t.item())
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
    if rank == 0:
        kern = {}
        names = {0: "sweep", 1: "tau_eval", 2: "mix_eval", 4: "check", 5: "tau_commit", 6: "tau_revert",
                 7: "mix_commit", 8: "sync"}
        for k, nm in names.items():
            st = s.class_stats(k)
            if st["launches"]:
                kern[nm] = {"launches": int(st["launches"]), "avg_ms": st["ms"] / st["launches"]}
        ach = (sweep["bytes"] / max(sweep["launches"], 1)) / (sweep["ms"] / max(sweep["launches"], 1) * 1e-3) / 1e9 \
            if sweep["ms"] > 0 else 0.0
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2
        # gfx950 correction + WRITE_SIZE, per launch) committed under profiles/; null if absent
        traffic = None
        tf = os.path.join(REPO, "profiles", "traffic_k_sweep.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("loci") == a.loci:
                    traffic = tj["hbm_bytes_per_launch"]
            except Exception:
                pass
        line = {
            "metric": "locus-likelihood evals/sec (+ MCMC iters/sec), 100k loci per MI355X",
            "value": evals / tmax, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": tmax / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "mcmc_iters_per_sec": a.steps / tmax,
            "config": {"workload": f"BASELINE configs[3]: {a.loci} loci/GPU x 1 kb, 8 diploid samples (16 leaves), "
                                   f"5 current pops + 4 migration bands, full MCMC iteration",
                       "loci_total": L_total, "leaves": int(pack.n), "pops": int(pack.K), "bands": int(pack.B),
                       "mean_phased_patterns": float(P.mean()), "max_phased_patterns": int(P.max()),
                       "evals_per_locus_iter": evals / (L_total * a.steps),
                       "recomputed_nodes_per_eval": cnt["eval_nodes"] / max(cnt["evals"], 1),
                       "algorithmic_bytes_per_eval": cnt["eval_bytes"] / max(cnt["evals"], 1),
                       "parallelism": f"loci sharded over {world} rank(s), one process per GPU"
                                      + (f", {ncoll[0] / max(a.warmup + a.steps, 1):.1f} RCCL all-gathers (<= 512 B) per "
                                         f"iteration" if dist else "")},
            "roofline": {"bound": "hbm", "kernel": "k_sweep (fused UpdateGB_InternalNode+MigrationNode+MigSPR)",
                         "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                         "avg_launch_ms": sweep["ms"] / max(sweep["launches"], 1),
                         "algorithmic_bytes_per_launch": sweep["bytes"] / max(sweep["launches"], 1),
                         "evals_per_launch": sweep["evals"] / max(sweep["launches"], 1)},
            "kernels": kern,
            "hbm_resident_bytes": s.hbm_bytes(),
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                cb = cpu_baseline_reference(a.config, pack, min(a.cpu_loci, L_total), a.cpu_iters)
                port = cpu_baseline(G, pack, min(a.cpu_loci, L_total), a.cpu_iters)
                if cb is None:
                    cb = port
                else:
                    cb["port_single_thread"] = {"value": port["value"], "kind": "port"}
                line["cpu_baseline"] = cb
            except Exception as ex:  # pragma: no cover
                line["cpu_baseline"] = {"error": str(ex)}
    s.close()
    if dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
