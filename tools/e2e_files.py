#!/usr/bin/env python3
"""End to end at BASELINE scale through the UNCHANGED file interfaces: the bench data set
(100 000 loci x 16 leaves, BASELINE configs[3]) written as a sequence file in the reference's input format + a control
file, then
  golden  (build container only: needs oracle/_ref, the real reference compiled from /root/reference/src)
          the REAL binary (`gphocs_ref main <ctl>`) runs ITERS iterations on those files; its trace file becomes
          tests/golden/e2e100k.trace (a few KB), its start-up seconds at 20k / 50k / 100k loci (`gphocs_ref ingest`:
          readControlFile + readSeqFile + processAlignments, AlignmentProcessor.c:468-983, 1490-1507) go to
          tests/golden/e2e100k.ref.json
  run     (GPU box) `G-PhoCS-hip <ctl>` on the same files (regenerated there from the same seeds); the trace file must
          equal the golden rows (GPhoCS.c:1763-1769: to the printed precision); `gph_loci_read` seconds at 20k / 50k /
          100k loci and the whole program's wall time are written to gpurun_out/e2e_100k.json
usage:  python tools/e2e_files.py golden|run [loci] [iterations]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
GOLDEN = os.path.join(REPO, "tests", "golden")
CONFIG, MUT, SEED0 = 4, 6.5, 20261006      # bench.py's default workload


def files(td, L, iters, sizes):
    """sequence files of the first n loci of the L-locus data set for every n in `sizes` (+ control files)"""
    import gphocs_amd as G
    import bench
    import gen_synth
    pack = bench.build_workload(G, CONFIG, L, MUT, SEED0, os.path.join(REPO, "bench_cache"))
    for n in sizes:
        t0 = time.perf_counter()
        bench.write_seq_sample(pack, n, os.path.join(td, f"e2e_{n}.seq"))
        gen_synth.write_ctl(os.path.join(td, f"e2e_{n}.ctl"), gen_synth.CONFIGS[CONFIG], f"e2e_{n}.seq", f"e2e_{n}.trace", n,
                            12345, iters, 100)
        print(f"wrote e2e_{n}.seq ({os.path.getsize(os.path.join(td, f'e2e_{n}.seq')) / 1e6:.0f} MB) in "
              f"{time.perf_counter() - t0:.0f} s", flush=True)
    return pack


def golden(L, iters):
    ref = os.path.join(REPO, "oracle", "_ref", "gphocs_ref")
    assert os.path.exists(ref), "golden mode needs oracle/_ref (build container)"
    sizes = sorted({20000, 50000, L} if L >= 50000 else {L})
    out = {"loci": L, "iterations": iters, "reference_startup_seconds": {}, "host": os.uname().nodename,
           "cpu_model": __import__("bench").cpu_model()}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        files(td, L, iters, sizes)
        for n in sizes:
            t0 = time.perf_counter()
            r = subprocess.run([ref, "ingest", f"e2e_{n}.ctl"], cwd=td, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["reference_startup_seconds"][str(n)] = json.loads(js[-1]) if js else {"wall": time.perf_counter() - t0}
            out["reference_startup_seconds"][str(n)]["wall"] = time.perf_counter() - t0
            print("reference ingest", n, out["reference_startup_seconds"][str(n)], flush=True)
        t0 = time.perf_counter()
        r = subprocess.run([ref, "main", f"e2e_{L}.ctl"], cwd=td, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out["reference_main_seconds"] = time.perf_counter() - t0
        tr = open(os.path.join(td, f"e2e_{L}.trace")).read()
        assert len(tr.splitlines()) == iters + 1, len(tr.splitlines())
        tag = f"e2e{L // 1000}k"
        open(os.path.join(GOLDEN, tag + ".trace"), "w").write(tr)
        open(os.path.join(GOLDEN, tag + ".ctl"), "w").write(open(os.path.join(td, f"e2e_{L}.ctl")).read())
        json.dump(out, open(os.path.join(GOLDEN, tag + ".ref.json"), "w"), indent=1)
    print(json.dumps(out))


def compare_trace(want_path, got_path):
    """the parameter columns (theta, tau, migration rates: %8.5f of values that do not depend on a sum over loci) must be
    the real binary's characters; the two log-likelihood columns (sums over all loci: serial order upstream, a fixed tree
    on the device) agree within north_star's 1e-10 relative (+ half a unit of the last printed digit)"""
    want = open(want_path).read().splitlines()
    got = open(got_path).read().splitlines()
    assert want[0] == got[0], "trace header differs"
    assert len(want) == len(got), (len(want), len(got))
    ndiff, worst = 0, 0.0
    for w, g in zip(want[1:], got[1:]):
        if w == g:
            continue
        ndiff += 1
        ws, gs = w.split(), g.split()
        assert len(ws) == len(gs) and ws[:-2] == gs[:-2], ("a parameter column differs", w, g)
        for x, y in zip(ws[-2:], gs[-2:]):
            x, y = float(x), float(y)
            assert abs(x - y) <= 1e-10 * abs(x) + 1e-6, ("log-likelihood column beyond 1e-10 relative", w, g)
            worst = max(worst, abs(x - y) / max(abs(x), 1e-300))
    return len(want) - 1, ndiff, worst


def run(L, iters):
    import ctypes as C
    import gphocs_amd as G
    tag = f"e2e{L // 1000}k"
    want = os.path.join(GOLDEN, tag + ".trace")
    assert os.path.exists(want), f"{want} missing: run `python tools/e2e_files.py golden {L} {iters}` in the build container"
    ref = json.load(open(os.path.join(GOLDEN, tag + ".ref.json")))
    sizes = sorted({20000, 50000, L} if L >= 50000 else {L})
    out = {"loci": L, "iterations": iters, "gph_loci_read_seconds": {}, "reference_startup_seconds_build_container":
           {k: v.get("wall") for k, v in ref["reference_startup_seconds"].items()},
           "reference_cpu_model_build_container": ref.get("cpu_model"), "host_cores": os.cpu_count(),
           "cpu_model": __import__("bench").cpu_model()}
    exe = os.path.join(REPO, "g-phocs_amd", "G-PhoCS-hip")
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        files(td, L, iters, sizes)
        lib = G.load_library(dims=(16, 9, 4))
        for n in sizes:
            for threads in (1, 0):       # 0 = all host threads
                ctl, loci, err = C.c_void_p(), C.c_void_p(), C.create_string_buffer(512)
                assert lib.gph_control_read(os.path.join(td, f"e2e_{n}.ctl").encode(), None, C.byref(ctl)) == 0, "control file"
                t0 = time.perf_counter()
                rc = lib.gph_loci_read(ctl, os.path.join(td, f"e2e_{n}.seq").encode(), threads, C.byref(loci), err, 512)
                dt = time.perf_counter() - t0
                assert rc == 0, err.value
                out["gph_loci_read_seconds"].setdefault(str(n), {})["1 thread" if threads == 1 else "all threads"] = dt
                lib.gph_loci_free(loci)
                lib.gph_control_free(ctl)
                print(f"gph_loci_read {n} loci, threads={threads or 'all'}: {dt:.2f} s", flush=True)
        t0 = time.perf_counter()
        r = subprocess.run([exe, f"e2e_{L}.ctl"], cwd=td, capture_output=True, text=True, timeout=1800)
        out["program_wall_seconds"] = time.perf_counter() - t0
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        rows, ndiff, worst = compare_trace(want, os.path.join(td, f"e2e_{L}.trace"))
        out["trace_rows_compared"], out["trace_rows_not_character_identical"] = rows, ndiff
        out["parameter_columns"] = "character-identical in every row"
        out["worst_relative_difference_of_a_log_likelihood_column"] = worst
        out["reference_main_seconds_build_container"] = ref.get("reference_main_seconds")
        out["library_build_id"] = lib.gph_build_id().decode()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", f"e2e_{L // 1000}k.json"), "w"), indent=1)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    mode = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    {"golden": golden, "run": run}[mode](L, iters)
