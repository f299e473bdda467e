#!/usr/bin/env python3
"""Copy the outputs of tools/profile_round.sh <tag> from gpurun_out/ into profiles/ (the committed evidence):
bench line (with the measured HBM traffic filled in), kernel stats, counter summary, traffic_k_sweep.json.
   python3 tools/profile_publish.py <tag> [round-prefix, default r03]
traffic_k_sweep.json carries the build id of the library the counters were taken with (bench line: library_build_id);
bench.py uses the file only for that build."""
import json
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
d = json.load(open(f"gpurun_out/pmc_{tag}.json"))
k = [x for x in d if "k_sweep" in x][0]
s = d[k]
nd = s["FETCH_SIZE"]["dispatches"]
fetch = s["FETCH_SIZE"]["sum"] * 1024 * 2 / nd
write = s["WRITE_SIZE"]["sum"] * 1024 / s["WRITE_SIZE"]["dispatches"]
n = s["SQ_WAVES"]["sum"]
print("k_sweep dispatches", nd, "fetch GB %.2f write GB %.2f" % (fetch / 1e9, write / 1e9))
for c in sorted(s):
    print("%-22s per wave %12.1f" % (c, s[c]["sum"] / n))
b = json.load(open(f"gpurun_out/bench_{tag}.json"))
build_id = b["config"].get("library_build_id")
json.dump({"kernel": "k_sweep", "loci": 100000, "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch,
           "write_bytes": write, "build": f"{rnd} {tag}, library variant s", "build_id": build_id,
           "compiler": b["config"].get("compiler"),      # the counters are the measurement of ONE compiler's code (bench.py checks both)
           "valu_per_wave": s["SQ_INSTS_VALU"]["sum"] / n, "salu_per_wave": s["SQ_INSTS_SALU"]["sum"] / n,
           "lds_per_wave": s["SQ_INSTS_LDS"]["sum"] / n, "smem_per_wave": s["SQ_INSTS_SMEM"]["sum"] / n,
           "wave_cycles_per_wave": s["SQ_WAVE_CYCLES"]["sum"] / n,
           "valu_mix_per_wave": ({kk: s["SQ_INSTS_VALU_" + kk.upper()]["sum"] / n for kk in
                                  ("add_f64", "mul_f64", "fma_f64", "trans_f64", "int32", "int64", "cvt")}
                                 if "SQ_INSTS_VALU_ADD_F64" in s and "SQ_INSTS_VALU_INT32" in s else None),
           "branches_per_wave": s["SQ_INSTS_BRANCH"]["sum"] / n if "SQ_INSTS_BRANCH" in s else None,
           # the second kernel class (evaluate kernels of the global proposals): counter bytes per launch, same corrections
           "secondary": {kk: {"fetch_bytes": d[kn]["FETCH_SIZE"]["sum"] * 1024 * 2 / d[kn]["FETCH_SIZE"]["dispatches"],
                              "write_bytes": d[kn]["WRITE_SIZE"]["sum"] * 1024 / d[kn]["WRITE_SIZE"]["dispatches"],
                              "launches_measured": d[kn]["FETCH_SIZE"]["dispatches"]}
                         for kk, kn in (("k_tau_eval", next((x for x in d if "k_tau_eval" in x), None)),
                                        ("k_mix_eval", next((x for x in d if "k_mix_eval" in x), None))) if kn},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_collect.sh): "
                     "FETCH_SIZE(KB)*1024*2 (gfx950 correction) + WRITE_SIZE(KB)*1024 per k_sweep dispatch (one "
                     f"dispatch per sweep), averaged over the {nd} dispatches of a 200-iteration pre-roll + 4 iterations"},
          open("profiles/traffic_k_sweep.json", "w"), indent=1)
shutil.copy(f"gpurun_out/pmc_{tag}.json", f"profiles/{rnd}_pmc_{tag}.json")
shutil.copy(f"gpurun_out/kstats_{tag}/k_kernel_stats.csv", f"profiles/{rnd}_bench_kernel_stats_{tag}.csv")
b["roofline"]["traffic"] = fetch + write
b["roofline"]["traffic_source"] = f"profiles/traffic_k_sweep.json ({rnd} {tag}: rocprofv3 --pmc passes of the same command on the same box)"
b["roofline"]["hbm_counter_frac"] = (fetch + write) / (b["roofline"]["avg_launch_ms"] * 1e-3) / 8e12
# the bench ran before this file existed for its build: the same arithmetic as bench.py
sys.path.insert(0, ".")
import bench as _bench
b["roofline"]["issue"] = _bench.issue_roofline(json.load(open("profiles/traffic_k_sweep.json")), b["roofline"]["avg_launch_ms"], 100000,
                                               b["roofline"]["traffic_source"])
b["roofline"]["secondary"] = _bench.secondary_roofline(json.load(open("profiles/traffic_k_sweep.json")), b["kernels"], b["roofline"]["traffic_source"])
json.dump(b, open(f"profiles/{rnd}_bench_{tag}.json", "w"), indent=1)
try:
    shutil.copy(f"gpurun_out/bench_{tag}_12500.json", f"profiles/{rnd}_bench_{tag}_12500loci.json")
except OSError:
    pass
print(json.dumps({k2: b[k2] for k2 in ("value", "ms_per_step", "mcmc_iters_per_sec")}), b["roofline"]["frac"], b.get("cpu_baseline", {}).get("value"))

# BASELINE configs[4] at its stated size (tools/profile_config5.sh), when its outputs are there: ONE run per file.  The data
# set has loci with more than 64 phased patterns, so every kernel class is TWO dispatches per launch point (the few
# pattern-rich loci on a side stream next to everything else): the per-dispatch average of the kernel-stats CSV is not a
# per-sweep time, and the summed durations overlap -- the per-sweep wall time comes from the trace (profile_config5.sh).
import csv
import os
if os.path.exists("gpurun_out/config5_bench.json") and os.path.exists("gpurun_out/config5_kstats/k_kernel_stats.csv"):
    cb = json.load(open("gpurun_out/config5_bench.json"))
    rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open("gpurun_out/config5_kstats/k_kernel_stats.csv"))}
    iters = cb.get("preroll_iterations", 0) + 15
    sw = rows["k_sweep"]
    cb["rocprofv3_kernel_stats"] = {
        "k_sweep_dispatches": int(sw["Calls"]), "iterations_of_the_run": iters, "dispatches_per_sweep": int(sw["Calls"]) / iters,
        "k_sweep_ms_per_sweep": float(sw["TotalDurationNs"]) / iters / 1e6,
        "k_sweep_max_dispatch_ms": float(sw["MaxNs"]) / 1e6, "k_sweep_min_dispatch_ms": float(sw["MinNs"]) / 1e6,
        "note": "two dispatches per sweep (loci with > 64 phased patterns on a side stream, next to the rest): the summed "
                "durations overlap; compare k_sweep_wall_ms_per_sweep (first start to last end of a sweep's two dispatches, "
                "from the same trace) with sweep_ms"}
    if os.path.exists("gpurun_out/config5_sweep_wall.json"):
        cb["rocprofv3_kernel_stats"].update(json.load(open("gpurun_out/config5_sweep_wall.json")))
    json.dump(cb, open(f"profiles/{rnd}_config5_bench.json", "w"), indent=1)
    shutil.copy("gpurun_out/config5_kstats/k_kernel_stats.csv", f"profiles/{rnd}_config5_kernel_stats.csv")
    if os.path.exists("gpurun_out/config5_pmc.json"):
        shutil.copy("gpurun_out/config5_pmc.json", f"profiles/{rnd}_config5_pmc.json")
    print("config5:", cb["evals_per_s"], cb["sweep_ms"], cb["sweep_roofline_frac"], cb["rocprofv3_kernel_stats"])
