#!/usr/bin/env python3
"""Copy the outputs of tools/profile_round.sh <tag> from gpurun_out/ into profiles/ (the committed evidence):
bench line (with the measured HBM traffic filled in), kernel stats, counter summary, traffic_k_sweep.json."""
import json
import shutil
import sys

tag = sys.argv[1]
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
json.dump({"kernel": "k_sweep", "loci": 100000, "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch,
           "write_bytes": write,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_collect.sh): "
                     "FETCH_SIZE(KB)*1024*2 (gfx950 correction) + WRITE_SIZE(KB)*1024 per k_sweep dispatch (one "
                     f"dispatch per sweep), averaged over {nd} dispatches; {tag} build (variant s)"},
          open("profiles/traffic_k_sweep.json", "w"), indent=1)
shutil.copy(f"gpurun_out/pmc_{tag}.json", f"profiles/r01_pmc_{tag}.json")
shutil.copy(f"gpurun_out/kstats_{tag}/k_kernel_stats.csv", f"profiles/r01_bench_kernel_stats_{tag}.csv")
b = json.load(open(f"gpurun_out/bench_{tag}.json"))
b["roofline"]["traffic"] = fetch + write
json.dump(b, open(f"profiles/r01_bench_{tag}.json", "w"), indent=1)
print(json.dumps({k2: b[k2] for k2 in ("value", "ms_per_step", "mcmc_iters_per_sec")}), b["roofline"]["frac"], b["cpu_baseline"]["value"], b["cpu_baseline"].get("openmp_all_cores"))
