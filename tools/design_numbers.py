#!/usr/bin/env python3
"""Fills the figures of DESIGN.md section 8 (and the first row of section 10) in from the committed profile set
(tools/design_section8.template.md holds that text with placeholders; DESIGN.md carries the markers):
   python3 tools/design_numbers.py <tag, e.g. v1>"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(REPO, "profiles")
tag = sys.argv[1]
b = json.load(open(f"{P}/r06_bench_{tag}.json"))
s12 = json.load(open(f"{P}/r06_bench_{tag}_12500loci.json"))
t = json.load(open(f"{P}/traffic_k_sweep.json"))
r = b["roofline"]
ks = next(row for row in csv.DictReader(open(f"{P}/r06_bench_kernel_stats_{tag}.csv")) if row["Name"].startswith("k_sweep"))
c2, c3 = json.load(open(f"{P}/r06_config2_bench.json")), json.load(open(f"{P}/r06_config3_bench.json"))
c5 = json.load(open(f"{P}/r06_config5_bench.json"))
e2e = json.load(open(f"{P}/r06_e2e_100k.json")) if os.path.exists(f"{P}/r06_e2e_100k.json") else json.load(open(f"{P}/r05_e2e_100k.json"))
sec = r["secondary"]
M = lambda v: f"{v / 1e6:.1f} M"
rep = {
    "V_EVALS": M(b["value"]), "V_ITS": f"{b['mcmc_iters_per_sec']:.1f}", "V_MS": f"{b['ms_per_step']:.2f}",
    "V_SWEEPK": f"{float(ks['AverageNs']) / 1e6:.2f}", "V_SWEEP": f"{r['avg_launch_ms']:.2f}", "V_FRAC": f"{r['frac']:.3f}",
    "V_CFRAC": f"{r['hbm_counter_frac']:.3f}", "VTAG": tag,
    "V_FETCH": f"{t['fetch_bytes'] / 1e9:.2f}", "V_WRITE": f"{t['write_bytes'] / 1e9:.2f}", "V_TRAFFIC": f"{t['hbm_bytes_per_launch'] / 1e9:.2f}",
    "V_VALU": f"{t['valu_per_wave'] / 1e3:.1f} k", "V_SALU": f"{t['salu_per_wave'] / 1e3:.1f} k", "V_LDS": f"{t['lds_per_wave'] / 1e3:.1f} k",
    "V_REST": f"{b['ms_per_step'] - r['avg_launch_ms']:.2f}",
    "T_TAUF": f"{sec['k_tau_eval']['frac']:.2f}", "T_TAU": f"{sec['k_tau_eval']['avg_launch_ms']:.3f}",
    "T_MIXF": f"{sec['k_mix_eval']['frac']:.2f}", "T_MIX": f"{sec['k_mix_eval']['avg_launch_ms']:.3f}",
    "S_EVALS": M(s12["value"]), "S_MS": f"{s12['ms_per_step']:.2f}", "S_SWEEP": f"{s12['roofline']['avg_launch_ms']:.2f}",
    "E_PROG": f"{e2e['program_wall_seconds']:.1f}", "B_BEST": f"{b['cpu_baseline']['value'] / 1e6:.2f}",
    "B_THR": str(b["cpu_baseline"]["cores"]), "B_ONE": f"{b['cpu_baseline']['by_threads']['1']['value'] / 1e6:.2f}",
    "T_TAUB": f"{(t['secondary']['k_tau_eval']['fetch_bytes'] + t['secondary']['k_tau_eval']['write_bytes']) / 1e9:.2f}",
    "T_MIXB": f"{(t['secondary']['k_mix_eval']['fetch_bytes'] + t['secondary']['k_mix_eval']['write_bytes']) / 1e9:.2f}",
    "SOAK_EVALS": f"{json.load(open(f'{P}/r06_soak.json'))['legs'][0]['evals_per_s'] / 1e6:.1f}",
    "AB_LINE": open(f"{P}/r06_ab_sparse_shadow.txt").read().split("SUMMARY: ")[1].strip() if os.path.exists(f"{P}/r06_ab_sparse_shadow.txt") else "(pending)",
}
for k, c in (("C2", c2), ("C3", c3), ("C5", c5)):
    rep[k + "_EVALS"] = M(c["evals_per_s"]); rep[k + "_ITS"] = f"{c['iters_per_s']:.1f}"; rep[k + "_MS"] = f"{c['ms_per_iteration']:.2f}"
    rep[k + "_SWEEP"] = f"{c['sweep_ms']:.2f}"; rep[k + "_FRAC"] = f"{c['sweep_roofline_frac']:.3f}"
    rep[k + "_CFRAC"] = f"{c['sweep_hbm_counter_frac']:.3f}" if "sweep_hbm_counter_frac" in c else "—"
tpl = open(os.path.join(REPO, "tools", "design_section8.template.md")).read()
tpl = tpl[tpl.index("-->\n") + 4:]
for k in sorted(rep, key=len, reverse=True):
    tpl = tpl.replace(k, rep[k])
body, row = tpl.split("<!-- ROW -->\n")
d = open(os.path.join(REPO, "DESIGN.md")).read()
a, b = d.index("<!-- SECTION8 BEGIN"), d.index("<!-- SECTION8 END -->")
d = d[:d.index("\n", a) + 1] + body.strip("\n") + "\n" + d[b:]
if "<!-- ROW1 BEGIN -->" in d:
    a, b = d.index("<!-- ROW1 BEGIN -->\n") + len("<!-- ROW1 BEGIN -->\n"), d.index("<!-- ROW1 END -->")
    d = d[:a] + row.strip("\n") + "\n" + d[b:]
open(os.path.join(REPO, "DESIGN.md"), "w").write(d)
print(len(d), "bytes;", {k: rep[k] for k in ("V_EVALS", "V_MS", "V_SWEEP", "V_FRAC", "C2_EVALS", "C3_EVALS", "C5_EVALS", "S_MS")})
