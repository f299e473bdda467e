#!/usr/bin/env python3
"""Randomised differential test of the model shapes: random binary population trees, 0-8 random LEGAL
migration bands incl. ancestral endpoints, random sample counts, an optional ancient sample (fixed or estimated).

    random_models.py diff  N OUTDIR [--summary FILE]   N models: real reference (oracle/_ref/gphocs_ref) vs oracle/gphocs_oracle,
                                                       records and final per-locus state byte for byte (build container only)
    random_models.py gpu   N OUTDIR [--summary FILE]   N models on the MI355X: control + sequence file through the library's own front
                                                       end, the HIP engine against the ORACLE run live (the GPU box has no reference;
                                                       the oracle equals it on these very models: `diff`) -- records (accept counters
                                                       exact, sums over loci <= 1e-10) and final per-locus state byte for byte; a model
                                                       the oracle aborts on must fail in the same iteration
    random_models.py fixtures DIR K...                 models K... as committed fixtures: <DIR>/rKK.gpk (the reference's pack) +
                                                       rKK.rtrace (the reference's records) + rKK.json (the model)

A band is legal when neither end is an ancestor of the other (MCMCcontrol.c:1229-1236) and the two populations co-exist at the
initial split times; UpdateTau keeps a living band alive (GPhoCS.c:3276-3292).  Model k is a pure function of k (seeded)."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.path.join(REPO, "oracle", "_ref", "gphocs_ref")
ORA = os.path.join(REPO, "oracle", "gphocs_oracle")


def random_model(k):
    rng = np.random.default_rng(9000 + k)
    kc = int(rng.integers(2, 8))
    cur = [chr(65 + i) for i in range(kc)]
    # random binary tree: join two random roots until one is left
    roots = [(c, [c]) for c in cur]
    parent, kids, taus, order = {}, {}, {}, []
    start = {c: 0.0 for c in cur}
    while len(roots) > 1:
        i, j = sorted(rng.choice(len(roots), 2, replace=False))
        (a, la), (b, lb) = roots[i], roots[j]
        if rng.random() < 0.5:
            a, la, b, lb = b, lb, a, la
        leaves = sorted(la + lb)
        nm = "".join(leaves)
        lo = max(start[a], start[b])
        taus[nm] = float(rng.uniform(3e-6, 8e-6)) if lo == 0.0 else lo * float(rng.uniform(1.25, 2.4))
        start[nm] = taus[nm]
        parent[a] = parent[b] = nm
        kids[nm] = (a, b)
        order.append(nm)
        roots = [r for q, r in enumerate(roots) if q not in (i, j)] + [((a, b), leaves)]
        roots[-1] = (nm, leaves)
    rootname = order[-1]

    def tree(nm):
        return nm if nm in cur else [tree(kids[nm][0]), tree(kids[nm][1])]

    def is_anc(a, b):      # a is an ancestor of b (or equal)
        while b is not None:
            if a == b:
                return True
            b = parent.get(b)
        return False
    end = {p: taus[parent[p]] for p in parent}
    end[rootname] = float("inf")
    allp = cur + order
    cand = [(s, t) for s in allp for t in allp if s != t and not is_anc(s, t) and not is_anc(t, s)
            and max(start[s], start[t]) < min(end[s], end[t])]
    nb = int(min(len(cand), rng.integers(0, 9)))
    bands = [cand[q] for q in sorted(rng.choice(len(cand), nb, replace=False))] if nb else []
    pops = [int(rng.integers(1, 3)) for _ in cur]
    while sum(pops) > 10:
        pops[int(np.argmax(pops))] -= 1
    cfg = dict(pops=pops, tree=tree(rootname), bands=[list(b) for b in bands], loci=6, data_seed=1000 + k,
               taus={("root" if nm == rootname else nm): taus[nm] for nm in order})
    if rng.random() < 0.4:
        cfg["ancient"] = int(rng.integers(0, kc))
        cfg["ancient_est"] = bool(rng.random() < 0.6)
    run = dict(loci=int(rng.integers(4, 9)), seqlen=200, iters=60, per_log=int(rng.choice([15, 20, 30])),
               mcmc_seed=int(rng.integers(1, 2 ** 31 - 1)),
               mig_beta=float(rng.choice([4e-8, 4e-8, 1e-7, 1e-5])), start_mig=int(rng.choice([0, 0, 0, 7])))
    return cfg, run


def generate(k, outdir):
    cfg, run = random_model(k)
    name = f"r{k:02d}"
    mj = os.path.join(outdir, name + ".json")
    with open(mj, "w") as f:
        json.dump(dict(model=cfg, run=run), f, indent=1, sort_keys=True)
        f.write("\n")
    tmpj = os.path.join(outdir, name + ".model.json")
    with open(tmpj, "w") as f:
        json.dump(cfg, f)
    cmd = [sys.executable, os.path.join(HERE, "gen_synth.py"), "--model-json", tmpj, "--loci", str(run["loci"]), "--seqlen",
           str(run["seqlen"]), "--iters", str(run["iters"]), "--per-log", str(run["per_log"]), "--mcmc-seed", str(run["mcmc_seed"]),
           "--mig-beta", f"{run['mig_beta']:.10f}", "--out", os.path.join(outdir, name)]
    if run["start_mig"]:
        cmd += ["--start-mig", str(run["start_mig"])]
    subprocess.run(cmd, check=True, capture_output=True)
    os.unlink(tmpj)
    return name, cfg, run


def reference_run(name, outdir, iters):
    r1 = subprocess.run([REF, "pack", name + ".ctl", name + ".gpk"], cwd=outdir, capture_output=True, timeout=600)
    r2 = subprocess.run([REF, "run", name + ".ctl", str(iters), name + ".rtrace", name + ".state", str(iters - 1), "1"], cwd=outdir,
                        capture_output=True, timeout=1800)
    return r1.returncode, r2.returncode


def diff(n, outdir, summary):
    os.makedirs(outdir, exist_ok=True)
    rows, bad = [], 0
    for k in range(n):
        name, cfg, run = generate(k, outdir)
        rc = reference_run(name, outdir, run["iters"])
        o = subprocess.run([ORA, "run", name + ".gpk", str(run["iters"]), name + ".o.rtrace", name + ".o.state",
                            str(run["iters"] - 1), "1"], cwd=outdir, capture_output=True, timeout=1800)
        def rd(ext):
            pth = os.path.join(outdir, name + ext)
            return open(pth, "rb").read() if os.path.exists(pth) else b""
        # the reference ABORTS on some legal models (a band a few 1e-9 long: "Fatal Error 0025", a failed checkAll): the oracle
        # must then abort too, after the same records
        aborted = rc[1] != 0
        same_t = rc[0] == 0 and (o.returncode != 0) == aborted and rd(".rtrace") == rd(".o.rtrace")
        same_s = same_t and (aborted or rd(".state") == rd(".o.state"))
        tr = rd(".rtrace").decode().splitlines()
        conflicts = [int(l.split()[1]) for l in tr if l.startswith("CONFLICTS")]
        migs = sum(int(l.split()[3]) for l in tr if " MIGN " in l)
        anc_bands = sum(1 for s, t in cfg["bands"] if len(s) > 1 or len(t) > 1)
        rows.append(dict(model=name, pops=len(cfg["pops"]), leaves=2 * sum(cfg["pops"]), bands=len(cfg["bands"]),
                         bands_with_ancestral_end=anc_bands, ancient=("e" if cfg.get("ancient_est") else "f") if "ancient" in cfg else "-",
                         tree=json.dumps(cfg["tree"]).replace('"', "").replace(" ", ""), loci=run["loci"], iters=run["iters"],
                         reference_aborts=aborted, records=len(tr), conflicts=conflicts[-1] if conflicts else None,
                         accepted_mig_node_moves=migs, records_equal=bool(same_t), state_equal=bool(same_s)))
        bad += not (same_t and same_s)
        print(rows[-1], flush=True)
    out = dict(what="random model shapes: real reference (oracle/_ref/gphocs_ref) vs oracle/gphocs_oracle, records and final per-locus "
                    "state byte for byte", models=n, failures=bad,
               with_ancestral_band_ends=sum(1 for r in rows if r["bands_with_ancestral_end"]),
               non_caterpillar=sum(1 for r in rows if r["tree"].count("],[") or r["tree"].count("],") and r["tree"].count(",[")),
               with_conflicts=sum(1 for r in rows if r["conflicts"]), reference_aborts=sum(1 for r in rows if r["reference_aborts"]),
               rows=rows)
    if summary:
        with open(summary, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    print(f"{n} models, {bad} failures")
    return bad


def gpu(n, outdir, summary):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import gphocs_amd as G
    from gphocs_amd_pkg import synth
    from parity_util import compare_records, compare_states
    os.makedirs(outdir, exist_ok=True)
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True, capture_output=True)
    rows, bad = [], 0
    for k in range(n):
        name, cfg, run = generate(k, outdir)
        it = run["iters"]
        pk = G.Pack.from_control(os.path.join(outdir, name + ".ctl"), seq_path=os.path.join(outdir, name + ".seq"))
        pth = os.path.join(outdir, name + ".gpk")
        synth.write_pack(pk, pth)
        o = subprocess.run([ORA, "run", pth, str(it), name + ".o.rtrace", name + ".o.state", str(it - 1), "1"], cwd=outdir, capture_output=True, timeout=1800)
        s = G.Sampler(pk, lib=G.load_library(dims=(pk.n, pk.K, pk.B)))
        tr, st = os.path.join(outdir, name + ".h.rtrace"), os.path.join(outdir, name + ".h.state")
        s.set_record_file(tr)
        failed = None
        try:
            try:
                s.initialize()
            except RuntimeError:
                failed = -1
            for q in range(it if failed is None else 0):
                try:
                    s.iteration(q)
                except RuntimeError:
                    failed = q
                    break
            if failed is None:
                s.dump_state(st, True)
        finally:
            s.set_record_file(None)
            s.close()
        ok, why = True, ""
        try:
            if o.returncode != 0:
                its = [int(l.split()[1]) for l in open(os.path.join(outdir, name + ".o.rtrace")) if l.startswith("IT ")]
                last = max(its) if its else -1
                assert failed is not None and failed in (last, last + 1), f"oracle aborts in {last}, the engine {failed}"
            else:
                assert failed is None, f"the engine failed in iteration {failed}"
                compare_records(tr, os.path.join(outdir, name + ".o.rtrace"))
                compare_states(st, os.path.join(outdir, name + ".o.state"))
        except AssertionError as ex:
            ok, why = False, str(ex)[:300]
        bad += not ok
        rows.append(dict(model=name, pops=len(cfg["pops"]), leaves=int(pk.n), bands=len(cfg["bands"]),
                         bands_with_ancestral_end=sum(1 for a, b in cfg["bands"] if len(a) > 1 or len(b) > 1),
                         tree=json.dumps(cfg["tree"]).replace('"', "").replace(" ", ""), oracle_aborts=o.returncode != 0, engine_failed_in=failed,
                         equal=ok, why=why))
        print(rows[-1], flush=True)
    out = dict(what="random model shapes on the MI355X: HIP engine (C ABI) against the oracle run live on the same pack: accept counters exact, "
                    "sums over loci <= 1e-10, final per-locus state byte for byte; oracle aborts must be matched in the same iteration",
               models=n, failures=bad, with_ancestral_band_ends=sum(1 for r in rows if r["bands_with_ancestral_end"]),
               oracle_aborts=sum(1 for r in rows if r["oracle_aborts"]), rows=rows)
    if summary:
        with open(summary, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    print(f"{n} models, {bad} failures")
    return bad


def fixtures(ids, d):
    """models `ids` as committed fixtures: the reference's pack and records (partial, when the reference aborts), the model"""
    os.makedirs(d, exist_ok=True)
    import tempfile
    for k in ids:
        with tempfile.TemporaryDirectory() as td:
            name, cfg, run = generate(k, td)
            rc = reference_run(name, td, run["iters"])
            assert rc[0] == 0, (name, rc)
            meta = json.load(open(os.path.join(td, name + ".json")))
            meta["reference_aborts"] = rc[1] != 0
            if rc[1] != 0:      # the last iteration the reference began
                its = [int(l.split()[1]) for l in open(os.path.join(td, name + ".rtrace")) if l.startswith("IT ")]
                meta["reference_last_iteration"] = max(its) if its else -1
            with open(os.path.join(td, name + ".json"), "w") as f:
                json.dump(meta, f, indent=1, sort_keys=True)
                f.write("\n")
            for ext in (".gpk", ".rtrace", ".json"):
                os.replace(os.path.join(td, name + ext), os.path.join(d, name + ext))
    print(f"wrote {len(ids)} fixtures into {d}")


if __name__ == "__main__":
    a = sys.argv
    if len(a) >= 4 and a[1] == "diff":
        sys.exit(1 if diff(int(a[2]), a[3], a[a.index("--summary") + 1] if "--summary" in a else None) else 0)
    if len(a) >= 4 and a[1] == "gpu":
        sys.exit(1 if gpu(int(a[2]), a[3], a[a.index("--summary") + 1] if "--summary" in a else None) else 0)
    if len(a) >= 4 and a[1] == "fixtures":
        fixtures([int(x) for x in a[3:]], a[2])
        sys.exit(0)
    sys.exit(__doc__)
