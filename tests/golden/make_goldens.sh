#!/bin/bash
# Regenerates every golden vector in this directory from the REAL reference
# (oracle/_ref/gphocs_ref = /root/reference/src compiled by oracle/Makefile).
# Only runs where /root/reference exists (the build container).  The fixtures
# are data: inputs (ctl/seq), the processed-locus pack, per-proposal records
# (accept counts + accumulators as hex floats) and per-locus state dumps.
set -euo pipefail
cd "$(dirname "$0")"
REPO=$(cd ../.. && pwd)
make -C "$REPO/oracle" ref >/dev/null
REF="$REPO/oracle/_ref/gphocs_ref"
GEN="python3 $REPO/tools/gen_synth.py"

timeout 60 $REF rng 12345 300 > rng_12345.txt
timeout 60 $REF rng 777 300 > rng_777.txt
timeout 60 $REF reflect > reflect.txt

gen() { # name config loci seqlen iters perlog extra...
  local name=$1 cfg=$2 loci=$3 seqlen=$4 iters=$5 perlog=$6; shift 6
  $GEN --config $cfg --loci $loci --seqlen $seqlen --iters $iters --per-log $perlog --out $name "$@" 2>/dev/null
  timeout 600 $REF pack $name.ctl $name.gpk >/dev/null
  timeout 600 $REF run $name.ctl 0 $name.init.rtrace $name.init.state -1 1 >/dev/null
  timeout 1800 $REF run $name.ctl $iters $name.rtrace $name.state $((iters-1)) 1 >/dev/null
  rm -f $name.init.rtrace
}
gen g1 1 24 400 30 10
gen g2 2 20 400 30 15
gen m3 3 16 300 120 40 --mig-beta 0.00000004
gen m4 4 12 300 60 20 --mig-beta 0.00000004
gen c5 5 10 300 30 10
gen s3 3 12 300 40 20 --start-mig 10 --mig-beta 0.0000001 --no-mixing

# estimated sample ages ("age x e"): UpdateSampleAge (GPhoCS.c:4006) is live
gen a6 6 12 300 80 20 --mig-beta 0.00000004
gen a7 7 12 300 100 25 --mig-beta 0.00000004
# variable locus rates ("locus-mut-rate VAR a"): UpdateLocusRate (GPhoCS.c:4598) is live; v9 has a large step
# (reflections at both ends of (0, rold + rref)) and alpha != 1 (the Dirichlet prior term)
gen v8 3 16 300 60 20 --mig-beta 0.00000004 --var-rates 1.0 0.3
gen v9 2 14 300 60 20 --var-rates 1.7 1.1
# the reference's own trace files (its main(), unmodified): what G-PhoCS-hip must reproduce
for name in g1 m3 a7 v8; do timeout 900 $REF main -n 1 $name.ctl >/dev/null 2>&1; done
# front-end stress input (IUPAC codes, haploids, missing/unknown samples, all-N columns)
python3 make_stress.py
timeout 600 $REF pack stress.ctl stress.gpk >/dev/null
ls -la
# z0: g1 with locus 3 all-N (a locus with zero informative columns is legal upstream: P = 0)
# (z0.ctl / z0.seq are derived from g1 by hand: sequences of locus3 replaced by N, file names changed)
timeout 600 $REF pack z0.ctl z0.gpk >/dev/null
timeout 600 $REF run z0.ctl 0 z0.init.rtrace z0.init.state -1 1 >/dev/null; rm -f z0.init.rtrace
timeout 600 $REF run z0.ctl 12 z0.rtrace z0.state 11 1 >/dev/null
# f3: the find-finetunes search (GPhoCS.c:1896-2180) -- config 3 with find-finetunes TRUE, 6 steps of 10 samples;
# only the reference's own trace file is kept (f3.ctl is gen_synth output with the three find-finetunes lines edited in)
timeout 900 $REF main -n 1 f3.ctl >/dev/null 2>&1

# the reference's readTrace tool on its own trace files (oracle/_ref/readTrace_ref = src/readTrace.c compiled as is);
# only block layouts without a trailing partial block are kept, and `-d` with the default block: a partial tail
# after complete blocks prints uninitialised storage upstream (readTrace.c:253-264)
RT=../../oracle/_ref/readTrace_ref
$RT g1.trace        > g1.readtrace_all.txt
$RT g1.trace -b 3   > g1.readtrace_b3.txt
$RT m3.trace -b 40  > m3.readtrace_b40.txt
$RT a7.trace -d 20  > a7.readtrace_d20.txt
$RT f3.trace -b 10 -d 20 > f3.readtrace_b10_d20.txt

# the reference binary's stdout log for the same runs (banner, title, one line per log period, finetune search):
# G-PhoCS-hip prints the same text from "Starting MCMC" on (the elapsed-time column aside)
for name in g1 f3 v8 a7; do timeout 900 $REF main -n 1 $name.ctl > $name.stdout 2>/dev/null; done

# w2: primary + SECONDARY control file (GPhoCS.c:35-43, 154-164; readSecondaryControlFile, MCMCcontrol.c:178-210):
# w2b.ctl (hand-written) overrides GENERAL-INFO keys (seed, iterations, log period, two finetunes, band print factor
# and prior) and replaces the two bands of w2.ctl by three others.  pack / run take it through GPH_REF_CTL2.
$GEN --config 3 --loci 14 --seqlen 300 --iters 60 --per-log 20 --mig-beta 0.00000004 --out w2 2>/dev/null
GPH_REF_CTL2=w2b.ctl timeout 600 $REF pack w2.ctl w2.gpk >/dev/null
GPH_REF_CTL2=w2b.ctl timeout 600 $REF run w2.ctl 0 w2.init.rtrace w2.init.state -1 1 >/dev/null; rm -f w2.init.rtrace
GPH_REF_CTL2=w2b.ctl timeout 600 $REF run w2.ctl 50 w2.rtrace w2.state 49 1 >/dev/null
timeout 900 $REF main -n 1 w2.ctl w2b.ctl > w2.stdout 2>/dev/null

# x8: the engine's hard caps -- 32 leaves, 31 populations (16 current), 16 migration bands (library variant `x`)
gen x8 8 10 300 24 8 --mig-beta 0.00000004
timeout 900 $REF main -n 1 x8.ctl >/dev/null 2>&1     # x8.trace: the reference's own trace file at the caps

# j1-j3 (round 6): population trees that are NOT caterpillars, migration bands with ANCESTRAL endpoints (tools/gen_synth.py configs
# 20-22): j1 = (((A,B),(C,D)),E) with AB->CD, CD->AB, C->AB, E->ABCD; j2 = ((A,(B,C)),((D,E),F)) with 8 mixed bands; j3 =
# ((A,B),(C,D)) with an ESTIMATED ancient sample in C below the band target CD.  UpdateTau's band-start branches (GPhoCS.c:3353-3431)
gen j1 20 12 300 150 50 --mig-beta 0.00000004
gen j2 21 10 300 100 25 --mig-beta 0.00000004
gen j3 22 12 300 120 40 --mig-beta 0.00000004
# randomised model shapes (tools/random_models.py: random binary trees, random legal bands incl. ancestral ends, optional ancient
# sample): the reference's pack and records of 12 models that run through and 3 on which the reference itself aborts
python3 $REPO/tools/random_models.py fixtures rnd 2 3 6 17 19 23 24 42 43 55 59 72 34 46 35

# kernel-level fixtures: single calls of the reference's per-locus functions after N iterations
for c in "m4 60" "a7 40" "g2 20" "j1 70" "j2 50"; do set -- $c; timeout 300 $REF unit $1.ctl $2 $1.unit >/dev/null 2>&1; done
for c in "m4 60" "a7 40" "g2 20" "j1 70" "j2 50"; do set -- $c; timeout 600 $REF unit2 $1.ctl $2 $1.unit2 >/dev/null 2>&1; done   # executeGenSPR, scaleAllNodeAges, rubberBandRipple (migration events; band START / END events), traceLineage

# y9: beyond 32 leaves / 32 populations -- 40 leaves, 20 current populations (the reference's NSPECIES cap: 39 populations),
# 16 migration bands (library variant `h`: two genealogy nodes per lane would not do, the node sets are 128 bits wide)
gen y9 9 8 300 16 8 --mig-beta 0.00000004
timeout 900 $REF main -n 1 y9.ctl >/dev/null 2>&1     # y9.trace: the reference's own trace file

# r5: locus-mut-rate FIXED <rate file> (readRateFile, GPhoCS.c:491-579; MCMCcontrol.c:700-712): 16 rates spread over 0.2 .. 5
# in a mixed layout, normalised to mean 1 by the reference; pack (carries the normalised rates), records, state, the real
# binary's trace file and stdout.  r5_{few,many,neg,missing}: the same control file with a rate file of 15 / 17 entries, a
# negative entry, no file -- the real binary's stderr (r5_*.ctl / r5_*.rates are derived from r5 by the lines below)
gen r5 3 16 300 60 20 --mig-beta 0.00000004 --fixed-rates
timeout 900 $REF main -n 1 r5.ctl > r5.stdout 2>/dev/null
python3 - <<'PY'
r = open('r5.rates').read().split()
open('r5_few.rates', 'w').write(' '.join(r[:15]) + '\n')
open('r5_many.rates', 'w').write(' '.join(r + ['1.5']) + '\n')
rr = list(r); rr[6] = '-0.25'
open('r5_neg.rates', 'w').write(' '.join(rr) + '\n')
c = open('r5.ctl').read()
for k in ('few', 'many', 'neg', 'missing'):
    open(f'r5_{k}.ctl', 'w').write(c.replace('r5.rates', f'r5_{k}.rates').replace('r5.trace', f'r5_{k}.trace'))
PY
for k in few many neg missing; do timeout 60 $REF main -n 1 r5_$k.ctl > /dev/null 2> r5_$k.stderr || true; rm -f r5_$k.trace; done

# b2: more than 16 migration bands (20; the reference allows MAX_MIG_BANDS 100, patch.h:17): library variant `b`
gen b2 12 10 300 24 8 --mig-beta 0.00000004
timeout 900 $REF main -n 1 b2.ctl >/dev/null 2>&1     # b2.trace: the reference's own trace file

# n7: more than 64 leaves (36 diploids over 6 populations = 72; the reference allows NS 200, patch.h:22): library variant `n`
gen n7 13 6 200 12 6 --mig-beta 0.0000001
timeout 900 $REF main -n 1 n7.ctl >/dev/null 2>&1     # n7.trace: the reference's own trace file
# q6: a locus whose sequence block outgrows the LDS budget -- 72 leaves, two 20-kb loci: 145 and 698 phased patterns, up to 512
# phases per pattern.  State dumps WITHOUT the conditional arrays (4 MB of hex floats); the 1.4-MB sequence file is regenerated
# here (deterministic generator) and not kept: the tests read the pack
$GEN --config 13 --loci 2 --seqlen 20000 --iters 8 --per-log 4 --mut-scale 1 --mig-beta 0.0000001 --out q6 2>/dev/null
timeout 600 $REF pack q6.ctl q6.gpk >/dev/null
timeout 600 $REF run q6.ctl 0 q6.init.rtrace q6.init.state -1 0 >/dev/null; rm -f q6.init.rtrace
timeout 1800 $REF run q6.ctl 8 q6.rtrace q6.state 7 0 >/dev/null
rm -f q6.seq

# p6: a repeated column of 16 heterozygotes among 18 diploids = 2^16 phases of one pattern, 65 554 phased patterns in one locus
# (make_p6.py writes the inputs): only the real binary's trace file is kept (the pack would be 2.7 MB)
python3 make_p6.py
timeout 900 $REF main -n 1 p6.ctl >/dev/null 2>&1

# decision-level fixtures: the reference compiled with -DLOG_STEPS (oracle/_ref/gphocs_ref_log), two loci each of m3 and a7
python3 make_logsteps.py
