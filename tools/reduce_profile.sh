#!/bin/bash
# durations of the reduction kernel by producer, and the gaps around it, for two builds (bench_cache/a.so, b.so)
export TMPDIR=/tmp
cd /tmp
for v in a b; do
  rm -rf /tmp/rp_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$v -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --lib $GRAFT_REPO_ROOT/bench_cache/$v.so --steps 10 --warmup 2 --preroll 30 --no-cpu-baseline > /tmp/rp_$v.log 2>&1
  f=$(find /tmp/rp_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no stats for $v"; tail -5 /tmp/rp_$v.log; continue; }
  echo "== $v"; grep -E "k_reduce_stage|k_apply_list|k_global" $f < /dev/null | cut -d, -f1,2,4,6,7 | sed 's/(GphKargs[^"]*"/"/'
  t=$(find /tmp/rp_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# duration of reduce kernels by the preceding kernel name
d=collections.defaultdict(list)
for i,r in enumerate(rows):
    if r["Kernel_Name"].startswith("k_reduce_stage") and i>0:
        d[rows[i-1]["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]), int(r["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"])))
for k,v in d.items():
    print(k, len(v), "avg dur us %.1f" % (sum(x[0] for x in v)/len(v)/1e3), "avg gap before us %.1f" % (sum(x[1] for x in v)/len(v)/1e3))
# gaps after reduce
g=[]
for i,r in enumerate(rows[:-1]):
    if r["Kernel_Name"].startswith("k_reduce_stage"):
        g.append(int(rows[i+1]["Start_Timestamp"])-int(r["End_Timestamp"]))
print("avg gap after reduce us %.1f" % (sum(g)/len(g)/1e3))
PY
done
