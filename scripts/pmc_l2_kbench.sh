#!/bin/bash
# L2 hit rate per kernel over kbench: $1 = kbench args, $2 = tag
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/$2
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/l2 -- ./scripts/kbench $1 > $out/l2.log 2>&1 || { tail -20 $out/l2.log; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/l2/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    if "fill" in k or "rocclr" in k: continue
    row={n: sum(v)/len(v) for n,v in c.items()}
    h,m=row.get("TCC_HIT_sum",0),row.get("TCC_MISS_sum",0)
    print("%-72s L2 req %10.0f k  hit %.3f  EA rdreq %10.0f k" % (k[:72], (h+m)/1e3, h/max(1,h+m), row.get("TCC_EA0_RDREQ_sum",0)/1e3))
PY
