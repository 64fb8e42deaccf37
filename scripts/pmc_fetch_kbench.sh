#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel over kbench: $1 = kbench args, $2 = tag
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/$2
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- ./scripts/kbench $1 > $out/f.log 2>&1 || { tail -20 $out/f.log; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/f/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    if "fill" in k or "rocclr" in k: continue
    row={n: sum(v)/len(v) for n,v in c.items()}
    print("%-75s fetch x2 = %8.1f MB   L2 hit %.3f" % (k[:75], row.get("FETCH_SIZE",0)*2/1024, row.get("TCC_HIT_sum",0)/max(1,row.get("TCC_HIT_sum",0)+row.get("TCC_MISS_sum",0))))
PY
