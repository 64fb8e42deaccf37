#!/bin/bash
# Kernel trace of the headline bench (no detector leg): per-kernel statistics into gpurun_out/<tag>/.
out=gpurun_out/${1:-trace}
mkdir -p "$out"
# optional second argument: another build of the library to trace instead (copied over the in-tree one on the box)
[ -n "$2" ] && cp "$2" face-recognition_amd/frhot/libfrhot.so
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-detector --steps 10 --no-cpu-baseline --no-host-leg --no-scan4096 --no-scan-extra > "$out/bench.json" 2> "$out/trace.err" || { tail -20 "$out/trace.err"; exit 1; }
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
t=$(find "$out/trace" -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && [ -n "$t" ] || { find "$out/trace" | head; exit 1; }
cp "$f" "$out/kernel_stats.csv"
python - "$t" "$out/timeline.txt" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last embed step: find the last resize kernel pair and print the kernels from there to the next scan
idx=[i for i,r in enumerate(rows) if 'resize_u8' in r['Kernel_Name']]
i0=idx[-2] if len(idx)>=2 else 0
t0=int(rows[i0]['Start_Timestamp'])
with open(sys.argv[2],'w') as f:
    for r in rows[i0:i0+70]:
        n=r['Kernel_Name']
        n=n[:70]
        f.write('%9.1f %9.1f q%-3s %s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),n))
P
rm -rf "$out/trace"
head -25 "$out/kernel_stats.csv" | cut -c1-150
