#!/bin/bash
# bench.py alone (no CPU baseline, no host leg unless asked): scripts/gpu_bench_only.sh <tag> [bench args...]
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p "$out"
timeout -k 10 400 python -u bench.py "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
python3 - "$out/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), {k: round(v,4) for k,v in d["stages_ms"].items()})
for k in d:
    if k.startswith("roofline"):
        r=d[k]; print(k, "frac", round(r["frac"],4), "ms", round(r.get("ms_per_launch", r.get("ms_per_batch", 0)),4), r.get("top1_is_the_asked_row"))
print("gates", d["gates"])
PY
