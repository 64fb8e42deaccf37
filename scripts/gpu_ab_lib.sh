#!/bin/bash
# Same-box comparison of builds of the library under the headline bench:
#   scripts/gpu_ab_lib.sh <tag> <rounds> <other.so>...     (the first variant is frhot/libfrhot.so as built)
out=gpurun_out/${1:-ablib}
rounds=${2:-2}
shift 2
mkdir -p "$out"
lib=face-recognition_amd/frhot/libfrhot.so
cp "$lib" "$out/v0.so" || exit 1
n=0
for o in "$@"; do n=$((n+1)); cp "$o" "$out/v$n.so" || exit 1; done
for r in $(seq 1 "$rounds"); do
  for v in $(seq 0 $n); do
    cp "$out/v$v.so" "$lib"
    timeout -k 10 200 python bench.py --no-detector --steps 40 --no-cpu-baseline --no-scan4096 --no-scan-extra > "$out/bench_${v}_$r.json" 2> "$out/bench_${v}_$r.err" || { tail -20 "$out/bench_${v}_$r.err"; cp "$out/v0.so" "$lib"; rm -f "$out"/v?.so; exit 1; }
    python - "$out/bench_${v}_$r.json" "$v" "$r" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('variant', sys.argv[2], 'round', sys.argv[3], round(d['value']), round(d['ms_per_step'],3), {k: round(x,3) for k,x in d['stages_ms'].items()}, 'one caller', round(d['pcie_inclusive']['ms_per_batch_one_caller'],2), 'two callers', round(d['pcie_inclusive']['ms_per_batch'],2), flush=True)
P
  done
done
cp "$out/v0.so" "$lib"
rm -f "$out"/v?.so
