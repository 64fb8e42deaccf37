#!/bin/bash
# Same-box A/B of the second half's start (staggered behind the first half's conv1_1 or not) and of
# conv1_1's block size.  scripts/gpu_stagger.sh <tag>
set -o pipefail
out=gpurun_out/${1:-stagger}
mkdir -p "$out"
run() {
  env "$@" timeout -k 10 200 python -u bench.py --steps 40 --warmup 5 --no-scan64 --no-host-leg --no-cpu-baseline --no-detector 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stages_ms'].items()})" | tee -a "$out/ab.txt"
}
for rep in 1 2; do
  run FRHOT_NO_STAGGER=1 FRHOT_CONV11_WAVES=4 || exit 1
  run FRHOT_NO_STAGGER=0 FRHOT_CONV11_WAVES=4 || exit 1
  run FRHOT_NO_STAGGER=0 FRHOT_CONV11_WAVES=2 || exit 1
  run FRHOT_NO_STAGGER=1 FRHOT_CONV11_WAVES=2 || exit 1
done
