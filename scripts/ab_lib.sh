#!/bin/bash
# Same-box A/B of two builds of the library: runs bench.py alternately with
# face-recognition_amd/frhot/libfrhot_old.so and libfrhot_new.so in place of libfrhot.so
# (rounds = $1, default 2) and prints faces/s and the stage times of every run.
set -o pipefail
cd "$(dirname "$0")/.."
d=face-recognition_amd/frhot
cp $d/libfrhot.so /tmp/libfrhot_keep.so
for r in $(seq 1 ${1:-2}); do
  for v in old new; do
    cp $d/libfrhot_$v.so $d/libfrhot.so
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-scan64 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), {k: round(x, 4) for k, x in d['stages_ms'].items()})" || { cp /tmp/libfrhot_keep.so $d/libfrhot.so; exit 1; }
  done
done
cp /tmp/libfrhot_keep.so $d/libfrhot.so
