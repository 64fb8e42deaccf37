#!/bin/bash
# Longer fuzz campaign of the scan: several seeds, then long descriptors.
set -o pipefail
out=gpurun_out/${1:-fuzz2}
mkdir -p "$out"
for seed in 21 22 23; do
  timeout -k 10 420 python -u scripts/fuzz_search.py 50 $seed 2>&1 | tee "$out/fuzz_search_$seed.log" | grep -v " ok (" || exit 1
done
FUZZ_DIMS=1024,2048,4096 timeout -k 10 420 python -u scripts/fuzz_search.py 30 31 2>&1 | tee "$out/fuzz_search_long_rows.log" | grep -v " ok (" || exit 1
timeout -k 10 300 python -u scripts/fuzz_service.py 2>&1 | tail -3
