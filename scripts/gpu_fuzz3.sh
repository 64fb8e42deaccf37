#!/bin/bash
# Fuzzers of the embed side and the service on one box: scripts/gpu_fuzz3.sh <tag>
set -o pipefail
out=gpurun_out/${1:-fuzz3}
mkdir -p "$out"
for f in fuzz_batches fuzz_preprocess fuzz_sharpness fuzz_recognise_keys fuzz_mutation; do
  timeout -k 10 400 python -u scripts/$f.py > "$out/$f.log" 2>&1 || { tail -20 "$out/$f.log"; exit 1; }
  tail -1 "$out/$f.log"
done
timeout -k 10 300 python -u scripts/fuzz_detector.py 30 2 > "$out/fuzz_detector.log" 2>&1 || { tail -20 "$out/fuzz_detector.log"; exit 1; }
tail -1 "$out/fuzz_detector.log"
