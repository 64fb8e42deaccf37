#!/bin/bash
# Fuzz campaigns of the search paths on one box: scripts/gpu_fuzz.sh <tag> [search cases] [seed]
set -o pipefail
out=gpurun_out/${1:-fuzz}
mkdir -p "$out"
timeout -k 10 500 python -u scripts/fuzz_search.py ${2:-60} ${3:-11} > "$out/fuzz_search.log" 2>&1 || { tail -20 "$out/fuzz_search.log"; exit 1; }
tail -2 "$out/fuzz_search.log"
timeout -k 10 300 python -u scripts/fuzz_recognise_keys.py > "$out/fuzz_recognise_keys.log" 2>&1 || { tail -20 "$out/fuzz_recognise_keys.log"; exit 1; }
tail -2 "$out/fuzz_recognise_keys.log"
timeout -k 10 300 python -u scripts/fuzz_mutation.py > "$out/fuzz_mutation.log" 2>&1 || { tail -20 "$out/fuzz_mutation.log"; exit 1; }
tail -2 "$out/fuzz_mutation.log"
