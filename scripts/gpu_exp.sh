#!/bin/bash
# A/B of kbench builds in one box: scripts/gpu_exp.sh <tag> <rounds> <binary>...
set -o pipefail
out=gpurun_out/$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq $rounds); do
  for b in "$@"; do
    echo "== $b (round $r)" | tee -a "$out/exp.txt"
    KBENCH_SEL=1 KBENCH_SELV=${SELV:-12} timeout -k 10 120 ./scripts/$b scan 2>&1 | grep -v "selector kernel: loop end" | tee -a "$out/exp.txt" || exit 1
  done
done
