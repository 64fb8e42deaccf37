#!/bin/bash
# Timing of ablated builds of the Winograd kernel (results are wrong by construction).
set -o pipefail
out=gpurun_out/${1:-wino_abl}
mkdir -p "$out"
for v in 1 2 3 4 8 12 16 32; do
  echo "== ablation $v (1 no transform math, 2 no V writes, 4 no U loads, 8 no patch DMA, 16 no MFMA, 32 no patch reads)"
  timeout -k 10 60 scripts/kbench_abl$v wino 256 2>&1 | grep "hw  28 cin 512 cout 512 pool 0" | tee -a "$out/abl.txt" || exit 1
done
