#!/bin/bash
# Winograd form of the conv layers against the direct kernels (kbench wino [B]).
set -o pipefail
out=gpurun_out/${1:-wino}
mkdir -p "$out"
timeout -k 10 120 scripts/kbench wino ${2:-256} 2>&1 | tee "$out/kbench_wino.txt" || exit 1
if [ -x scripts/kbench_dbg ]; then
  timeout -k 10 120 scripts/kbench_dbg wino ${2:-256} 2>&1 | tee "$out/kbench_wino_stamps.txt"
fi
