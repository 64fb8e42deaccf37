#!/bin/bash
# Selector-wave scan: micro-benchmark A/B of the 64-query tile (and in-kernel stamps), then the scan parity tests.
set -o pipefail
out=gpurun_out/${1:-sel}
mkdir -p "$out"
KBENCH_SEL=${2:-2} timeout -k 10 240 ./scripts/kbench scan > "$out/kbench_sel.txt" 2>&1 || { tail -20 "$out/kbench_sel.txt"; exit 1; }
cat "$out/kbench_sel.txt"
if [ -x scripts/kbench_dbg ]; then
  KBENCH_SEL=1 timeout -k 10 240 ./scripts/kbench_dbg scan > "$out/kbench_dbg.txt" 2>&1 || { tail -20 "$out/kbench_dbg.txt"; exit 1; }
  cat "$out/kbench_dbg.txt"
fi
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "search or scan or config or recognise or no_alloc or shard" > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
fi
