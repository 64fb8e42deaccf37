#!/bin/bash
# GPU pass: parity tests, kernel micro-bench, SQ counters over the micro-bench, host-path rate.
set -o pipefail
tag=${1:-r1d}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
timeout -k 10 120 ./scripts/kbench all > "$out/kbench.log" 2>&1 || { tail -30 "$out/kbench.log"; exit 1; }
cat "$out/kbench.log"
timeout -k 10 200 python -u scripts/host_path_rate.py > "$out/host_path.log" 2>&1 || { tail -30 "$out/host_path.log"; exit 1; }
tail -1 "$out/host_path.log"
bash scripts/pmc_kbench.sh conv $tag/sqconv > "$out/sq_conv.txt" 2>&1 || { tail -30 "$out/sq_conv.txt"; exit 1; }
bash scripts/pmc_kbench.sh scan $tag/sqscan > "$out/sq_scan.txt" 2>&1 || { tail -30 "$out/sq_scan.txt"; exit 1; }
cat "$out/sq_conv.txt" "$out/sq_scan.txt"
