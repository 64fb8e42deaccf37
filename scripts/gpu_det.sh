#!/bin/bash
# Detector: GPU tests, throughput at three batch sizes, per-kernel trace.  scripts/gpu_det.sh <tag>
set -o pipefail
out=gpurun_out/${1:-det}
mkdir -p "$out"
timeout -k 10 400 python -u -m pytest tests/test_gpu_detector.py -x -q --timeout 180 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
for b in 1 8 32; do
  timeout -k 10 120 python -u scripts/det_bench.py $b 20 2>/dev/null | tee -a "$out/det_bench.txt" || exit 1
done
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace" -- python3 "$root/scripts/det_bench.py" 32 10 > "$root/$out/trace.log" 2>&1 || { tail -20 "$root/$out/trace.log"; exit 1; }
cd "$root"
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
head -12 "$f" | cut -c1-200
