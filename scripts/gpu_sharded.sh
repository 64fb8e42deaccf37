#!/bin/bash
# Sharded serving on one box: the one-rank GPU test, then 2- and 3-rank rehearsals sharing the GPU (gloo).
set -o pipefail
out=gpurun_out/${1:-sharded}
mkdir -p "$out"
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "sharded_service" > "$out/test.log" 2>&1 || { tail -40 "$out/test.log"; exit 1; }
tail -2 "$out/test.log"
for w in 2 3; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29530 + w)) scripts/sharded_service_rehearsal.py 40 > "$out/rehearsal_$w.json" 2> >(tee "$out/rehearsal_$w.err" >&2) || { tail -30 "$out/rehearsal_$w.err"; exit 1; }
  cat "$out/rehearsal_$w.json"
done
