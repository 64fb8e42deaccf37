#!/bin/bash
# Two and four ranks of bench.py sharing the box's one GPU over gloo (RCCL refuses two ranks on
# one device): the N > 1 code path of the step on real kernels.  scripts/gpu_two_ranks.sh <tag>
set -o pipefail
out=gpurun_out/${1:-two_ranks}
mkdir -p "$out"
export FRHOT_BENCH_SHARE_GPU=1 FRHOT_BENCH_BACKEND=gloo FRHOT_BENCH_WATCHDOG_S=150
for n in 2 4; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n \
    bench.py --gpus $n --steps 5 --warmup 2 --batch 64 --gallery-rows 2000000 > "$out/bench_${n}_ranks.json" 2> "$out/bench_${n}_ranks.err" \
    || { tail -30 "$out/bench_${n}_ranks.err"; exit 1; }
  python3 - "$out/bench_${n}_ranks.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ranks", d["n_gpus"], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "gates", d["gates"], "sane", d["result_sane"])
PY
done
