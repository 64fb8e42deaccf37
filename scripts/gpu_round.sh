#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench (with CPU baseline), kernel-trace profile and
# two PMC passes of the bench.  Every GPU step has its own time limit; the chain stops at
# the first failure.  Outputs land under gpurun_out/$1.
set -o pipefail
tag=${1:-run}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { tail -30 "$out/smoke.log"; exit 1; }
tail -1 "$out/smoke.log"
timeout -k 10 400 python -u bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
cat "$out/bench.json"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-scan4096 --no-scan-extra > "$out/trace.log" 2>&1 || { tail -30 "$out/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg --no-scan4096 --no-scan-extra --no-detector > "$out/pmc_fetch.log" 2>&1 || { tail -30 "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg --no-scan4096 --no-scan-extra --no-detector > "$out/pmc_write.log" 2>&1 || { tail -30 "$out/pmc_write.log"; exit 1; }
# the scan-only shapes that share a kernel with another workload: counter passes of their own
for shape in rank8 10m_q64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d="$out/pmc_${c%_SIZE}_$shape"
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d "$d" -- python3 bench.py --scan-shape-only $shape > "$d.log" 2>&1 || { tail -30 "$d.log"; exit 1; }
  done
done
# the headline step held for 2,000 steps (clocks under sustained load)
timeout -k 10 300 python -u bench.py --steps 2000 --warmup 3 --no-cpu-baseline --no-host-leg --no-scan64 > "$out/bench_2000_steps.json" 2> "$out/bench_2000_steps.err" || { tail -30 "$out/bench_2000_steps.err"; exit 1; }
find "$out/trace" -name '*kernel_stats.csv' | head -1 | xargs cat
