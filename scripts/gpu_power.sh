#!/bin/bash
# Power, clocks and temperature sampled twice a second while the embed pipeline, then the 256-query scan, loop (developer tool).
out=gpurun_out/${1:-power}
mkdir -p "$out"
sample() {   # $1 = file, $2 = samples
  for i in $(seq 1 "$2"); do
    rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|Temperature \(Sensor junction\)|Temperature \(Sensor memory\)" | sed -e 's/^GPU\[0\][[:space:]]*: //' | tr '\n' ';'
    echo
    sleep 0.5
  done > "$1" 2>&1
}
rocm-smi --showmaxpower --showpower 2>&1 | grep -E "Power|power" > "$out/cap.txt"
sample "$out/smi_embed.txt" 40 &
smi=$!
timeout -k 10 120 python scripts/embed_loop.py 256 2000 > "$out/embed.txt" 2>&1 || { tail "$out/embed.txt"; kill $smi; exit 1; }
wait $smi
sample "$out/smi_scan.txt" 30 &
smi=$!
timeout -k 10 200 python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-host-leg --no-scan64 --no-scan4096 --no-scan-extra --no-detector > "$out/bench.json" 2> "$out/bench.err" || { tail "$out/bench.err"; kill $smi; exit 1; }
wait $smi
cat "$out/cap.txt" "$out/embed.txt"; echo embed; cut -c1-330 "$out/smi_embed.txt" | sed -n '5,40p;' | awk 'NR%3==0'; echo bench; cut -c1-330 "$out/smi_scan.txt" | awk 'NR%3==0'
