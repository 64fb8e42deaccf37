#!/bin/bash
# The two-stream conv stack cut into more parts, with shares of the CUs per part for the early layers (the list is run twice).
out=gpurun_out/${1:-parts}
mkdir -p "$out"
: > "$out/parts.txt"
for rep in 1 2; do
while read -r parts blocks; do
  KBENCH_SYNC=1 KBENCH_PARTS=$parts KBENCH_BLOCKS=$blocks timeout -k 10 120 ./scripts/kbench stack2 256 > "$out/one.txt" 2>&1 || { tail -20 "$out/one.txt"; exit 1; }
  echo "parts $parts blocks $blocks: $(grep -o 'fused: [0-9.]* ms' "$out/one.txt")" >> "$out/parts.txt"
done <<'L'
2 128,128,128,128,128,128,0,0,0,0,0,0
4 64,64,64,64,64,64,0,0,0,0,0,0
4 64,64,64,64,64,64,64,64,64,0,0,0
4 64,64,64,64,64,64,128,128,128,0,0,0
4 128,128,128,128,128,128,0,0,0,0,0,0
4 64,64,64,64,64,64,64,64,64,64,64,64
8 32,32,32,32,32,32,0,0,0,0,0,0
8 32,32,32,32,32,32,64,64,64,0,0,0
L
done
cat "$out/parts.txt"
