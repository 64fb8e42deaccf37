#!/bin/bash
# The first two conv layers as one launch against two: bit equality at two batch sizes, timings alone and
# inside the two-stream conv stack (one block per CU, and the library's shares of the CUs for the early layers).
out=gpurun_out/${1:-fuse}
mkdir -p "$out"
timeout -k 10 120 ./scripts/kbench fuse 3 > "$out/fuse.txt" 2>&1 || { tail -20 "$out/fuse.txt"; exit 1; }
timeout -k 10 120 ./scripts/kbench fuse 256 >> "$out/fuse.txt" 2>&1 || { tail -20 "$out/fuse.txt"; exit 1; }
for blocks in 0,0,0,0,0,0,0,0,0,0,0,0 128,128,128,128,128,128,0,0,0,0,0,0; do
  echo "blocks per launch $blocks" >> "$out/fuse.txt"
  KBENCH_SYNC=1 KBENCH_BLOCKS=$blocks timeout -k 10 120 ./scripts/kbench stack2 256 >> "$out/fuse.txt" 2>&1 || { tail -20 "$out/fuse.txt"; exit 1; }
done
cat "$out/fuse.txt"
