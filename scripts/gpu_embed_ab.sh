#!/bin/bash
# Same-box comparison of builds of the library under scripts/embed_loop.py (the embed pipeline alone):
#   [BATCHES="32 64 256"] [DATA=noise] scripts/gpu_embed_ab.sh <tag> <rounds> <other.so>...
out=gpurun_out/${1:-embedab}
rounds=${2:-2}
shift 2
mkdir -p "$out"
lib=face-recognition_amd/frhot/libfrhot.so
cp "$lib" "$out/v0.so" || exit 1
n=0
for o in "$@"; do n=$((n+1)); cp "$o" "$out/v$n.so" || exit 1; done
: > "$out/embed_loop.txt"
for r in $(seq 1 "$rounds"); do
  for v in $(seq 0 $n); do
    cp "$out/v$v.so" "$lib"
    for batch in ${BATCHES:-256}; do
      echo -n "variant $v: " >> "$out/embed_loop.txt"
      timeout -k 10 120 python scripts/embed_loop.py $batch 40 0 ${DATA:-crops} >> "$out/embed_loop.txt" 2> "$out/err.txt" || { tail -20 "$out/err.txt"; cp "$out/v0.so" "$lib"; rm -f "$out"/v?.so; exit 1; }
    done
  done
done
cp "$out/v0.so" "$lib"
rm -f "$out"/v?.so
cat "$out/embed_loop.txt"
