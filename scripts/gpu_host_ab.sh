#!/bin/bash
# Same-box comparison of builds of the library under scripts/host_path_rate.py (one caller, then two):
#   scripts/gpu_host_ab.sh <tag> <rounds> <other.so>...
out=gpurun_out/${1:-hostab}
rounds=${2:-2}
shift 2
mkdir -p "$out"
lib=face-recognition_amd/frhot/libfrhot.so
cp "$lib" "$out/v0.so" || exit 1
n=0
for o in "$@"; do n=$((n+1)); cp "$o" "$out/v$n.so" || exit 1; done
: > "$out/host.txt"
for r in $(seq 1 "$rounds"); do
  for v in $(seq 0 $n); do
    cp "$out/v$v.so" "$lib"
    echo "variant $v round $r" >> "$out/host.txt"
    timeout -k 10 200 python scripts/host_path_rate.py 256 10000000 512 60 2 >> "$out/host.txt" 2> "$out/err.txt" || { tail -20 "$out/err.txt"; cp "$out/v0.so" "$lib"; rm -f "$out"/v?.so; exit 1; }
  done
done
cp "$out/v0.so" "$lib"
rm -f "$out"/v?.so
cat "$out/host.txt"
