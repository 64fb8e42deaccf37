#!/bin/bash
# same-box A/B of kbench builds: [AB_MODE=conv] scripts/ab.sh <rounds> <binary>...   (KBENCH_SHORT etc. pass through)
rounds=$1; shift
for r in $(seq $rounds); do
  for b in "$@"; do
    echo "== $b (round $r)"
    timeout -k 5 120 ./scripts/$b ${AB_MODE:-scan} || exit 1
  done
done
