#!/bin/bash
# A/B an environment variable inside ONE gpurun call (boxes differ by a few per cent):
#   bash tools/ab.sh SPLAT_SCAN_THREADS "1024 512 256" [repeats] [extra bench args]
var=$1; vals=$2; reps=${3:-3}; shift 3
for r in $(seq $reps); do
  for v in $vals; do
    out=$(env $var=$v timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | head -1)
    echo "$var=$v $out"
  done
done
