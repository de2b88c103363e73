#!/bin/bash
# usage (GPU box): tools/c4_ab.sh "ENV_A" "ENV_B" [reps] -- alternates two environments on bench.py's C4 leg (same box)
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for v in "$A" "$B"; do
    echo "[$v]"
    env $v python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/other_summary.py /dev/stdin | grep -E "C4|C1"
  done
done
