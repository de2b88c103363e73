#!/bin/bash
# usage (GPU box): tools/c4_ab.sh "OPTS_A" "OPTS_B" [reps] -- alternates two launch-plan option strings (include/lpc.h) on
# the C4 workload, same box
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for v in "$A" "$B"; do python tools/probe/c4_probe.py "$v" 2>/dev/null | tail -1; done
done
