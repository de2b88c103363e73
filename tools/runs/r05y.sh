#!/bin/bash
# batches: the TV / W half inside 256-lane paired rows against the tiled kernel + 128-lane rows
out=gpurun_out/r05y; mkdir -p $out
{
for b in 8 16 64; do
  reps=5; [ $b = 64 ] && reps=2
  echo "== $b frames 270x480x3, 20 it"
  for o in "" "prow_nt128=0,k1_rows=0" "prow_nt128=0,k1_rows=1" "" "prow_nt128=0,k1_rows=1"; do tools/gpu_ab.sh admm 1 270 480 3 $b 20 $reps "$o" 2>&1 | grep -A1 best | cut -c1-420; done
done
} > $out/ab.log 2>&1
grep best $out/ab.log | cut -c1-200
