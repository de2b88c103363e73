#!/bin/bash
out=gpurun_out/r05zd; mkdir -p $out
{
for b in 8 16 64; do
  reps=5; [ $b = 64 ] && reps=2
  echo "== $b frames 270x480x3, 20 it: which middle"
  for o in "" "mid_seq=0" "mid_seq=0,mid_nt=1024,mid_rad=6.10.9" "mid_seq=0,mid_nt=512,mid_rad=6.10.9" ""; do tools/gpu_ab.sh admm 1 270 480 3 $b 20 $reps "$o" 2>&1 | grep -A1 best | cut -c1-400; done
done
} > $out/ab.log 2>&1
grep -A1 best $out/ab.log | grep -v "^--" | sed 's/.*columns:/   columns:/' | cut -c1-200
