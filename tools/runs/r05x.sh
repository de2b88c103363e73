#!/bin/bash
out=gpurun_out/r05x; mkdir -p $out
{
for b in 2 4 6 8; do echo "== $b frames 270x480x3, 20 it"; tools/gpu_ab.sh admm 1 270 480 3 $b 20 5 "k1_rows=0" "k1_rows=1" "k1_rows=0" "k1_rows=1" 2>&1 | grep -A1 best | cut -c1-300; done
echo "== 512x512x3"; tools/gpu_ab.sh admm 1 512 512 3 1 5 30 "k1_rows=0" "k1_rows=1" 2>&1 | grep -A1 best | cut -c1-300
echo "== 128x128x3"; tools/gpu_ab.sh admm 1 128 128 3 1 5 50 "k1_rows=0" "k1_rows=1" 2>&1 | grep -A1 best | cut -c1-300
} > $out/ab.log 2>&1
grep -v "^    " $out/ab.log
(time python -m pytest tests -m gpu -q -x --durations=8 > $out/gputests.log 2>&1); tail -15 $out/gputests.log
