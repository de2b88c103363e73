#!/bin/bash
out=gpurun_out/r05za; mkdir -p $out
{
echo "== C4 64 frames: step 3 of the sequential middle in batches"
for o in "" "module_dir=_ab_x/knockb2" "module_dir=_ab_x/knockb3" "module_dir=_ab_x/knockb5" "module_dir=_ab_x/knockb9" "" "module_dir=_ab_x/knockb3"; do tools/gpu_ab.sh admm 1 270 480 3 64 20 2 "$o" 2>&1 | grep best | cut -c1-200; done
} > $out/ab.log 2>&1
cat $out/ab.log
LPC_STAMP_CHILD=1 LPC_STAMP_SO=$(ls $PWD/_ab_x/knocksb3/*.so) python tools/stamp_timeline.py 1 270 480 3 64 20 > $out/stamps_c4_b3.log 2>&1; tail -7 $out/stamps_c4_b3.log | cut -c1-700
