#!/bin/bash
out=gpurun_out/r05v; mkdir -p $out
{
echo "== C1 5 it"; for o in "k1_rows=0" "" "module_dir=_ab_x/knockw7" "module_dir=_ab_x/knockw8" "" "module_dir=_ab_x/knockw7"; do tools/gpu_ab.sh admm 1 270 480 3 1 5 50 "$o" 2>&1 | grep best; done
echo "== C1 100 it"; for o in "k1_rows=0" "" "module_dir=_ab_x/knockw7"; do tools/gpu_ab.sh admm 1 270 480 3 1 100 5 "$o" 2>&1 | grep best; done
echo "== 380 507"; tools/gpu_ab.sh admm 1 380 507 3 1 5 50 "k1_rows=0" "" 2>&1 | grep best
} > $out/ab.log 2>&1
python tools/stamp_timeline.py 1 270 480 3 1 5 > $out/stamps_270x480.log 2>&1
cut -c1-200 $out/ab.log; tail -7 $out/stamps_270x480.log | cut -c1-420
