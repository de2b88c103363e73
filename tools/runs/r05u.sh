#!/bin/bash
# three launches per iteration for small frames (TV / W half inside the forward rows): parity on the GPU, A/B, timeline
out=gpurun_out/r05u; mkdir -p $out
python -m pytest tests/test_parity_small.py -q -m gpu -k "tv_half_inside" > $out/tests.log 2>&1; tail -3 $out/tests.log
{
for shp in "270 480" "380 507" "256 256" "540 960"; do
  echo "== $shp x3, ADMM 5 it"; tools/gpu_ab.sh admm 1 $shp 3 1 5 50 "k1_rows=0" "k1_rows=1" "k1_rows=0" "k1_rows=1" 2>&1 | grep best
done
echo "== C1 100 it"; tools/gpu_ab.sh admm 1 270 480 3 1 100 5 "k1_rows=0" "k1_rows=1" 2>&1 | grep best
echo "== trees C1"; tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 1 5 50
} > $out/ab.log 2>&1
python tools/stamp_timeline.py 1 270 480 3 1 5 "k1_rows=1" > $out/stamps_270x480_k1rows.log 2>&1
python tools/stamp_timeline.py 1 270 480 3 1 5 "k1_rows=0" > $out/stamps_270x480_tiled.log 2>&1
cut -c1-260 $out/ab.log; tail -7 $out/stamps_270x480_k1rows.log | cut -c1-420
