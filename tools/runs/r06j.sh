#!/bin/bash
# round 6, run j: the three-launch plan on the WIDEST paired rows (padded widths 2049 ... 3640): default against k1_rows=0
mkdir -p gpurun_out/r06j
for shp in "1024 1280" "1200 1600" "1500 1800"; do
  set -- $shp
  tools/gpu_ab.sh admm 1 $1 $2 3 1 20 3 "" "k1_rows=0" > gpurun_out/r06j/s$1x$2.log 2>&1; echo "$1x$2x3"; cut -c1-230 gpurun_out/r06j/s$1x$2.log | grep best
done
tools/gpu_ab.sh admm 1 1024 1280 3 4 20 3 "" "k1_rows=0" > gpurun_out/r06j/s1024x1280b4.log 2>&1; echo "4 x 1024x1280x3"; cut -c1-230 gpurun_out/r06j/s1024x1280b4.log | grep best
