#!/bin/bash
# round 6, run l: where do half-length rows + the tiled kernel overtake paired rows with the TV / W half inside?
mkdir -p gpurun_out/r06l
for shp in "2160 2880" "3000 3000" "2000 2500" "1080 2560"; do
  set -- $shp
  tools/gpu_ab.sh admm 1 $1 $2 3 1 20 3 "" "rows_half=0" > gpurun_out/r06l/s$1x$2.log 2>&1; echo "$1x$2x3"; cut -c1-200 gpurun_out/r06l/s$1x$2.log | grep -E "best|padded" | cut -c1-160
done
tools/gpu_ab.sh admm 2 1080 1920 3 1 50 3 "" "rows_half=0" > gpurun_out/r06l/c5share.log 2>&1; echo "C5 share"; cut -c1-200 gpurun_out/r06l/c5share.log | grep best
