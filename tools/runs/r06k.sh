#!/bin/bash
# round 6, run k: WIDE frames on paired rows with the TV / W half inside (two quads per lane) against half-length rows + tiled kernel
mkdir -p gpurun_out/r06k
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 2 "" "rows_half=0" "rows_half=0,k1_rows=0" > gpurun_out/r06k/c2.log 2>&1; cut -c1-250 gpurun_out/r06k/c2.log
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "rows_half=0" "rows_half=0,k1_rows=0" > gpurun_out/r06k/c5.log 2>&1; cut -c1-250 gpurun_out/r06k/c5.log | grep best
tools/gpu_ab.sh admm 1 1520 2028 3 1 40 3 "" "rows_half=0" "rows_half=0,k1_rows=0" > gpurun_out/r06k/c1520.log 2>&1; cut -c1-250 gpurun_out/r06k/c1520.log | grep best
