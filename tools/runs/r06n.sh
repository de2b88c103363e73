#!/bin/bash
# round 6, run n: C5 on paired rows (two radix choices) against half-length rows, TWO instances of every variant in one process
mkdir -p gpurun_out/r06n
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "rows_half=1" "row_rad=8.24.20" "" "rows_half=1" "row_rad=8.24.20" > gpurun_out/r06n/c5.log 2>&1; cut -c1-230 gpurun_out/r06n/c5.log | grep -E "best"
tools/gpu_ab.sh admm 2 1080 1920 3 1 50 3 "" "rows_half=1" "row_rad=8.24.20" "" "rows_half=1" "row_rad=8.24.20" > gpurun_out/r06n/c5share.log 2>&1; cut -c1-230 gpurun_out/r06n/c5share.log | grep -E "best"
tools/gpu_ab.sh admm 1 1520 2028 3 1 40 3 "" "rows_half=1" "" "rows_half=1" "" "rows_half=1" > gpurun_out/r06n/c1520.log 2>&1; cut -c1-230 gpurun_out/r06n/c1520.log | grep -E "best"
