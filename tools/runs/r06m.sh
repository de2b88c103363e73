#!/bin/bash
# round 6, run m: radices of the paired 3840-point rows (C5 on paired rows with the TV / W half inside)
mkdir -p gpurun_out/r06m
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "row_rad=8.8.6.10" "row_rad=8.8.4.15" "row_rad=8.16.30" "row_rad=8.24.20" "row_rad=8.30.16" "rows_half=1" > gpurun_out/r06m/c5.log 2>&1; cut -c1-230 gpurun_out/r06m/c5.log | grep -E "best"
tools/gpu_ab.sh admm 1 1520 2028 3 1 40 3 "" "row_rad=16.16.16" "row_rad=8.16.32" "rows_half=1" > gpurun_out/r06m/c1520.log 2>&1; cut -c1-230 gpurun_out/r06m/c1520.log | grep -E "best"
python -m pytest tests/test_longrun_pins.py tests/test_parity_fullsize.py -m gpu -x -q -s -k "c5" 2>&1 | grep -E "it [0-9]+:|passed|failed|C5" | cut -c1-260
