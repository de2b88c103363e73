#!/bin/bash
# round 6, run c: pins at BASELINE length (C4, C5), C1 with pair lines in the side-by-side middle, run length of the XCD
# order, C5 row plans
mkdir -p gpurun_out/r06c
python -m pytest tests/test_longrun_pins.py tests/test_longrun_fixtures.py -m gpu -x -q -s > gpurun_out/r06c/pins.log 2>&1
grep -E "it [0-9]+:|passed|failed|Error|error" gpurun_out/r06c/pins.log | cut -c1-330
python -m pytest tests/test_parity_fullsize.py -m gpu -x -q -k "c4" > gpurun_out/r06c/c4tests.log 2>&1
tail -3 gpurun_out/r06c/c4tests.log
tools/gpu_ab.sh admm 1 270 480 3 1 5 20 "" "spec_lay=0" > gpurun_out/r06c/c1.log 2>&1
cut -c1-230 gpurun_out/r06c/c1.log
tools/gpu_ab.sh admm 1 270 480 3 1 100 3 "" "spec_lay=0" > gpurun_out/r06c/c1_100.log 2>&1
cut -c1-230 gpurun_out/r06c/c1_100.log
tools/gpu_ab.sh admm 1 380 507 3 1 5 20 "" "spec_lay=0" > gpurun_out/r06c/c380.log 2>&1
cut -c1-230 gpurun_out/r06c/c380.log
tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "k1_group=16" "k1_group=12" "k1_group=20" "k1_group=24" "k1_group=0" > gpurun_out/r06c/c4_64.log 2>&1
cut -c1-200 gpurun_out/r06c/c4_64.log | grep best
tools/gpu_ab.sh admm 1 270 480 3 8 20 10 "k1_group=16" "k1_group=12" "k1_group=20" "k1_group=24" "k1_group=0" > gpurun_out/r06c/c4_8.log 2>&1
cut -c1-200 gpurun_out/r06c/c4_8.log | grep best
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "row_rad=8.8.6.5,row_nt=256" "row_rad=16.8.15,row_nt=256" "row_rad=8.8.6.5,row_nt=128" "row_rad=16.15.8,row_nt=128" > gpurun_out/r06c/c5.log 2>&1
cut -c1-200 gpurun_out/r06c/c5.log | grep best
