#!/bin/bash
# round 6, run g: the sequential middle with its point-wise constants precombined (k_mid_consts)
mkdir -p gpurun_out/r06g
python -m pytest tests/test_parity_small.py tests/test_longrun_pins.py tests/test_parity_fullsize.py -m gpu -x -q -k "c4 or pair_line or unrolled" > gpurun_out/r06g/tests.log 2>&1; tail -2 gpurun_out/r06g/tests.log
tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "mid_pc=0" > gpurun_out/r06g/c4_64.log 2>&1
cut -c1-200 gpurun_out/r06g/c4_64.log | grep best
tools/gpu_ab.sh admm 1 270 480 3 8 20 10 "" "mid_pc=0" > gpurun_out/r06g/c4_8.log 2>&1
cut -c1-200 gpurun_out/r06g/c4_8.log | grep best
tools/gpu_ab.sh admm 1 270 480 3 16 20 10 "" "mid_pc=0" > gpurun_out/r06g/c4_16.log 2>&1
cut -c1-200 gpurun_out/r06g/c4_16.log | grep best
tools/gpu_pmc_mem.sh r06g/mem "k_cols_mid" admm 1 270 480 3 64 20 1 "" "mid_pc=0" > gpurun_out/r06g/mem.log 2>&1
grep "k_cols_mid" gpurun_out/r06g/mem.log | cut -c1-200
