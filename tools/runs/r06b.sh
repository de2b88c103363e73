#!/bin/bash
# round 6, run b: pair-line spectra with lane-constant addressing in the sequential middle; HBM traffic of C4's three kernels
mkdir -p gpurun_out/r06b
python -m pytest tests/test_parity_small.py -m gpu -x -q -k "c4_sequential or backward_grid" > gpurun_out/r06b/tests.log 2>&1
tail -3 gpurun_out/r06b/tests.log
tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "spec_lay=0" "k1_group=16" "mid_pre=0" > gpurun_out/r06b/c4_64.log 2>&1
cut -c1-200 gpurun_out/r06b/c4_64.log
tools/gpu_ab.sh admm 1 270 480 3 8 20 10 "" "spec_lay=0" "k1_group=16" "mid_pre=1" > gpurun_out/r06b/c4_8.log 2>&1
cut -c1-200 gpurun_out/r06b/c4_8.log
tools/gpu_ab.sh admm 1 270 480 3 16 20 10 "" "spec_lay=0" > gpurun_out/r06b/c4_16.log 2>&1
cut -c1-200 gpurun_out/r06b/c4_16.log
tools/gpu_pmc_mem.sh r06b/mem "k_" admm 1 270 480 3 64 20 1 "" "spec_lay=0" "k1_group=16" > gpurun_out/r06b/mem.log 2>&1
grep -E "k_cols_mid|k_rfwd|k_rinv" gpurun_out/r06b/mem.log | cut -c1-260
