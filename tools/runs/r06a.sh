#!/bin/bash
# round 6, run a: pair-line spectra + XCD group order of the fused forward rows, C4 (64 frames and the 8-frame shard)
mkdir -p gpurun_out/r06a
python -m pytest tests/test_parity_small.py -m gpu -x -q -k "c4_sequential or backward_grid" > gpurun_out/r06a/tests.log 2>&1
tail -3 gpurun_out/r06a/tests.log
tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "spec_lay=0" "k1_group=4" "k1_group=8" "k1_group=16" "k1_group=32" "spec_lay=0,k1_group=16" > gpurun_out/r06a/c4_64.log 2>&1
cat gpurun_out/r06a/c4_64.log
tools/gpu_ab.sh admm 1 270 480 3 8 20 10 "" "spec_lay=0" "k1_group=4" "k1_group=8" "k1_group=16" "k1_group=32" > gpurun_out/r06a/c4_8.log 2>&1
cat gpurun_out/r06a/c4_8.log
