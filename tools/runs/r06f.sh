#!/bin/bash
# round 6, run f: the TV / W half inside the half-length forward rows of wide frames (k_rfwd_half_x<.., K1>): r_sp never stored
mkdir -p gpurun_out/r06f
python -m pytest tests/test_parity_small.py -m gpu -x -q -k "admm_matches_reference_golden or unusual" > gpurun_out/r06f/tests.log 2>&1; tail -2 gpurun_out/r06f/tests.log
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 2 "" "k1_rows=2" "k1_rows=2,k1_group=8" "k1_rows=2,k1_group=32" "k1_rows=2,k1_group=64" "k1_rows=2,k1_group=0" > gpurun_out/r06f/c2.log 2>&1
cut -c1-220 gpurun_out/r06f/c2.log | grep best
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "k1_rows=2" "k1_rows=2,k1_group=32" > gpurun_out/r06f/c5.log 2>&1
cut -c1-220 gpurun_out/r06f/c5.log | grep best
tools/gpu_ab.sh admm 1 1520 2028 3 1 40 3 "" "k1_rows=2" "k1_rows=2,k1_group=32" > gpurun_out/r06f/c1520.log 2>&1
cut -c1-220 gpurun_out/r06f/c1520.log | grep best
tools/gpu_pmc_mem.sh r06f/mem "k_rfwd_half_x" admm 1 3040 4056 3 1 40 1 "k1_rows=2" "k1_rows=2,k1_group=32" "k1_rows=2,k1_group=0" > gpurun_out/r06f/mem.log 2>&1
grep "k_rfwd" gpurun_out/r06f/mem.log | cut -c1-200
