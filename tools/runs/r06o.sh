#!/bin/bash
# round 6, run o: block-order bits (rev_order) re-checked on this round's kernels, two instances per variant
mkdir -p gpurun_out/r06o
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "rev_order=0" "rev_order=12" "rev_order=14" "" "rev_order=0" "rev_order=12" "rev_order=14" > gpurun_out/r06o/c5.log 2>&1; cut -c1-200 gpurun_out/r06o/c5.log | grep best
tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "rev_order=0" "" "rev_order=0" > gpurun_out/r06o/c4.log 2>&1; cut -c1-200 gpurun_out/r06o/c4.log | grep best
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 2 "" "rev_order=0" "rev_order=15" "" "rev_order=0" "rev_order=15" > gpurun_out/r06o/c2.log 2>&1; cut -c1-200 gpurun_out/r06o/c2.log | grep best
