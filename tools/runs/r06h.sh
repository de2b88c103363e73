#!/bin/bash
# round 6, run h: two trees on one box -- 16-byte spectrum accesses in the paired-row tangling (pair lines) against HEAD
mkdir -p gpurun_out/r06h
python -m pytest tests/test_parity_small.py tests/test_parity_fullsize.py -m gpu -x -q -k "c4 or pair_line" > gpurun_out/r06h/tests.log 2>&1; grep -E "passed|failed" gpurun_out/r06h/tests.log | tail -1
tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 64 20 3 > gpurun_out/r06h/c4_64.log 2>&1; cut -c1-160 gpurun_out/r06h/c4_64.log
tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 8 20 10 > gpurun_out/r06h/c4_8.log 2>&1; cut -c1-160 gpurun_out/r06h/c4_8.log
tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 1 5 20 > gpurun_out/r06h/c1.log 2>&1; cut -c1-160 gpurun_out/r06h/c1.log
