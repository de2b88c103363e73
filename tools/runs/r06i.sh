#!/bin/bash
# round 6, run i: three launches per iteration for paired rows of TWO quads per lane (padded widths 1025 ... 2048)
mkdir -p gpurun_out/r06i
tools/gpu_ab.sh admm 1 760 1014 1 1 5 20 "" "k1_rows=2" > gpurun_out/r06i/p760g_5.log 2>&1; cut -c1-260 gpurun_out/r06i/p760g_5.log | grep best
tools/gpu_ab.sh admm 1 760 1014 1 1 100 3 "" "k1_rows=2" > gpurun_out/r06i/p760g_100.log 2>&1; cut -c1-260 gpurun_out/r06i/p760g_100.log | grep best
tools/gpu_ab.sh admm 1 760 1014 3 1 20 5 "" "k1_rows=2" > gpurun_out/r06i/p760rgb.log 2>&1; cut -c1-260 gpurun_out/r06i/p760rgb.log | grep best
tools/gpu_ab.sh admm 1 540 960 3 1 20 5 "" "k1_rows=2" > gpurun_out/r06i/p540.log 2>&1; cut -c1-260 gpurun_out/r06i/p540.log | grep best
tools/gpu_ab.sh admm 1 600 800 3 8 20 3 "" "k1_rows=2" > gpurun_out/r06i/p600b8.log 2>&1; cut -c1-260 gpurun_out/r06i/p600b8.log | grep best
