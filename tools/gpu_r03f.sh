#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
tools/c4_ab.sh "seq_tiles_first=1" "" 3 2>&1 | tee $out/c4_ab.log
tools/gpu_prof.sh r03f_c2 > /dev/null 2>&1; python tools/summarize_prof.py r03f_c2 gpurun_out/r03f_c2 > /dev/null; cp profiles/traffic.json profiles/r03f_c2_counters.md profiles/r03f_c2_kernel_stats.csv $out/ 2>/dev/null
tools/gpu_prof.sh r03f_c4 --config c4 > /dev/null 2>&1; python tools/summarize_prof.py r03f_c4 gpurun_out/r03f_c4 > /dev/null; cp profiles/r03f_c4_counters.md profiles/r03f_c4_kernel_stats.csv $out/ 2>/dev/null
LPC_OPTIONS="seq_tiles_first=1" tools/gpu_prof.sh r03f_c4t --config c4 > /dev/null 2>&1; python tools/summarize_prof.py r03f_c4t gpurun_out/r03f_c4t > /dev/null; cp profiles/r03f_c4t_counters.md $out/ 2>/dev/null
tools/gpu_prof.sh r03f_c3 --algo fista > /dev/null 2>&1; python tools/summarize_prof.py r03f_c3 gpurun_out/r03f_c3 > /dev/null; cp profiles/r03f_c3_counters.md profiles/r03f_c3_kernel_stats.csv $out/ 2>/dev/null
cat profiles/traffic.json | head -50
grep "k_cols_mid_admm_seq" $out/r03f_c4_counters.md $out/r03f_c4t_counters.md
