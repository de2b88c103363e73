mkdir -p gpurun_out
tools/gpu_pmc_ab.sh r05c_pmc "k_gd_\|k_rinv_gd" fista 1 3040 4056 3 1 20 1 "gd_v2=0" "row_lay=0" "row_lay=3" > gpurun_out/r05c_pmc.log 2>&1
cat gpurun_out/r05c_pmc.log
