mkdir -p gpurun_out
python tools/probe/v2_check.py > gpurun_out/r05b_v2check.log 2>&1
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "gd_v2=0" "" "row_lay=0" "row_lay=3" "gd_v2=0" "" "row_lay=0" "row_lay=3" "row_lay=2" > gpurun_out/r05b_ab_c3.log 2>&1
tools/gpu_pmc_mem.sh r05b_c4mem "k_" admm 1 270 480 3 64 20 1 "" "rev_order=0" "mid_pre=0" "hv_full=1,xi_full=1" > gpurun_out/r05b_c4mem.log 2>&1
cat gpurun_out/r05b_v2check.log gpurun_out/r05b_ab_c3.log; grep "k_cols_mid\|k_rfwd_arr\|k_rinv_arr\|k_admm_spatial" gpurun_out/r05b_c4mem.log
