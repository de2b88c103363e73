mkdir -p gpurun_out
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "" "lds_pad=4096" "lds_pad=12288" "" "lds_pad=4096" "lds_pad=12288" > gpurun_out/r05h_ldspad.log 2>&1
tools/gpu_ab.sh admm 1 270 480 3 64 20 2 "" "seq_pair=1" "" "seq_pair=1" "" "seq_pair=1" > gpurun_out/r05h_c4pair.log 2>&1
tools/gpu_ab.sh admm 1 270 480 3 8 20 5 "" "seq_pair=1" "" "seq_pair=1" > gpurun_out/r05h_c4shard.log 2>&1
tools/gpu_pmc_mem.sh r05h_c4mem "k_cols_mid" admm 1 270 480 3 64 20 1 "" "seq_pair=1" > gpurun_out/r05h_c4mem.log 2>&1
grep "best" gpurun_out/r05h_ldspad.log gpurun_out/r05h_c4pair.log gpurun_out/r05h_c4shard.log; cat gpurun_out/r05h_c4mem.log
