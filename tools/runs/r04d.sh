mkdir -p gpurun_out/r04d
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 1 "" "col_single=1" "col_single=1,mid_rad=16.16.24" "col_single=1,mid_nt=768" "col_single=1,seq_t=1" "col_single=1,seq_tiles_first=1" > gpurun_out/r04d/ab_single.log 2>&1
cat gpurun_out/r04d/ab_single.log | cut -c1-400
