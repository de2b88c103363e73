mkdir -p gpurun_out/r04e
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 1 "" "col_single=1,mid_rad=16.16.24" "col_single=1,seq_t=1,mid_twg=1" "col_single=1,seq_t=1,mid_twg=1,mid_rad=16.16.24" "col_single=1,mid_twg=1,mid_rad=16.16.24" > gpurun_out/r04e/ab_single.log 2>&1
grep best gpurun_out/r04e/ab_single.log | cut -c1-250
