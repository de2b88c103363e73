mkdir -p gpurun_out
python tools/probe/v2_check.py > gpurun_out/r05d_v2check.log 2>&1
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "gd_v2=0" "" "gd_v2=0" "" "gd_v2=0" "" > gpurun_out/r05d_ab_c3.log 2>&1
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 2 "" "" > gpurun_out/r05d_ab_c2.log 2>&1
tools/gpu_ab.sh admm 1 270 480 3 64 20 2 "" "seq_pair=1" > gpurun_out/r05d_ab_c4.log 2>&1
cat gpurun_out/r05d_v2check.log gpurun_out/r05d_ab_c3.log gpurun_out/r05d_ab_c2.log gpurun_out/r05d_ab_c4.log
