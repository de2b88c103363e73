mkdir -p gpurun_out
python tools/probe/v2_check.py > gpurun_out/r05k_v2check.log 2>&1
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "tw_lane=0" "" "tw_lane=0" "" "gd_v2=0" "gd_v2=0,tw_lane=0" > gpurun_out/r05k_c3.log 2>&1
tools/gpu_ab.sh admm 1 3040 4056 3 1 40 2 "tw_lane=0" "" "tw_lane=0" "" > gpurun_out/r05k_c2.log 2>&1
tools/gpu_ab.sh admm 1 270 480 3 64 20 2 "tw_lane=0" "" "tw_lane=0" "" > gpurun_out/r05k_c4.log 2>&1
tools/gpu_ab.sh admm 1 270 480 3 1 5 20 "tw_lane=0" "" "tw_lane=0" "" > gpurun_out/r05k_c1.log 2>&1
tools/gpu_ab.sh admm 16 1080 1920 3 1 20 1 "tw_lane=0" "" > gpurun_out/r05k_c5.log 2>&1
cat gpurun_out/r05k_v2check.log; grep best gpurun_out/r05k_c3.log gpurun_out/r05k_c2.log gpurun_out/r05k_c4.log gpurun_out/r05k_c1.log gpurun_out/r05k_c5.log
