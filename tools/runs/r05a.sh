mkdir -p gpurun_out
python -m pytest tests/test_parity_small.py -q -x -m gpu -k "second_form or prefetching_residual or gd_matches" 2>&1 | tail -5 > gpurun_out/r05a_tests.log
tools/gpu_ab.sh fista 1 3040 4056 3 1 40 2 "gd_v2=0" "gd_v2=1" "gd_v2=0" "gd_v2=1" "gd_v2=1,row_lay=2" "gd_v2=1,row_lay=0" > gpurun_out/r05a_ab_c3.log 2>&1
tools/gpu_seq_calib.sh r05a_seqcal > gpurun_out/r05a_seqcal.log 2>&1
cat gpurun_out/r05a_tests.log gpurun_out/r05a_ab_c3.log; tail -40 gpurun_out/r05a_seqcal.log
