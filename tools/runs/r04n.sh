mkdir -p gpurun_out/r04n
{
echo "== C2"; tools/gpu_ab.sh admm 1 3040 4056 3 1 40 1 "" "k1_half=0"
echo "== C4"; tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "k1_half=0"
echo "== C5"; tools/gpu_ab.sh admm 16 1080 1920 3 1 20 1 "" "k1_half=0"
echo "== C1"; tools/gpu_ab.sh admm 1 270 480 3 1 5 20 "" "k1_half=0"
echo "== 1520x2028"; tools/gpu_ab.sh admm 1 1520 2028 3 1 50 2 "" "k1_half=0"
} > gpurun_out/r04n/ab.log 2>&1
grep "==\|best" gpurun_out/r04n/ab.log | cut -c1-220
python -m pytest tests/test_parity_fullsize.py tests/test_parity_large.py tests/test_norm_scale.py -m gpu -q -x -k "c4 or c1_admm or c2_admm_5 or c5_one or forward_l2" > gpurun_out/r04n/tests.log 2>&1; tail -3 gpurun_out/r04n/tests.log
