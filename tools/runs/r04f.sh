mkdir -p gpurun_out/r04f
{
echo "== C2 ADMM 40 it"; tools/gpu_ab.sh admm 1 3040 4056 3 1 40 1 "row_lay=1" "" "row_lay=0"
echo "== C3 FISTA 40 it"; tools/gpu_ab.sh fista 1 3040 4056 3 1 40 1 "row_lay=1" "" "row_lay=0"
echo "== C4"; tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "row_lay=1" "" "row_lay=0"
echo "== C1"; tools/gpu_ab.sh admm 1 270 480 3 1 5 20 "row_lay=1" "" "row_lay=0"
echo "== C5"; tools/gpu_ab.sh admm 16 1080 1920 3 1 20 1 "row_lay=1" "" "row_lay=0"
} > gpurun_out/r04f/ab_rowlay.log 2>&1
grep "==\|best" gpurun_out/r04f/ab_rowlay.log | cut -c1-230
