mkdir -p gpurun_out/r04g
{
echo "== C4 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 64 20 3
echo "== C4 pre"; tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "mid_pre=1"
echo "== C4 shard8 pre"; tools/gpu_ab.sh admm 1 270 480 3 8 20 5 "" "mid_pre=1"
echo "== C2 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 3040 4056 3 1 40 1
echo "== C3 versions"; tools/gpu_ab_versions.sh _ab_r03 fista 1 3040 4056 3 1 40 1
echo "== C5 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 16 1080 1920 3 1 20 1
echo "== C1 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 1 5 20
} > gpurun_out/r04g/ab.log 2>&1
grep "==\|best" gpurun_out/r04g/ab.log | cut -c1-220
