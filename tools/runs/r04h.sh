mkdir -p gpurun_out/r04h
{
echo "== C4 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 64 20 3
echo "== C4 pre"; tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "mid_pre=1" "mid_pre=1,seq_t=16" "seq_t=16"
echo "== C4 shard8 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 8 20 5
echo "== C4 shard8 pre"; tools/gpu_ab.sh admm 1 270 480 3 8 20 5 "" "mid_pre=1"
} > gpurun_out/r04h/ab.log 2>&1
grep "==\|best" gpurun_out/r04h/ab.log | cut -c1-220
