mkdir -p gpurun_out/r04j
{
echo "== C4 fuse1 (tree _ab_x) vs not, both mid_pre=0"; tools/gpu_ab_versions.sh _ab_x admm 1 270 480 3 64 20 3 "mid_pre=0"
echo "== C4 shard8"; tools/gpu_ab_versions.sh _ab_x admm 1 270 480 3 8 20 5 "mid_pre=0"
} > gpurun_out/r04j/ab.log 2>&1
grep "==\|best" gpurun_out/r04j/ab.log | cut -c1-220
