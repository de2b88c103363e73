mkdir -p gpurun_out/r04i
{
echo "== C4"; tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "seq_pair=0" "mid_pre=0" "mid_pre=0,seq_pair=0"
echo "== C4 shard8"; tools/gpu_ab.sh admm 1 270 480 3 8 20 5 "" "seq_pair=0" 
} > gpurun_out/r04i/ab.log 2>&1
grep "==\|best" gpurun_out/r04i/ab.log | cut -c1-220
