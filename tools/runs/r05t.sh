#!/bin/bash
# branch-free point-wise steps / constants behind the tile loads / one-trip X half: old tree against new, and variants
out=gpurun_out/r05t; mkdir -p $out
{
echo "== C4 64 frames, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 64 20 2
echo "== C4 variants"; for o in "" "module_dir=_ab_x/knockb5" "module_dir=_ab_x/knockb9" "module_dir=_ab_x/knockka" "module_dir=_ab_x/knocktw0" "mid_minw=6" "mid_minw=7"; do tools/gpu_ab.sh admm 1 270 480 3 64 20 2 "$o" 2>&1 | grep best; done
echo "== shard 8 frames, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 8 20 5
echo "== C1, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 270 480 3 1 5 50
echo "== 380x507, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 380 507 3 1 5 50
echo "== C2, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 3040 4056 3 1 40 1
echo "== 1080p / 1520x2028, trees"; tools/gpu_ab_versions.sh _ab_head admm 1 1080 1920 3 1 20 3; tools/gpu_ab_versions.sh _ab_head admm 1 1520 2028 3 1 20 2
} > $out/ab.log 2>&1
python tools/stamp_timeline.py 1 270 480 3 1 5 > $out/stamps_270x480.log 2>&1
grep -v "^    " $out/ab.log | cut -c1-220
tail -8 $out/stamps_270x480.log | cut -c1-400
