#!/bin/bash
# round 4's HEAD (4960eb0, built under _ab_r04) against this round's last commit, same box, alternating processes
out=gpurun_out/r05zc; mkdir -p $out
{
echo "== C2 ADMM 12 MP, 40 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 3040 4056 3 1 40 1
echo "== C3 FISTA 12 MP, 40 it"; tools/gpu_ab_versions.sh _ab_r04 fista 1 3040 4056 3 1 40 1
echo "== C4 64 frames, 20 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 64 20 2
echo "== C4 shard 8 frames, 20 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 8 20 5
echo "== C1, 5 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 1 5 50
echo "== 380x507x3, 5 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 380 507 3 1 5 50
echo "== 1520x2028x3, 20 it"; tools/gpu_ab_versions.sh _ab_r04 admm 1 1520 2028 3 1 20 2
echo "== FISTA 1080p, 20 it"; tools/gpu_ab_versions.sh _ab_r04 fista 1 1080 1920 3 1 20 3
} > $out/ab.log 2>&1
grep -v "^    " $out/ab.log | cut -c1-230
