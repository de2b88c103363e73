mkdir -p gpurun_out
L=gpurun_out/r05r_old_vs_new.log
echo "== C4 64 frames, 20 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 64 20 2 "" >> $L 2>&1
echo "== C4 shard 8 frames, 20 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 8 20 5 "" >> $L 2>&1
echo "== C1, 5 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 1 270 480 3 1 5 20 "" >> $L 2>&1
echo "== C5 16 planes, 20 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 16 1080 1920 3 1 20 1 "" >> $L 2>&1
echo "== 380x507x3, 5 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 1 380 507 3 1 5 20 "" >> $L 2>&1
echo "== FISTA 1080x1920x3, 60 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 fista 1 1080 1920 3 1 60 2 "" >> $L 2>&1
echo "== C3 FISTA 12 MP, 40 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 fista 1 3040 4056 3 1 40 2 "" >> $L 2>&1
echo "== C2 ADMM 12 MP, 40 it" >> $L; tools/gpu_ab_versions.sh _ab_r04 admm 1 3040 4056 3 1 40 2 "" >> $L 2>&1
cat $L | cut -c1-260
