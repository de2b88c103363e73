mkdir -p gpurun_out/r04l
{
echo "== C1"; tools/gpu_ab.sh admm 1 270 480 3 1 5 20 "" "col_t=4" "mid_seq=1" "mid_seq=1,seq_t=4" "mid_rad=6.10.9" "col_t=4,mid_rad=6.10.9" "col_t=4,mid_nt=256" "col_t=4,mid_rad=6.10.9,mid_nt=256" "mid_seq=1,seq_t=4,mid_pre=1"
echo "== C1 versions"; tools/gpu_ab_versions.sh _ab_r03 admm 1 270 480 3 1 5 20
echo "== 380x507"; tools/gpu_ab.sh admm 1 380 507 3 1 5 20 "" "col_t=4"
} > gpurun_out/r04l/ab.log 2>&1
grep "==\|best" gpurun_out/r04l/ab.log | cut -c1-200
