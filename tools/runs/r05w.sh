#!/bin/bash
out=gpurun_out/r05w; mkdir -p $out
{
echo "== C1 5 it: column tile width / lanes of the LDS middle"
for o in "" "col_t=4" "col_t=4,mid_nt=512" "col_t=6" "col_t=5" "col_t=4,mid_rad=30.18,mid_nt=512" "mid_seq=1" "mid_seq=1,seq_t=8,mid_nt=512" ""; do tools/gpu_ab.sh admm 1 270 480 3 1 5 50 "$o" 2>&1 | grep -A1 best | cut -c1-330; done
} > $out/ab.log 2>&1
cat $out/ab.log
