#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
{
echo "== 760x1014 gray ADMM (paired 2048)"; python tools/probe/ab_probe.py admm 1 760 1014 1 1 20 5 "" "row_rad=16.16.8" "row_rad=16.16.8,row_nt=256" "row_rad=16.8.16"
echo "== 760x1014 gray FISTA (half 1024)"; python tools/probe/ab_probe.py fista 1 760 1014 1 1 60 3 "" "row_rad=16.8.8" "row_rad=16.8.8,row_nt=128" "row_rad=16.16.4,row_nt=128"
echo "== 1520x2028 ADMM (half 2048)"; python tools/probe/ab_probe.py admm 1 1520 2028 3 1 50 2 "" "row_rad=16.16.8" "row_rad=16.16.8,row_nt=256" "passa_rad=16.8"
echo "== C2 passA 16.8"; python tools/probe/ab_probe.py admm 1 3040 4056 3 1 40 2 "" "passa_rad=16.8" "passa_rad=8.16"
echo "== C3 fista passA"; python tools/probe/ab_probe.py fista 1 3040 4056 3 1 40 2 "" "passa_rad=16.8"
echo "== 380x507 ADMM (paired 1024)"; python tools/probe/ab_probe.py admm 1 380 507 3 1 5 20 "" "row_rad=16.8.8" "row_rad=16.16.4"
echo "== C5 ADMM D=4 (half 1920, passA 90)"; python tools/probe/ab_probe.py admm 4 1080 1920 3 1 20 2 "" "passa_rad=10.9" "passa_rad=9.10" "passa_rad=18.5" "passa_rad=30.3" "row_rad=8.8.30"
echo "== C4 rows 960"; python tools/probe/ab_probe.py admm 1 270 480 3 64 20 3 "" "row_rad=8.4.30" "mid_rad=10.6.9" "mid_rad=9.10.6" "mid_rad=6.9.10"
} 2>&1 | grep -v "^$\|amdgpu.ids\|^    padded" | tee $out/ab.log
