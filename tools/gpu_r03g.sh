#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
{
echo "== C4 seq order"; python tools/probe/ab_probe.py admm 1 270 480 3 64 20 3 "" "seq_tiles_first=1"
echo "== C1 single frame"; python tools/probe/ab_probe.py admm 1 270 480 3 1 5 20 "" "col_t=4" "mid_seq=1" "prow_nt128=1" "hv_full=1" "rows_half=1"
echo "== C1 FISTA"; python tools/probe/ab_probe.py fista 1 270 480 3 1 60 5 "" "rows_half=1"
echo "== 380x507"; python tools/probe/ab_probe.py admm 1 380 507 3 1 5 20 "" "col_t=4" "mid_seq=1" "rows_half=1" "hv_full=1"
echo "== 380x507 FISTA"; python tools/probe/ab_probe.py fista 1 380 507 3 1 60 5 "" "rows_half=1"
echo "== 760x1014 gray ADMM"; python tools/probe/ab_probe.py admm 1 760 1014 1 1 5 20 "" "rows_half=1" "hv_full=1"
echo "== 1520x2028"; python tools/probe/ab_probe.py admm 1 1520 2028 3 1 50 2 "" "rows_half=0" "hv_full=1"
} 2>&1 | grep -v "^$" | tee $out/ab.log
