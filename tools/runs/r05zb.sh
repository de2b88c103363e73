#!/bin/bash
# XCD-aware block order of the forward rows that hold the TV / W half: this tree against the tree with it
out=gpurun_out/r05zb; mkdir -p $out
{
echo "== C4 64 frames"; tools/gpu_ab_versions.sh _ab_new admm 1 270 480 3 64 20 2
echo "== 8 frames"; tools/gpu_ab_versions.sh _ab_new admm 1 270 480 3 8 20 5
echo "== C1"; tools/gpu_ab_versions.sh _ab_new admm 1 270 480 3 1 5 50
echo "== 380x507"; tools/gpu_ab_versions.sh _ab_new admm 1 380 507 3 1 5 50
} > $out/ab.log 2>&1
cut -c1-200 $out/ab.log
cd _ab_new; GRAFT_REPO_ROOT=$PWD python -m pytest tests/test_parity_small.py -q -m gpu -k "tv_half" 2>&1 | tail -2
