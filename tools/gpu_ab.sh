#!/bin/bash
# usage (GPU box): tools/gpu_ab.sh ALGO D H W C B N_ITER REPS "opts_a" "opts_b" ...
# Same-box A/B of launch-plan option strings (include/lpc.h, lpc_config.options) on one workload: every variant is run
# five times, interleaved; prints best / median ms per call and the HIP-event kernel table (tools/probe/ab_probe.py).
# A variant that names another plan (radices, tile widths, ...) costs one plan module: ~3 s of hipcc on first use.
#   tools/gpu_ab.sh admm 1 270 480 3 64 20 3 "" "spec_lay=0" "k1_group=0"   # C4: plain-row spectra, launch-order rows
#   tools/gpu_ab.sh admm 16 1080 1920 3 1 20 2 "" "row_rad=8.8.6.5"            # C5, row radices
python tools/probe/ab_probe.py "$@" 2>&1 | grep -v "amdgpu.ids"
