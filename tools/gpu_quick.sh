#!/bin/bash
# usage: tools/gpu_quick.sh <tag> -- on the GPU box: GPU tests + a short headline bench (no CPU legs) under gpurun_out/<tag>/
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export MPLBACKEND=Agg
(time python -m pytest tests -m gpu -q -x --durations=8 > $out/gputests.log 2>&1); tail -15 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
(time python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.log); tail -2 $out/bench.log
python tools/kernel_summary.py $out/bench.json; python tools/other_summary.py $out/bench.json
ls lenslesspicam_amd/_lib/modules | wc -l
