#!/bin/bash
# usage: tools/gpu_evidence.sh <tag> -- on the GPU box: the evidence set of a commit under gpurun_out/<tag>/
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export MPLBACKEND=Agg
(time python -m pytest tests -m gpu -q --durations=8 > $out/gputests.log 2>&1); tail -4 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
(time python bench.py > $out/bench.json 2> $out/bench.log); tail -2 $out/bench.log
python tools/kernel_summary.py $out/bench.json; python tools/other_summary.py $out/bench.json
python bench.py --algo fista --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/bench_fista.json 2> $out/bench_fista.log
python tools/kernel_summary.py $out/bench_fista.json
python tools/profile_admm.py > $out/profile.log 2>&1; python tools/profile_admm.py --algo fista >> $out/profile.log 2>&1; python tools/profile_admm.py --raw >> $out/profile.log 2>&1; cat $out/profile.log
tools/gpu_prof.sh $tag > /dev/null 2>&1
ls $out | tr '\n' ' '
