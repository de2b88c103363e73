#!/bin/bash
# usage: tools/gpu_evidence.sh <tag> -- on the GPU box: the evidence set of a commit under gpurun_out/<tag>/
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export MPLBACKEND=Agg
(time python -m pytest tests -m gpu -q -s --durations=12 > $out/gputests.log 2>&1); grep -v "^$" $out/gputests.log | grep "C2\|C3\|C5\|passed\|failed\|FAILED" | tail -16
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
# profiles first: the bench line then carries this tree's PMC traffic (profiles/traffic.json is stamped with the source fingerprint)
for cfg in "c2:" "c4:--config c4" "c3:--algo fista"; do
  t=${cfg%%:*}; a=${cfg#*:}
  tools/gpu_prof.sh ${tag}_$t $a > /dev/null 2>&1; python tools/summarize_prof.py ${tag}_$t gpurun_out/${tag}_$t > /dev/null
  cp profiles/${tag}_${t}_counters.md profiles/${tag}_${t}_kernel_stats.csv $out/ 2>/dev/null
done
cp profiles/traffic.json $out/
(time python bench.py > $out/bench.json 2> $out/bench.log); tail -2 $out/bench.log
python tools/kernel_summary.py $out/bench.json; python tools/other_summary.py $out/bench.json
python bench.py --algo fista --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/bench_fista.json 2> $out/bench_fista.log
python tools/kernel_summary.py $out/bench_fista.json
python bench.py --dtype float64 --steps 2 --no-cpu-baseline --no-other-configs > $out/bench_f64.json 2> $out/bench_f64.log
python tools/kernel_summary.py $out/bench_f64.json
python tools/profile_admm.py > $out/profile.log 2>&1; python tools/profile_admm.py --algo fista >> $out/profile.log 2>&1; python tools/profile_admm.py --raw >> $out/profile.log 2>&1; cat $out/profile.log
ls $out | tr '\n' ' '
