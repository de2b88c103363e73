#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export MPLBACKEND=Agg
(time python -m pytest tests -m gpu -q -x --durations=10 -s > $out/gputests.log 2>&1); grep -v "^$" $out/gputests.log | grep "C2\|C3\|C5\|TV-active\|passed\|failed\|FAILED\|Error" | tail -20
(python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.log); tail -2 $out/bench.log
python tools/kernel_summary.py $out/bench.json; python tools/other_summary.py $out/bench.json
python bench.py --dtype float64 --no-cpu-baseline --no-other-configs --steps 2 > $out/bench_f64.json 2> $out/bench_f64.log; python tools/kernel_summary.py $out/bench_f64.json
python bench.py --algo fista --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/bench_fista.json 2> $out/bench_fista.log; python tools/kernel_summary.py $out/bench_fista.json
