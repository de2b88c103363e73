#!/bin/bash
out=gpurun_out/r05z; mkdir -p $out
(time python -m pytest tests -m gpu -q -x --durations=6 > $out/gputests.log 2>&1); tail -4 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/gpu_prof.sh r05z_c4 --config c4 > /dev/null 2>&1; python tools/summarize_prof.py r05z_c4 gpurun_out/r05z_c4 > /dev/null
cp profiles/r05z_c4_counters.md profiles/r05z_c4_kernel_stats.csv profiles/traffic.json $out/
(time python bench.py > $out/bench.json 2> $out/bench.log); tail -2 $out/bench.log
python tools/kernel_summary.py $out/bench.json; python tools/other_summary.py $out/bench.json
